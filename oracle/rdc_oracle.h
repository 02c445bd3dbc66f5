/* rdc_oracle.h — TEST INFRASTRUCTURE ONLY (see rdc_oracle.c header).  PARITY UNPINNED. */
#ifndef RDC_ORACLE_H
#define RDC_ORACLE_H
#include <stdint.h>
#include "../include/rdc_assembly.h" /* POD parameter structs only */
#ifdef __cplusplus
extern "C" {
#endif
int oracle_nqp(int elem_type);
int oracle_fe_reinit(int elem_type, const double* X, double* phi, double* dphi, double* JxW);
void oracle_pihna_element(int nen, int nqp, const double* phi, const double* dphi, const double* JxW,
                          const double* u, const rdc_pihna_params* P, double* Ke, double* Fe);
void oracle_ripf_element(int nen, int nqp, const double* phi, const double* dphi, const double* JxW,
                         const double* u, const double* aux, const rdc_ripf_params* P, double* Ke, double* Fe);
void oracle_hcc_element(int nen, int nqp, const double* phi, const double* dphi, const double* JxW,
                        const double* u, const rdc_hcc_params* P, double* Ke, double* Fe);
void oracle_hyperelastic_point(const double* gradX9, const double* lambda3, const double* fibre3,
                               double Young, double Poisson, double K, double* sigma9, double* C36);
void oracle_solid_element(int nen, int nqp, const double* dphi, const double* JxW, const double* Xu,
                          const double* fibre3, const rdc_solid_material* M, double pseudo_time,
                          int request_jacobian, int use_symmetry, double* Je, double* Re);
void oracle_solid_side(int nen, int side, const double* x, const double* Xu, const double* disp3,
                       double pseudo_time, double penalty, int request_jacobian, double* Je, double* Re);
int64_t oracle_build_node_pattern(int nen, int64_t n_elem, int64_t n_node, int64_t n_owned,
                                  const uint32_t* conn, int64_t* bptr, int32_t* bcol);
void oracle_expand_pattern(int nvar, int64_t n_owned, const int64_t* bptr, const int32_t* bcol,
                           int64_t* row_ptr, int32_t* col_idx);
int oracle_assemble(int model, int elem_type, int64_t e_begin, int64_t e_end, int64_t n_owned,
                    const uint32_t* conn, const double* xyz, int nvar, const double* u_old,
                    const double* aux_nodal, const double* xyz_undeformed, const double* elem_fibre,
                    const int32_t* elem_material, const rdc_solid_material* materials,
                    const void* params, int request_jacobian, const int64_t* row_ptr,
                    const int32_t* col_idx, double* val, double* rhs);
/* oracle_assemble on n_threads host cores (rows split over threads; bitwise equal to the serial loop) */
int oracle_assemble_mt(int n_threads, int model, int elem_type, int64_t e_begin, int64_t e_end, int64_t n_owned,
                       const uint32_t* conn, const double* xyz, int nvar, const double* u_old,
                       const double* aux_nodal, const double* xyz_undeformed, const double* elem_fibre,
                       const int32_t* elem_material, const rdc_solid_material* materials,
                       const void* params, int request_jacobian, const int64_t* row_ptr,
                       const int32_t* col_idx, double* val, double* rhs);
int oracle_assemble_solid_sides(int elem_type, int64_t n_sides, const int64_t* side_elem,
                                const int32_t* side_id, const double* side_disp, int64_t n_owned,
                                const uint32_t* conn, const double* xyz, const double* xyz_undeformed,
                                const rdc_solid_params* sp, int request_jacobian, const int64_t* row_ptr,
                                const int32_t* col_idx, double* val, double* rhs);
void oracle_adpm_element(int nen, int nqp, const double* phi, const double* dphi, const double* JxW,
                         const double* u, const double* tracts3, const rdc_adpm_params* P, double* Ke, double* Fe);
void oracle_proteas_element(int nen, int nqp, const double* phi, const double* dphi, const double* JxW,
                            const double* u, const double* aux0, const rdc_proteas_params* P, double* Ke, double* Fe);
int oracle_pihna_volume_integrals(int elem_type, int64_t n_elem, const uint32_t* conn, const double* xyz, const double* u,
                                  const rdc_pihna_ranges* r, double* out);
int oracle_ripf_volume_integrals(int elem_type, int64_t n_elem, const uint32_t* conn, const double* xyz, const double* u,
                                 const rdc_ripf_ranges* r, double* out);
int oracle_adpm_parcellation_integrals(int elem_type, int64_t n_elem, const uint32_t* conn, const double* xyz,
                                       const double* u, const rdc_adpm_ranges* r, const int32_t* elem_subdomain,
                                       const int32_t* ids, int32_t n_ids, double* out);
void oracle_clamp_nonnegative(double* u, int64_t n);
void oracle_stress_measures(const double* A9, double* ev, double* pressure, double* von_mises);
int oracle_solid_post_process(int elem_type, int64_t n_elem, const uint32_t* conn, const double* xyz,
                              const double* xyz_undeformed, const double* elem_fibre, const int32_t* elem_material,
                              const rdc_solid_material* materials, double pseudo_time, double* pressure,
                              double* von_mises, double* fibre_current);
double oracle_ripf_check_solution(int64_t n, const rdc_ripf_check_params* p, double* sol, double* prev, double* td,
                                  double* rt, double* aux);
#ifdef __cplusplus
}
#endif
#endif
