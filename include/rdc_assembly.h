/*
 * rdc_assembly.h — C-ABI of the MI355X-native element-assembly path of rdcFEs.
 *
 * This is the drop-in boundary.  Every entry point replaces a piece of
 * what the reference does inside its libMesh assemble callbacks:
 *
 *   reference (file:line, under the upstream tree)             this header
 *   ---------------------------------------------------------  ---------------------------
 *   mesh.active_local_element_ptr_range()   src/pihna.C:383     rdc_mesh_upload
 *   DofMap::dof_indices                     src/pihna.C:385-394 rdc_mesh_upload (dof = node*nvar+var)
 *   es.init() sparsity pattern              src/pihna.C:48      rdc_csr_dims / rdc_csr_pattern_download
 *   system.old_solution(dof)                src/pihna.C:433     rdc_field_upload(RDC_FIELD_OLD_SOLUTION)
 *   TD_system / RT_system.current_solution  src/ripf.C:470,477  rdc_field_upload(RDC_FIELD_AUX_NODAL)
 *   aux_system undeformed coordinates       src/solid_system.C:221-229  rdc_field_upload(RDC_FIELD_UNDEFORMED_XYZ)
 *   fibre_sys / subdomain_id                src/solid_system.C:183-216  rdc_field_upload(RDC_FIELD_ELEM_FIBRE), rdc_solid_set_materials
 *   mesh node positions (moving mesh)       src/solid_system.C:103-123  rdc_mesh_update_coords
 *   assemble_pihna                          src/pihna.C:318-758 rdc_assemble_pihna
 *   assemble_ripf                           src/ripf.C:337-673  rdc_assemble_ripf
 *   assemble_hcc                            src/coupled_hcc.C:414-649   rdc_assemble_hcc
 *   SolidSystem::element_time_derivative    src/solid_system.C:146-271  rdc_solid_assemble
 *   SolidSystem::side_time_derivative       src/solid_system.C:273-371  rdc_solid_assemble (sides set by rdc_solid_set_sides)
 *   matrix.add_matrix / rhs->add_vector     src/pihna.C:754-755 results: rdc_csr_values_device_ptr / rdc_csr_download
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every call returns an int status (RDC_OK == 0);
 *     rdc_last_error() gives a human-readable message for the last failure on a context.
 *   - all caller buffers are caller-owned; the library copies what it needs.
 *   - one context == one GPU == one mesh partition == one system of `nvar` FIRST-LAGRANGE variables.
 *     Calls on one context are not thread-safe (the reference callback is single-threaded per rank).
 *   - DoF numbering: dof = local_node * nvar + var  (libMesh variable-group numbering).
 *   - nodes [0, n_owned_nodes) are owned: their matrix rows / rhs entries are assembled completely
 *     on this context.  Nodes [n_owned_nodes, n_nodes) are ghosts (values supplied by halo exchange).
 *     The element list must contain every element touching an owned node (owned + one ghost layer).
 *   - matrix = scalar CSR (PETSc AIJ layout) over the owned rows, column indices are LOCAL dof ids
 *     (owned and ghost), sorted ascending within a row.  Row r = node*nvar + var.
 *   - there is NO CPU fallback: without a usable HIP device rdc_ctx_create fails with RDC_ERR_HIP.
 */
#ifndef RDC_ASSEMBLY_H
#define RDC_ASSEMBLY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDC_ABI_VERSION 3   /* 2: round-2 additions (solid post-process, ADPM/PROTEAS, chunked hand-back); 3: per-part entry points, part-1 bound recorded by the call */

/* status codes */
#define RDC_OK               0
#define RDC_ERR_INVALID      1   /* bad argument (null pointer, bad size, out-of-range index) */
#define RDC_ERR_HIP          2   /* HIP runtime failure / no device */
#define RDC_ERR_STATE        3   /* call order violated (e.g. assemble before mesh upload) */
#define RDC_ERR_UNSUPPORTED  4   /* element type / model combination not implemented */
#define RDC_ERR_ALLOC        5   /* host or device allocation failed */

/* element types (value == nodes per element) */
#define RDC_TET4 4
#define RDC_HEX8 8

/* scatter strategies */
#define RDC_SCATTER_AUTO       0  /* pick the fastest implemented for (model, element type) */
#define RDC_SCATTER_COLOURED   1  /* element-parallel, colour batches, plain read-modify-write, no atomics */
#define RDC_SCATTER_ROWGATHER  2  /* row-owner gather: every CSR value written exactly once, streaming stores */

/* kernel variants (test / benchmarking hook) */
#define RDC_VARIANT_AUTO     0  /* factored TET4 kernels where available, generic otherwise */
#define RDC_VARIANT_GENERIC  1  /* always the generic quadrature-loop kernels */

/* fields */
#define RDC_FIELD_OLD_SOLUTION   0  /* [n_nodes][nvar]   system.old_local_solution                     */
#define RDC_FIELD_AUX_NODAL      1  /* [n_nodes][naux]   RIPF: {cc_dtime, fb_dtime, RT_total}          */
#define RDC_FIELD_UNDEFORMED_XYZ 2  /* [n_nodes][3]      solid: "SolidSystem::auxiliary"               */
#define RDC_FIELD_ELEM_FIBRE     3  /* [n_elem][3]       solid: fibre direction, vars 0..2 of "::fibre" */
#define RDC_FIELD_ELEM_TRACTS    3  /* [n_elem][3]       ADPM: the "Tracts" system (src/adpm.C:448-453); same slot as the fibre */
#define RDC_FIELD_PREV_SOLUTION  4  /* [n_nodes][nvar]   RIPF check_solution: prev_soln (src/ripf.C:675,769)         */
#define RDC_FIELD_TIME_DERIV     5  /* [n_nodes][nvar]   "RIPF-TimeDeriv" system (src/ripf.C:738-740)                */
#define RDC_FIELD_RT_DOSE        6  /* [n_nodes][3]      "RT" system {broad, focus, total} (src/ripf.C:749-760)      */
#define RDC_FIELD_COUNT          7

typedef struct rdc_ctx rdc_ctx;

/* ---- parameters: POD mirrors of the string-keyed es.parameters the callbacks read ---- */

/* src/pihna.C:358-381.  necrosis_* are the RAW input values; the division by
 * cells_max_capacity (src/pihna.C:364-366) happens inside the library, as in the reference. */
typedef struct rdc_pihna_params {
  double time_step;                    /* "time_step"                   */
  double cells_min_capacity;           /* "cells_min_capacity"  Lambda_k */
  double cells_max_capacity;           /* "cells_max_capacity"  Kappa_k  */
  double cytokines_max_capacity;       /* "cytokines_max_capacity" Kappa_a */
  double cells_max_capacity_exponent;  /* "cells_max_capacity/exponent" ek */
  double necrosis_c, necrosis_h, necrosis_v;   /* "necrosis/c|h|v" */
  double diffuse_c, taxis_c;           /* "diffuse/c", "taxis/c" */
  double diffuse_h, taxis_h;           /* "diffuse/h", "taxis/h" */
  double produce_c;                    /* "produce/c" */
  double switch_c2h, switch_h2c, switch_h2n;   /* "switch/c/to/h" ... */
  double diffuse_v, taxis_v, produce_v;        /* "diffuse/v", "taxis/v", "produce/v" */
  double secrete_a_c, secrete_a_h;     /* "secrete/a/from/c|h" */
  double uptake_a_v, decay_a;          /* "uptake/a/from/v", "decay/a" */
} rdc_pihna_params;

/* src/ripf.C:377-408.  lambda_RT_r / omicro_RT_r: if 0 the reference substitutes the runtime
 * value "RT_dose/total/max" (src/ripf.C:398-403); pass that in RT_dose_total_max. */
typedef struct rdc_ripf_params {
  double time_step;
  double VolFr_stroma, VolFr_parenchyma, VolFr_exponent, VolFr_min_vacant, VolFr_max_vacant;
  double phi_cc_B, phi_cc_D, phi_cc, phi_fb_B, phi_fb_D, phi_fb, phi_tol;
  double kappa, kappa_RT_c, delta, delta_RT_a, delta_RT_b;
  double lambda, lambda_RT_r, lambda_HU_r;
  double omicro, omicro_RT_r, omicro_fb_b;
  double omega, diffusion, haptotaxis, radiotaxis;
  int32_t RT_dose_total_max;           /* es.parameters.get<int>("RT_dose/total/max") */
  int32_t _pad;
} rdc_ripf_params;

/* src/coupled_hcc.C:450-461 (necrosis_* raw; divided by cells/max_capacity inside). */
typedef struct rdc_hcc_params {
  double time_step;
  double cells_min_capacity, cells_max_capacity, cells_max_capacity_exponent;
  double produce_l;
  double diffuse_c, mechano_c, produce_c;
  double necrosis_l, necrosis_c, necrosis_pressure;
} rdc_hcc_params;

/* per-subdomain material, src/solid_system.C:183-190 */
/* es.parameters of assemble_adpm, src/adpm.C:367-413 (keys in the comments; defaults src/adpm.C:163-233).
 * pulse / sigmoid triples = {magnitude, c0, c1}, trapezoid = {magnitude, c0, c1, c2, c3} (src/utils.h:100-187). */
typedef struct rdc_adpm_params {
  double time_step;                 /* "time_step"                                             */
  double time;                      /* system.time: decay/PrP is scaled by pow(time, exponent) */
  double decay_PrP_time_exponent;   /* "decay/PrP/time_exponent"                               */
  double decay_PrP[3];              /* "decay/PrP", ".../pulse/0", ".../pulse/1"               */
  double transform_A_b[5];          /* "transform/A_b", ".../trapezoid/0..3"                   */
  double transform_Tau[5];          /* "transform/Tau", ".../trapezoid/0..3"                   */
  double diffuse_A_b[3];            /* "diffuse/A_b", ".../pulse/0,1"                          */
  double taxis1_A_b[3];             /* "taxis_1/A_b", ".../pulse/0,1"                          */
  double taxis2_A_b[3];             /* "taxis_2/A_b", ".../pulse/0,1"                          */
  double produce_A_b[3];            /* "produce/A_b", ".../sigmoid/0,1"                        */
  double decay_A_b[3];              /* "decay/A_b", ".../pulse/0,1"                            */
  double diffuse_Tau[3];            /* "diffuse/Tau", ...                                      */
  double taxis1_Tau[3];             /* "taxis_1/Tau", ...                                      */
  double taxis2_Tau[3];             /* "taxis_2/Tau", ...                                      */
  double produce_Tau[3];            /* "produce/Tau", ".../sigmoid/0,1"                        */
  double decay_Tau[3];              /* "decay/Tau", ...                                        */
  double taxis_A_b_angle;           /* "taxis/A_b/angle" in RADIANS (input() converts degrees, src/adpm.C:193) */
  double taxis_Tau_angle;           /* "taxis/Tau/angle" in radians                            */
} rdc_adpm_params;

/* es.parameters of assemble_proteas_model, src/proteas.C:376-409 (all defaults 1.0, time_step 1e-9; :135,180-212) */
typedef struct rdc_proteas_params {
  double time_step;            /* "time_step"                */
  double cells_total_capacity; /* "cells/total_capacity"     */
  double RT_max_dosage;        /* "radiotherapy/max_dosage"  */
  double host_proliferation, host_vsc_threshold, host_RT_death_rate, host_RT_exp_a, host_RT_exp_b, host_necrosis_rate;
  double tumour_diffusion, tumour_diffusion_host, tumour_proliferation, tumour_vsc_threshold, tumour_RT_death_rate,
         tumour_RT_exp_a, tumour_RT_exp_b, tumour_necrosis_rate;
  double necrosis_clearance, necrosis_slope, necrosis_vsc_threshold;
  double vascular_proliferation, vascular_necrosis_rate;
  double oedema_diffusion, oedema_proliferation, oedema_vsc_threshold, oedema_RT_coeff, oedema_RT_exp,
         oedema_reabsorption_rate;
} rdc_proteas_params;

typedef struct rdc_solid_material {
  double Young, Poisson, FibreStiffness;
  double rate[3];                      /* VolumetricStretchRatio/rate_0..2 */
} rdc_solid_material;

/* src/solid_system.C:181,234,291,306 */
typedef struct rdc_solid_params {
  double pseudo_time;                  /* "pseudo_time" */
  double displacement_penalty;         /* "BCs/displacement_penalty" */
  int32_t use_symmetry;                /* "solver/assembly_use_symmetry" */
  int32_t _pad;
} rdc_solid_params;

/* ---- context ---- */
int rdc_abi_version(void);
/* GPUs visible to this process (one MPI rank drives one of them: device_ordinal = node-local rank % count) */
int rdc_device_count(int* n_devices);
int rdc_ctx_create(int device_ordinal, rdc_ctx** out);
int rdc_ctx_destroy(rdc_ctx* ctx);
/* message for the last failing call on ctx (ctx may be NULL: last rdc_ctx_create failure) */
const char* rdc_last_error(const rdc_ctx* ctx);
/* run all work of this context on an externally created hipStream_t (NULL = default stream) */
int rdc_set_stream(rdc_ctx* ctx, void* hip_stream);
int rdc_synchronize(rdc_ctx* ctx);
int rdc_set_scatter(rdc_ctx* ctx, int strategy);
int rdc_get_scatter(const rdc_ctx* ctx, int* strategy);
int rdc_set_kernel_variant(rdc_ctx* ctx, int variant);
/* tuning / profiling knobs, not needed for normal use.  "occupancy": launch-bound waves per SIMD of
 * the TET4 row-gather kernel; "ablate": 1..6 remove parts of that kernel (results are then WRONG);
 * "moments": 1 (default) evaluates PIHNA/TET4 rows in moment form when the parameters have the shipped pattern (cell
 * transport off), 0 in coefficient form -- same sums, other association; "specialise": 0 disables that parameter-
 * pattern variant altogether; "kernel", "staged", "slim", "stagger", "prefetch", "xcd", "schedule", "block", "grid",
 * "ev_occupancy", "ev_lds", "evc_occupancy", "ev_resident" (1, default = whole-mesh launches of at least 56 clusters per resident workgroup run as three resident
 * workgroups per CU that fetch the whole next cluster by LDS-DMA while the current one is expanded and copied out, clusters handed
 * out by a counter; 2 = launches of any size; 0 = never) -- the others are experimental and
 * slower than the default -- select alternative / diagnostic kernels.  PIHNA / TET4, element-visit kernel:
 * "ev_general" 1 (default) = any parameter values through the kernel with all 22 moments, 0 = through the (node, element) pair
 * kernel as before round 3; "ev_background" 1 (default) = waves all of whose elements are in the background state of the shipped
 * field file (n = c = h = a = 0, v > 0) do not evaluate the moments and right-hand sides that are sums of exact zeros -- same
 * matrix, the kernel time then depends on the state --, 0 = everything is evaluated.  HEX8 with three unknowns: "hex_kernel" 0 =
 * producer / consumer cluster kernel (default), 1 = (node, element) pair kernels, 2 = persistent form of the cluster
 * kernel; "solid_kernel" 0 = fused cluster kernel for HEX8 tangent requests and the two-pass form otherwise (default),
 * 1 = coloured read-modify-write, 2 = two-pass always (bitwise reproducible sums), 3 = fused (error when unavailable);
 * "solid_cl_waves", "solid_cl_order" = shape / pair order of the cluster lists; "solid_gather", "solid_split",
 * "solid_store" = diagnostics of the solid kernels.
 * Two-part assembly, for overlapping a halo exchange with the assembly of rows that do not need it:
 * "interior_nodes" = n states that no element of the owned nodes [0, n) contains a ghost node (the caller numbers
 * its owned nodes interior-first; set it BEFORE rdc_mesh_upload where possible: the work lists are then built so that
 * part 1 covers every interior row, otherwise only the workgroups that happen to lie inside [0, n)); with "part" = 1 an assemble call then writes only the rows of leading workgroups
 * inside [0, n), with "part" = 2 the remaining rows, with "part" = 0 (default) all rows.  Part 1 followed by part 2
 * gives exactly the matrix and residual of one whole call.  The TET4 row-gather kernels and the HEX8 cluster kernels
 * (reaction-diffusion models and the fused solid tangent; their cluster lists keep interior and near-ghost nodes apart, the
 * penalty sides of the solid system are added by part 2 behind part 1) launch sub-ranges; the paths that cannot (the
 * HEX8 pair kernels, the coloured strategy, the two-pass solid form) write nothing in part 1 and everything in part 2.
 * Stream contract: part 1 and part 2 of a step may be issued on DIFFERENT streams (rdc_set_stream in between).  Part 1
 * reads the values of owned nodes only, so the ghost rows of the bound fields may be rewritten (halo exchange)
 * while it runs; part 2 must be issued behind that exchange on its stream.  The library itself orders part 2 behind
 * part 1's preparation of the owned node data (an internal event), and neither part writes data the other may be
 * reading.  The caller orders the NEXT step's exchange behind this step's part 2 (it reads the ghost rows). */
int rdc_set_option(rdc_ctx* ctx, const char* key, int value);

/* ---- mesh / pattern (one-time set-up; replaces es.init()) ---- */
/* conn: [n_elem][elem_type] local node ids, libMesh/Gmsh node order; xyz: [n_node][3]. */
int rdc_mesh_upload(rdc_ctx* ctx, int elem_type, int64_t n_elem, int64_t n_node,
                    int64_t n_owned_nodes, const uint32_t* conn, const double* xyz, int nvar);
/* moving mesh: new CURRENT node positions (host pointer), same numbering */
int rdc_mesh_update_coords(rdc_ctx* ctx, const double* xyz);
/* device pointer of the current coordinates [n_node][3] (e.g. to update them in place on the GPU) */
int rdc_mesh_coords_device_ptr(rdc_ctx* ctx, double** d_xyz);
int rdc_mesh_dims(const rdc_ctx* ctx, int64_t* n_elem, int64_t* n_node, int64_t* n_owned_nodes,
                  int* elem_type, int* nvar, int* n_colours);
int rdc_csr_dims(const rdc_ctx* ctx, int64_t* n_rows, int64_t* nnz);
int rdc_csr_pattern_download(const rdc_ctx* ctx, int64_t* row_ptr, int32_t* col_idx);
/* element colouring used by the coloured scatter (test hook): colour id per element */
int rdc_mesh_colours_download(const rdc_ctx* ctx, int32_t* colour_of_elem);

/* ---- fields ---- */
int rdc_field_upload(rdc_ctx* ctx, int field, const double* host, int64_t count);
int rdc_field_download(rdc_ctx* ctx, int field, double* host, int64_t count);
/* library-owned device storage of a field (allocated on first use) */
int rdc_field_device_ptr(rdc_ctx* ctx, int field, int64_t count, double** d_ptr);
/* use caller-owned device memory for a field (must stay valid until rebound / ctx destroyed) */
int rdc_field_bind_device(rdc_ctx* ctx, int field, double* d_ptr, int64_t count);

/* ---- solid-only set-up ---- */
/* subdomain index per element (index into materials[]), materials table */
int rdc_solid_set_materials(rdc_ctx* ctx, const int32_t* elem_material, int32_t n_materials,
                            const rdc_solid_material* materials);
/* boundary sides with a displacement BC: element id, libMesh side number, prescribed displacement
 * (NaN component = unconstrained, src/solid_system.C:346,358) */
int rdc_solid_set_sides(rdc_ctx* ctx, int64_t n_sides, const int64_t* side_elem,
                        const int32_t* side_id, const double* side_displacement /* [n_sides][3] */);

/* ---- the hot path: one call == one invocation of the reference's assemble callback ----
 * Pre: mesh + fields uploaded.  Post: CSR values and rhs hold the assembled sums for the owned rows
 * (the library zeroes/overwrites them itself: libMesh zeroes matrix & rhs before the callback). */
int rdc_assemble_pihna(rdc_ctx* ctx, const rdc_pihna_params* p);
int rdc_assemble_ripf(rdc_ctx* ctx, const rdc_ripf_params* p);
int rdc_assemble_hcc(rdc_ctx* ctx, const rdc_hcc_params* p);
/* assemble_adpm, src/adpm.C:324-652: unknowns PrP, A_b, Tau; needs RDC_FIELD_ELEM_TRACTS */
int rdc_assemble_adpm(rdc_ctx* ctx, const rdc_adpm_params* p);
/* assemble_proteas_model, src/proteas.C:338-705: unknowns host, tumour, necrotic, vascular,
 * oedema; RDC_FIELD_AUX_NODAL = {AUX "HU", AUX "RTD", 0} per node.  As upstream, the radiotherapy dose at a point is
 * phi_1 * (AUX variable 0 at LOCAL NODE 1 of the element) (src/proteas.C:481): only component 0 is read, and only
 * at that node of every element. */
int rdc_assemble_proteas(rdc_ctx* ctx, const rdc_proteas_params* p);
/* residual (+ Jacobian if request_jacobian) of the SolidSystem; nvar must be 3 */
int rdc_solid_assemble(rdc_ctx* ctx, const rdc_solid_params* p, int request_jacobian);
/* One part of a two-part step in ONE call: as the assemble call of the same name with "part" = part (0 = whole, 1, 2) and
 * on `hip_stream` (a hipStream_t; NULL = the default stream) for this call only -- the context's "part" option and stream
 * are left as they were.  The step of a partitioned run (src/pihna.C:801 system.update() overlapped with the assembly of
 * the interior rows) is then: rdc_assemble_pihna_part(ctx, p, 1, main); [halo exchange on side]; rdc_assemble_pihna_part(ctx, p, 2, side). */
int rdc_assemble_pihna_part(rdc_ctx* ctx, const rdc_pihna_params* p, int part, void* hip_stream);
int rdc_assemble_hcc_part(rdc_ctx* ctx, const rdc_hcc_params* p, int part, void* hip_stream);
int rdc_solid_assemble_part(rdc_ctx* ctx, const rdc_solid_params* p, int request_jacobian, int part, void* hip_stream);

/* ---- results ---- */
int rdc_csr_values_device_ptr(rdc_ctx* ctx, double** d_val, double** d_rhs);
int rdc_csr_download(rdc_ctx* ctx, double* val, double* rhs);
/* Chunked hand-back: the CSR values and rhs entries of the rows of nodes [node_begin, node_end) only, written to the
 * SAME positions of the full-size host arrays val / rhs as rdc_csr_download uses (either may be NULL).  async != 0:
 * the copies are only enqueued on the context's stream (use pinned host memory; complete after rdc_synchronize or an
 * event of the caller) -- so the rows of part 1 of a two-part assembly can travel while part 2 is still computing. */
int rdc_csr_download_rows(rdc_ctx* ctx, int64_t node_begin, int64_t node_end, double* val, double* rhs, int async);
/* Pipelined hand-back (src/pihna.C:754-755 hands Ke/Fe to PETSc element by element; here whole row ranges travel while the
 * device keeps assembling): as rdc_csr_download_rows, but the copies run on a stream of the context's own, behind the work
 * ENQUEUED SO FAR on the context's stream and independent of anything enqueued later -- the rows of part 1 of a two-part
 * assembly, or of the first node ranges of any assembly, are inserted into the PETSc matrix while the next chunk is in flight.
 * *ticket identifies the call; rdc_ticket_wait blocks the host until that chunk is in host memory (at most 16 calls in flight).
 * Pin the destination arrays once (rdc_host_pin: hipHostRegister) or the copies are staged and serialise. */
int rdc_csr_download_rows_async(rdc_ctx* ctx, int64_t node_begin, int64_t node_end, double* val, double* rhs, int* ticket);
int rdc_ticket_wait(rdc_ctx* ctx, int ticket);
int rdc_host_pin(rdc_ctx* ctx, void* host_ptr, size_t bytes);
int rdc_host_unpin(rdc_ctx* ctx, void* host_ptr);
/* two-part assembly: the rows of nodes [0, *n_nodes) are complete after "part" = 1 (whole workgroups inside the
 * interior; 0 when the active kernel path cannot launch sub-ranges).  After a part-1 assemble call the value is what
 * THAT call completed -- the kernel path, and with it the split, depends on the model, the parameter values and the
 * tuning options; before any part-1 call since the upload it is the prediction for the default path. */
int rdc_part1_nodes(const rdc_ctx* ctx, int64_t* n_nodes);

/* ---- post-solve nodal kernel: negativity clamp of check_solution,
 * src/pihna.C:785-790, applied in place to a device-resident field ---- */
int rdc_clamp_nonnegative(rdc_ctx* ctx, int field);

/* RIPF check_solution, src/ripf.C:675-775, on device-resident fields of a 3-variable context.  Per node:
 *   RDC_FIELD_OLD_SOLUTION  (the freshly solved state) is clamped in place: HU to [HU_min, HU_max], cc, fb >= 0 (:722-724);
 *   RDC_FIELD_TIME_DERIV    = (clamped - RDC_FIELD_PREV_SOLUTION) / time_step                          (:738-740);
 *   RDC_FIELD_RT_DOSE[2]    = total dose of the fractionation schedule on `day` from [0] broad, [1] focus (:754-758);
 *   RDC_FIELD_PREV_SOLUTION = the UNCLAMPED solved state, as upstream's `prev_soln = soln`             (:769);
 *   RDC_FIELD_AUX_NODAL     = {TIME_DERIV[1], TIME_DERIV[2], RT total}: what the next rdc_assemble_ripf reads (:470-478).
 * *rt_total_max receives max(-1, max over the context's nodes of RT total) (:705,761); upstream stores it truncated
 * to int in "RT_dose/total/max" and aborts when it is <= 0 (:771-772) -- with several ranks reduce it with MAX first.
 * TIME_DERIV, AUX_NODAL are (re)allocated as needed; OLD_SOLUTION, PREV_SOLUTION, RT_DOSE must have been set. */
typedef struct rdc_ripf_check_params {
  double time_step;              /* "time_step"                 */
  double HU_min, HU_max;         /* "HU/min", "HU/max"          */
  int32_t RT_broad_fractions;    /* "RT_dose/broad/fractions"   */
  int32_t RT_focus_fractions;    /* "RT_dose/focus/fractions"   */
  int32_t day;                   /* floor(system.time), :703    */
  int32_t _pad;
} rdc_ripf_check_params;
int rdc_ripf_check_solution(rdc_ctx* ctx, const rdc_ripf_check_params* p, double* rt_total_max);

/* SolidSystem::post_process, src/solid_system.C:394-538: per element the plain average over
 * the quadrature points of the Cauchy stress and of F*eta, then hydrostatic pressure (:522), von Mises stress (:524)
 * -- upstream takes both from the principal stresses of eigen_decomposition (src/eig3.C:261-271); they are the
 * invariants tr/3 and sqrt(I1^2 - 3 I2), which is what the kernel evaluates -- and the current fibre vector (:526).
 * Uses the mesh coordinates, RDC_FIELD_UNDEFORMED_XYZ, RDC_FIELD_ELEM_FIBRE and the materials of the context.
 * Host outputs (any may be NULL): pressure [n_elem], von_mises [n_elem], fibre_current [n_elem][3]. */
int rdc_solid_post_process(rdc_ctx* ctx, const rdc_solid_params* p, double* pressure, double* von_mises,
                           double* fibre_current);

/* save_solution of PIHNA, src/pihna.C:842-976 ("CSV volume integrals"): the four element-volume
 * sums of the CSV line -- an element counts when ALL its nodes have (c+h), n, v, (n+c+h+v)/cells_max_capacity inside
 * the closed range -- over the first n_elem elements of the context (pass the number of elements the rank owns, or
 * -1 for all; upstream loops over every active element on rank 0), read from RDC_FIELD_OLD_SOLUTION (5 unknowns).
 * out[4] = {ACTIVE_TUMOR_VOLUME, NECROTIC_VOLUME, VASCULARITY_VOLUME, TOTAL_CELL_VOLUME}; with several ranks add them. */
typedef struct rdc_pihna_ranges {
  double active_tumor_min, active_tumor_max;   /* "range/active_tumor/min,max" */
  double necrotic_min, necrotic_max;           /* "range/necrotic/min,max"     */
  double vascularity_min, vascularity_max;     /* "range/vascularity/min,max"  */
  double total_cell_min, total_cell_max;       /* "range/total_cell/min,max"   */
  double cells_max_capacity;                   /* "cells_max_capacity"         */
} rdc_pihna_ranges;
int rdc_pihna_volume_integrals(rdc_ctx* ctx, const rdc_pihna_ranges* r, int64_t n_elem, double* out4);

/* RIPF save_solution (replaces the element loop of src/ripf.C:812-858): out[2] = {Tumour_Volume, Fibrosis_Volume},
 * the volume of the elements ALL of whose nodes have HU inside [HU_min, HU_max] and cc (fb) >= min; read from
 * RDC_FIELD_OLD_SOLUTION (3 unknowns HU, cc, fb); first n_elem elements, -1 for all; with several ranks add. */
typedef struct rdc_ripf_ranges {
  double cc_HU_min, cc_HU_max, cc_min;   /* "range_cc/HU/min,max", "range_cc/min" */
  double fb_HU_min, fb_HU_max, fb_min;   /* "range_fb/HU/min,max", "range_fb/min" */
} rdc_ripf_ranges;
int rdc_ripf_volume_integrals(rdc_ctx* ctx, const rdc_ripf_ranges* r, int64_t n_elem, double* out2);

/* ADPM save_solution (replaces the element loop of src/adpm.C:747-813) over the brain parcellation = the set of
 * subdomain ids: elem_subdomain[n_elem] is elem->subdomain_id(), ids[n_ids] the parcellation (ascending, as the
 * std::set iterates).  out[n_ids][4] = {CONCENTRATION__A_b, CONCENTRATION__Tau, VOLUME__A_b, VOLUME__Tau}: the
 * volumes sum the elements of the region ALL of whose nodes lie inside the range (add over ranks); the
 * concentrations are -- as upstream, which assigns instead of accumulating (:780-783) -- the element average
 * sum_qp JxW*value / volume of the LAST element of the region in element order, and last_elem[n_ids] (may be
 * NULL) returns that element (-1: region empty here) so that several ranks can keep the one with the highest
 * global element.  Unknowns (PrP, A_b, Tau) from RDC_FIELD_OLD_SOLUTION (nvar = 3).  n_ids <= 2048. */
typedef struct rdc_adpm_ranges {
  double A_b_min, A_b_max;   /* "range/A_b/min,max" */
  double Tau_min, Tau_max;   /* "range/Tau/min,max" */
} rdc_adpm_ranges;
int rdc_adpm_parcellation_integrals(rdc_ctx* ctx, const rdc_adpm_ranges* r, const int32_t* elem_subdomain,
                                    const int32_t* ids, int32_t n_ids, int64_t n_elem, double* out,
                                    int64_t* last_elem);

/* ---- instrumentation ---- */
/* when enabled every rdc_assemble_* brackets its dominant kernel(s) -- the assembly kernel, or all
 * colour launches; not the small node-record pack -- with HIP events on the context stream */
int rdc_timing_enable(rdc_ctx* ctx, int on);
/* device time of the last assemble call in ms (valid after the stream has been synchronised) */
int rdc_timing_last_ms(rdc_ctx* ctx, float* ms);
/* sum of the device times of every assemble call since the last enable / sum, and their count;
 * synchronises on the recorded events and resets the pool (no host sync happens inside assemble) */
int rdc_timing_sum_ms(rdc_ctx* ctx, float* total_ms, int* n_calls);
/* the same, call by call (oldest first) into ms[0 .. min(*n_calls, capacity)): for the median / minimum of a timed run */
int rdc_timing_samples_ms(rdc_ctx* ctx, float* ms, int capacity, int* n_calls);

/* diagnostic only: call with host_out == NULL to arm (the next shipped-parameter PIHNA/TET4 assembly then
 * runs a separately compiled kernel that records s_memtime stamps per workgroup phase; *n_written = number
 * of values), call again with a buffer to fetch them: [workgroup][wave][6] shader-clock stamps.
 * With "kernel" = 7 and "ablate" = 4 set before arming, the stamped build is the element-visit kernel itself (results
 * unchanged): [cluster][wave][12] = 11 stamps + hardware id (tools/ev_timeline.py; "ev_resident" = 1: 6 stamps per cluster). */
int rdc_debug_stamps(rdc_ctx* ctx, long long* host_out, int64_t capacity, int64_t* n_written);

#ifdef __cplusplus
}
#endif
#endif /* RDC_ASSEMBLY_H */
