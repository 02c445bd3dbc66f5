// rdc_solid.h — launch interface of the SolidSystem kernels (rdc_solid.hip).
#ifndef RDC_SOLID_H
#define RDC_SOLID_H
#include "rdc_internal.h"
namespace rdc {
using SolidClDev = ClDev;   // cluster lists of the fused HEX8 kernel (rdc_solid_cl.hip)
struct SolidArgs {
  MeshDev m;
  int nen;
  const double* Xu;            // undeformed coordinates [n_node][3]
  const double* fibre;         // [n_elem][3]
  const int32_t* elem_material;
  const rdc_solid_material* materials;
  int64_t n_sides;
  const int64_t* side_elem;
  const int32_t* side_id;
  const double* side_disp;
  rdc_solid_params params;
  int request_jacobian;
  double* val;
  double* rhs;
  hipStream_t stream;
  const int64_t* colour_ptr;   // host
  int n_colours;
  int kernel;                  // 0 = two-pass (element matrices + gather), 1 = coloured read-modify-write, 3 = fused cluster kernel (HEX8 tangent)
  SolidClDev cl;
  double* ke;                  // [n_elem][nen][nen][3][3] element matrices (two-pass)
  double* fe;                  // [n_elem][nen][3]
  int64_t nblocks;             // node blocks of the owned rows
  const uint32_t* gptr;
  const uint32_t* gsrc;
  const int32_t* brow;
  int split;                   // pass 1: 1 = one thread per element row (default), 0 = columns of a HEX8 row split between two threads
  int store_mode;              // pass 1 diagnostics: 0 = staged stores (default), 1 = direct per-thread stores, 2 = none (timing only)
  int gather;                  // pass 2: 0 = stores staged through LDS (runs of consecutive doubles), 1 = 24-byte pieces
  hipEvent_t sides_wait = nullptr;    // two-part assembly, part 2: the sides add into rows part 1 wrote on another stream: wait for it first
  hipEvent_t done_record = nullptr;   // two-part assembly, part 1: recorded behind the element kernel
};
hipError_t launch_solid(const SolidArgs& a);
hipError_t launch_solid_cl(const SolidArgs& a);   // element part only (rdc_solid_cl.hip); launch_solid adds the sides
size_t solid_cl_lds_bytes(int cw, int pw, size_t max_row_doubles);
hipError_t launch_solid_post(const SolidArgs& a, double* out /* [n_elem][5] device */);
}  // namespace rdc
#endif
