// rdc_launch.h — dispatch of the generic kernels on (element type, exponent mode, strategy).
// Included by one translation unit per model so the models compile in parallel.
#ifndef RDC_LAUNCH_H
#define RDC_LAUNCH_H
#include "rdc_internal.h"
#include "rdc_hex8_cl_kernel.h"

namespace rdc {

template <class M, int NEN, int EXP_MODE>
static hipError_t launch_rd_impl(const LaunchArgs& a, const typename M::K& k) {
  if (a.ev_start) (void)hipEventRecord(a.ev_start, a.stream);
  if (a.strategy == RDC_SCATTER_ROWGATHER) {
    constexpr int BLOCK = 256;
    if (a.n_wg > 0) {
      if constexpr (NEN == 8 && M::NV == 3) {
        // producer / consumer cluster kernel (rdc_hex8_cl.h) when the context built its lists
        if (a.cl.n_wg > 0) return launch_hex8_cl<M, EXP_MODE>(a, k);
      }
      if constexpr (NEN == 8 && M::NV == 5) {
        // five unknowns: the cluster kernel one equation row at a time
        if (a.cl.n_wg > 0) return launch_hex8_cl_rows<M, EXP_MODE>(a, k);
      }
      if constexpr (NEN == 8) {
        // node-staged form when the workgroup's row slice and node table fit 80 KB of LDS (two workgroups per CU)
        constexpr int REC = 3 + M::NV + (M::NAUX > 0 ? M::NAUX : 0);
        const size_t tab_off = (a.lds_bytes / sizeof(double) + 3) & ~(size_t)1;  // + slice phase shift, rounded to 16 bytes
        const size_t staged_bytes = sizeof(double) * (tab_off + (size_t)a.hx_max_nodes * REC);
        // opt_staged: 1 = where the model profits (M::HEX_STAGED), 2 = always, 0 = never
        if ((a.opt_staged == 2 || (a.opt_staged == 1 && M::HEX_STAGED)) && a.hx_ploc && staged_bytes <= 80 * 1024) {
          static std::atomic<uint64_t> attr_set[1];  /* per instantiation and device */
    dyn_lds_once(attr_set[0], (const void*)k_rowgather_staged<M, NEN, EXP_MODE, BLOCK>, 80 * 1024);
          hipLaunchKernelGGL((k_rowgather_staged<M, NEN, EXP_MODE, BLOCK>), dim3(a.n_wg), dim3(BLOCK), staged_bytes, a.stream,
                             a.m, k, a.u, a.aux, a.elem, a.hx_nl_ptr, a.hx_nlist, a.hx_ploc, (int)tab_off, a.val, a.rhs);
          return hipGetLastError();
        }
      }
      hipLaunchKernelGGL((k_rowgather<M, NEN, EXP_MODE, BLOCK>), dim3(a.n_wg), dim3(BLOCK), a.lds_bytes + 16,
                         a.stream, a.m, k, a.u, a.aux, a.elem, a.val, a.rhs);
    }
    return hipGetLastError();
  }
  // coloured: one launch per colour, stream order is the only synchronisation needed
  for (int c = 0; c < a.n_colours; c++) {
    const int64_t first = a.colour_ptr[c], count = a.colour_ptr[c + 1] - first;
    if (count <= 0) continue;
    const int block = 256;
    const int64_t grid = (count + block - 1) / block;
    hipLaunchKernelGGL((k_coloured<M, NEN, EXP_MODE>), dim3((unsigned)grid), dim3(block), 0, a.stream, a.m, k,
                       first, count, a.u, a.aux, a.elem, a.val, a.rhs);
  }
  return hipGetLastError();
}

template <class M>
hipError_t launch_rd(const LaunchArgs& a, const typename M::K& k) {
  if (a.nen == 4 && a.variant != RDC_VARIANT_GENERIC) {
    // per-element inputs (M::NELEM > 0) and the local-node aux mask (Proteas) travel only in k_tet4_rg5 (needs the pair -> element list) and k_tet4_coloured
    const bool rg5 = a.rg2.n_wg > 0 && a.rg2.pair_aux && a.rg2.nlist && a.rg2.block == 256 && a.rg2.pair_eid;
    if ((M::NELEM == 0 && M::AUX_LOCAL_NODE < 0) || a.strategy != RDC_SCATTER_ROWGATHER || rg5) return launch_tet4_fast<M>(a, k);
  }
  if (a.nen == 4) {
    if (a.exp_mode == M::FAST_EXP_MODE) return launch_rd_impl<M, 4, M::FAST_EXP_MODE>(a, k);
    return launch_rd_impl<M, 4, 0>(a, k);
  }
  if (a.exp_mode == M::FAST_EXP_MODE) return launch_rd_impl<M, 8, M::FAST_EXP_MODE>(a, k);
  return launch_rd_impl<M, 8, 0>(a, k);
}

}  // namespace rdc
#endif
