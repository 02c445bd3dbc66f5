// rdc_hex8_cl_kernel.h — the kernel of rdc_hex8_cl.h (HIP only; included by rdc_launch.h).
#ifndef RDC_HEX8_CL_KERNEL_H
#define RDC_HEX8_CL_KERNEL_H
#include "rdc_hex8_cl.h"
#include "rdc_internal.h"

#include <type_traits>

namespace rdc {

// block (a, b) of the model is structurally non-zero (any of the A / B / D coefficient masks)
template <class M>
constexpr bool hex8_cl_block(int a, int b) {
  bool nz = M::hasA(a, b) || M::hasD(a, b);
  for (int g = 0; g < M::NG; g++) nz = nz || M::hasB(a, b, g);
  return nz;
}

template <class M, int EXP_MODE, int CW, int PW, int PPR>
__global__ void __launch_bounds__((CW + PW) * 64, 2)
k_hex8_cl(const MeshDev m, const typename M::K k, const HostPrepCl::Desc* __restrict__ desc, const HostPrepCl::Node* __restrict__ ntab,
          const uint32_t* __restrict__ eid, const uint32_t* __restrict__ pair, const uint32_t* __restrict__ pslot,
          const double* __restrict__ u, const double* __restrict__ aux, const double* __restrict__ elem,
          double* __restrict__ val, double* __restrict__ rhs, const int diag /* "ablate" option: 64 = copy-out with 8-byte stores (timing comparison) */) {
  constexpr int NV = M::NV, NA = (M::NAUX > 0 ? M::NAUX : 1);
  constexpr int MAXP = CW * 64, MAXE = PW * 64, MAXN = CW * 8, NT = (CW + PW) * 64, NW = CW + PW;
  using R = Hex8Rec<M>;
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int w = blockIdx.x;
  // roles rotate over the waves from workgroup to workgroup (wave i of every workgroup sits on SIMD i; rdc_solid_cl.hip)
  const int tid = (int)(((threadIdx.x >> 6) + ((blockIdx.x >> 3) % NW)) % NW) * 64 + (int)(threadIdx.x & 63);
  const HostPrepCl::Desc d = desc[w];
  const bool producer = tid >= MAXP;
  const int nimg = (int)d.row_doubles;
  double* const img = lds;
  double* const lrhs = lds + ((nimg + 1) & ~1);
  // the image of the CSR rows of all owned nodes overlays the point buffers once the points are consumed
  auto zero_image = [&]() {
    v2d_t* z = reinterpret_cast<v2d_t*>(lds);
    const v2d_t zero = {0.0, 0.0};
    for (int x = tid; x < (((nimg + 1) & ~1) + NV * (int)d.nown + 1) / 2; x += NT) z[x] = zero;
  };
  // a half-wave per node: its NV rows are NV^2 * len consecutive doubles of the CSR array
  // (16-byte stores: the node's image segment has the 16-byte phase of its CSR segment, rdc_prep_cl.cpp)
  auto copy_out = [&]() {
    for (int a = tid >> 5; a < (int)d.nown; a += NT / 32) {
      const HostPrepCl::Node nd = ntab[(size_t)w * MAXN + a];
      const int nn = NV * NV * (int)nd.len, l32 = tid & 31;
      double* dst = val + (int64_t)(NV * NV) * nd.bptr;
      const double* src = img + nd.off;
      const int sh = (int)(nd.off & 1), npair = (nn - sh) >> 1;
      const v2d_t* s2 = reinterpret_cast<const v2d_t*>(src + sh);
      v2d_t* d2 = reinterpret_cast<v2d_t*>(dst + sh);
      if (diag & 64) {
        for (int x = l32; x < nn; x += 32) __builtin_nontemporal_store(src[x], dst + x);
      } else {
        for (int x = l32; x < npair; x += 32) __builtin_nontemporal_store(s2[x], d2 + x);
        if (sh && l32 == 0) __builtin_nontemporal_store(src[0], dst);
        if (((nn - sh) & 1) && l32 == 1) __builtin_nontemporal_store(src[nn - 1], dst + nn - 1);
      }
      if (l32 < NV) rhs[(int64_t)NV * nd.node + l32] = lrhs[NV * a + l32];
    }
  };
  // workgroup barrier that orders LDS accesses only (no wait for the global stores in flight)
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // Two code paths with the SAME sequence of workgroup barriers (the branch is uniform per wave): the register allocator
  // never holds the consumers' accumulators and the producers' element at once.
  if (producer) {
    const int pl = tid - MAXP;
    const uint32_t e = eid[(size_t)w * MAXE + pl];
    const bool plive = e != 0xFFFFFFFFu;
    double X[8][3], U[8][NV], AX[8][NA];
    if (plive) {
#pragma unroll
      for (int n = 0; n < 8; n++) {
        const int64_t I = m.conn[(int64_t)e * 8 + n];
#pragma unroll
        for (int c = 0; c < 3; c++) X[n][c] = m.xyz[3 * I + c];
#pragma unroll
        for (int v = 0; v < NV; v++) U[n][v] = u[NV * I + v];
#pragma unroll
        for (int v = 0; v < NA; v++) AX[n][v] = (M::NAUX > 0 && (M::AUX_LOCAL_NODE < 0 || n == M::AUX_LOCAL_NODE)) ? aux[(int64_t)M::NAUX * I + (M::NAUX > 0 ? v : 0)] : 0.0;
      }
    }
    const double* ED = M::NELEM > 0 ? elem + (int64_t)e * M::NELEM : nullptr;
    // PPR points per round and barrier: buffer (round & 1) holds the records [point in round][element]
    if (plive) {
#pragma unroll
      for (int x = 0; x < PPR; x++) hex8_cl_produce<M>(k, X, U, AX, ED, x, lds + (x * MAXE + pl) * R::STRIDE);
    }
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < 8; q += PPR) {   // one round ahead of the consumers
      if (plive && q + PPR < 8) {
#pragma unroll
        for (int x = 0; x < PPR; x++)
          hex8_cl_produce<M>(k, X, U, AX, ED, q + PPR + x, lds + (((((q / PPR) + 1) & 1) * PPR + x) * MAXE + pl) * R::STRIDE);
      }
      __syncthreads();
    }
    zero_image();
    lds_barrier();
    lds_barrier();                  // consumers: atomics
    copy_out();
    return;
  }
  // ---- consumer: one (owned node, element) pair per lane -----------------------------------------------------------------
  double acc[NV][NV][8], fe[NV];
  rd_row_zero<M, 8>(acc, fe);
  int le = 0, li = 0, na = 0;
  const uint32_t pr = pair[(size_t)w * MAXP + tid];
  const bool cvalid = pr != 0xFFFFFFFFu;
  if (cvalid) { le = (int)(pr & 0xFF); li = (int)((pr >> 8) & 0xFF); na = (int)((pr >> 16) & 0xFF); }
  __syncthreads();                  // producers: the first round of points
#pragma unroll 1
  for (int q = 0; q < 8; q += PPR) {
    if (cvalid) {
#pragma unroll
      for (int x = 0; x < PPR; x++)
        hex8_cl_consume<M, EXP_MODE>(k, lds + ((((q / PPR) & 1) * PPR + x) * MAXE + le) * R::STRIDE, q + x, li, acc, fe);
    }
    __syncthreads();
  }
  zero_image();
  uint32_t sl0 = 0, sl1 = 0;
  int off = 0, lenv = 0;
  if (cvalid) {
    sl0 = pslot[((size_t)w * MAXP + tid) * 2];
    sl1 = pslot[((size_t)w * MAXP + tid) * 2 + 1];
    const HostPrepCl::Node nd = ntab[(size_t)w * MAXN + na];
    off = (int)nd.off;
    lenv = NV * (int)nd.len;
  }
  lds_barrier();
  if (cvalid) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int s = (int)(((j < 4 ? sl0 : sl1) >> (8 * (j & 3))) & 0xFF);
      double* p = img + off + NV * s;
#pragma unroll
      for (int a = 0; a < NV; a++)
#pragma unroll
        for (int b = 0; b < NV; b++)
          if (hex8_cl_block<M>(a, b))   // structurally zero blocks stay the zeros of the image
            __hip_atomic_fetch_add(p + a * lenv + b, acc[a][b][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int a = 0; a < NV; a++)
      __hip_atomic_fetch_add(lrhs + NV * na + a, fe[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  lds_barrier();
  copy_out();
}

// ---- persistent form ("hex_kernel" = 2) ------------------------------------------------------------------------------------------
// In k_hex8_cl the zero / atomics / copy-out epilogue of a cluster and the drain of its stores are serial inside the
// workgroup and only overlap with the other workgroup of the CU: for the cheap integrands they cost more than the
// arithmetic.  Measured on H(126) this form ties with k_hex8_cl (HCC 2.9 vs 2.8 ms: the consumer waves' own instruction
// stream -- copy, accumulate, atomics -- is the critical path either way), so it is not the default.  A workgroup walks over the clusters blockIdx.x, blockIdx.x + gridDim.x, ..., the image has its own LDS
// region, and the memory instructions are split by role so that no wave ever waits for a store:
//   * the PRODUCER wave issues every global LOAD: its elements (fetched for the next cluster when it has no point left to
//     produce) and the work lists of the next cluster, which it passes on through LDS (three list buffers: previous,
//     current, next cluster);
//   * the CONSUMER waves issue every global STORE: while they accumulate cluster n they copy the image of cluster n - 1
//     out (one node per wave and quadrature point, zeroing what they have read), then add their rows of cluster n into
//     the image and go straight on.  vmcnt counts loads and stores in order on gfx9, so a wave that loaded after it
//     stored would wait for its stores to land; these waves never load.
// Nine workgroup barriers per cluster (LDS ordering only), none for an epilogue.
template <int CW>
struct Hex8ClLists {   // one list buffer in LDS (32-bit words)
  static constexpr int PAIR = 0, SLOT = CW * 64, NTAB = 3 * CW * 64, NOWN = NTAB + 4 * CW * 8, WORDS = (NOWN + 4 + 3) & ~3;
};

template <class M, int EXP_MODE, int CW, int PW>
__global__ void __launch_bounds__((CW + PW) * 64, 2)
k_hex8_clp(const MeshDev m, const typename M::K k, const HostPrepCl::Desc* __restrict__ desc, const HostPrepCl::Node* __restrict__ ntab,
           const uint32_t* __restrict__ eid, const uint32_t* __restrict__ pair, const uint32_t* __restrict__ pslot,
           const double* __restrict__ u, const double* __restrict__ aux, const double* __restrict__ elem,
           double* __restrict__ val, double* __restrict__ rhs, const int n_wg, const int img_doubles,
           const int diag /* timing diagnostics ("ablate" option), bit mask: 1 = consumers idle, 2 = producers idle, 8 = no atomics, 16 = no copy-out */) {
  constexpr int NV = M::NV, NA = (M::NAUX > 0 ? M::NAUX : 1);
  constexpr int MAXP = CW * 64, MAXE = PW * 64, MAXN = CW * 8, NW = CW + PW;
  static_assert(PW == 1, "one producer wave");
  using R = Hex8Rec<M>;
  using L = Hex8ClLists<CW>;
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int PBUF = (2 * MAXE * R::STRIDE + 1) & ~1;
  double* const img = lds + PBUF;
  double* const lrhs = img + img_doubles;          // img_doubles is even
  uint32_t* const lists = reinterpret_cast<uint32_t*>(lrhs + ((NV * MAXN + 1) & ~1));
  int w = blockIdx.x;
  const int G = gridDim.x;
  // roles rotate over the waves so that the producers of the two workgroups of a CU sit on different SIMDs (wave i of a
  // workgroup runs on SIMD i; workgroups b and b + gridDim.x / 2 share a CU when two are resident per CU)
  const int tid = (int)(((threadIdx.x >> 6) + (((blockIdx.x >> 3) + (2 * blockIdx.x >= gridDim.x ? 1 : 0)) % NW)) % NW) * 64 + (int)(threadIdx.x & 63);
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // node a of the image of the cluster whose lists are in buffer b: copied out by one wave as runs of consecutive doubles, zeroed
  auto copy_node = [&](int b, int a, bool zero) {
    const uint32_t* lb = lists + b * L::WORDS;
    const int ln = tid & 63;
    if (a >= (int)lb[L::NOWN] || (diag & 16)) return;
    const uint32_t bptr = lb[L::NTAB + 4 * a], lo = lb[L::NTAB + 4 * a + 1], node = lb[L::NTAB + 4 * a + 2];
    const int nn = NV * NV * (int)(lo & 0xFFFF), off = (int)(lo >> 16);
    double* dst = val + (int64_t)(NV * NV) * bptr;
    double v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = (ln + 64 * i < nn) ? img[off + ln + 64 * i] : 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (ln + 64 * i < nn) {
        __builtin_nontemporal_store(v[i], dst + ln + 64 * i);
        if (zero) img[off + ln + 64 * i] = 0.0;
      }
    for (int x = ln + 256; x < nn; x += 64) {   // rows longer than 28 blocks (unstructured meshes)
      __builtin_nontemporal_store(img[off + x], dst + x);
      if (zero) img[off + x] = 0.0;
    }
    if (ln < NV) {
      rhs[(int64_t)NV * node + ln] = lrhs[NV * a + ln];
      if (zero) lrhs[NV * a + ln] = 0.0;
    }
  };
  // slice s of that image = nodes s, s + 8, s + 16: one per consumer wave (a single wave -- the producer -- copying all of
  // it was measured: 3.7 instead of 2.9 ms, one wave does not keep enough stores in flight)
  auto copy_slice = [&](int b, int s, bool zero) { copy_node(b, s + 8 * (tid >> 6), zero); };
  if (tid >= MAXP) {
    // ================= producer: every global load of the workgroup ===================================================================
    const int pl = tid - MAXP;
    bool plive = false;
    double X[8][3], U[8][NV], AX[8][NA];
    const double* ED = nullptr;
    auto load_element = [&](int ww) {
      const uint32_t e = eid[(size_t)ww * MAXE + pl];
      plive = e != 0xFFFFFFFFu;
      if (!plive) return;
#pragma unroll
      for (int n = 0; n < 8; n++) {
        const int64_t I = m.conn[(int64_t)e * 8 + n];
#pragma unroll
        for (int c = 0; c < 3; c++) X[n][c] = m.xyz[3 * I + c];
#pragma unroll
        for (int v = 0; v < NV; v++) U[n][v] = u[NV * I + v];
#pragma unroll
        for (int v = 0; v < NA; v++) AX[n][v] = (M::NAUX > 0 && (M::AUX_LOCAL_NODE < 0 || n == M::AUX_LOCAL_NODE)) ? aux[(int64_t)M::NAUX * I + (M::NAUX > 0 ? v : 0)] : 0.0;
      }
      ED = M::NELEM > 0 ? elem + (int64_t)e * M::NELEM : nullptr;
    };
    // work lists of cluster ww -> list buffer b
    auto stage_lists = [&](int ww, int b) {
      uint32_t* lb = lists + b * L::WORDS;
      const uint32_t* gp = pair + (size_t)ww * MAXP;
      const uint32_t* gs = pslot + (size_t)ww * MAXP * 2;
      const uint32_t* gn = reinterpret_cast<const uint32_t*>(ntab + (size_t)ww * MAXN);
#pragma unroll
      for (int i = 0; i < CW; i++) lb[L::PAIR + pl + 64 * i] = gp[pl + 64 * i];
#pragma unroll
      for (int i = 0; i < 2 * CW; i++) lb[L::SLOT + pl + 64 * i] = gs[pl + 64 * i];
      for (int x = pl; x < 4 * MAXN; x += 64) lb[L::NTAB + x] = gn[x];
      if (pl == 0) lb[L::NOWN] = desc[ww].nown;
    };
    {
      v2d_t* z = reinterpret_cast<v2d_t*>(img);
      const v2d_t zero = {0.0, 0.0};
      for (int x = pl; x < (img_doubles + NV * MAXN + 1) / 2; x += 64) z[x] = zero;
    }
    stage_lists(w, 0);
    load_element(w);
    int cb = 0;                             // list buffer of the current cluster
    for (;;) {
      if (plive && !(diag & 2)) hex8_cl_produce<M>(k, X, U, AX, ED, 0, lds + pl * R::STRIDE);
      lds_barrier();                        // point 0 and the lists are out; the consumers' atomics of the previous cluster are in the image
      const int nb = cb == 2 ? 0 : cb + 1;
#pragma unroll 1
      for (int q = 0; q < 8; q++) {         // one point ahead of the consumers
        if (q + 1 < 8) { if (plive && !(diag & 2)) hex8_cl_produce<M>(k, X, U, AX, ED, q + 1, lds + (((q + 1) & 1) * MAXE + pl) * R::STRIDE); }
        else if (w + G < n_wg) load_element(w + G);
        if (q == 0 && w + G < n_wg) stage_lists(w + G, nb);
        lds_barrier();
      }
      cb = nb;
      w += G;
      if (w >= n_wg) break;
    }
    lds_barrier();                          // the consumers' atomics of the last cluster
    return;
  }
  // ================= consumers: every global store of the workgroup ====================================================================
  double acc[NV][NV][8], fe[NV];
  int cb = 0, pb = -1;                      // list buffers of the current / previous cluster
  for (;;) {
    rd_row_zero<M, 8>(acc, fe);
    lds_barrier();                          // producer: point 0 and the lists of this cluster
    const uint32_t* lb = lists + cb * L::WORDS;
    const uint32_t pr = lb[L::PAIR + tid];
    const bool cvalid = pr != 0xFFFFFFFFu;
    int le = 0, li = 0, na = 0;
    if (cvalid) { le = (int)(pr & 0xFF); li = (int)((pr >> 8) & 0xFF); na = (int)((pr >> 16) & 0xFF); }
#pragma unroll 1
    for (int q = 0; q < 8; q++) {
      if (pb >= 0) copy_slice(pb, q, true);
      if (cvalid && !(diag & 1)) hex8_cl_consume<M, EXP_MODE>(k, lds + ((q & 1) * MAXE + le) * R::STRIDE, q, li, acc, fe);
      lds_barrier();
    }
    // the previous cluster's image has been copied out and zeroed during these eight rounds
    if (cvalid && !(diag & 8)) {
      const uint32_t sl0 = lb[L::SLOT + 2 * tid], sl1 = lb[L::SLOT + 2 * tid + 1], lo = lb[L::NTAB + 4 * na + 1];
      const int off = (int)(lo >> 16), lenv = NV * (int)(lo & 0xFFFF);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int s = (int)(((j < 4 ? sl0 : sl1) >> (8 * (j & 3))) & 0xFF);
        double* p = img + off + NV * s;
#pragma unroll
        for (int a = 0; a < NV; a++)
#pragma unroll
          for (int b = 0; b < NV; b++)
            if (hex8_cl_block<M>(a, b))
              __hip_atomic_fetch_add(p + a * lenv + b, acc[a][b][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
#pragma unroll
      for (int a = 0; a < NV; a++)
        __hip_atomic_fetch_add(lrhs + NV * na + a, fe[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    pb = cb;
    cb = cb == 2 ? 0 : cb + 1;
    w += G;
    if (w >= n_wg) break;
  }
  lds_barrier();
  for (int q = 0; q < 8; q++) copy_slice(pb, q, false);
}

template <class M>
inline size_t hex8_clp_lds_bytes(int cw, int pw, size_t max_row_doubles) {
  const size_t pbuf = ((size_t)2 * pw * 64 * Hex8Rec<M>::STRIDE + 1) & ~(size_t)1;
  const size_t image = ((max_row_doubles + 1) & ~(size_t)1) + (((size_t)M::NV * cw * 8 + 1) & ~(size_t)1);
  return sizeof(double) * (pbuf + image) + sizeof(uint32_t) * 3 * Hex8ClLists<3>::WORDS;
}

template <class M>
inline size_t hex8_cl_lds_bytes(int cw, int pw, size_t max_row_doubles, int ppr) {
  const size_t points = (size_t)2 * ppr * pw * 64 * Hex8Rec<M>::STRIDE;
  const size_t image = ((max_row_doubles + 1) & ~(size_t)1) + (size_t)M::NV * cw * 8;
  return sizeof(double) * (points > image ? points : image);
}

template <class M, int EXP_MODE>
static hipError_t launch_hex8_cl(const LaunchArgs& a, const typename M::K& k) {
  constexpr int CW = 3, PW = 1;
  if (a.cl.cw != CW || a.cl.pw != PW) return hipErrorInvalidValue;
  const size_t pbytes = hex8_clp_lds_bytes<M>(CW, PW, a.cl.max_row_doubles);
  if (a.cl.grid > 0 && pbytes <= 80 * 1024) {   // persistent form: two workgroups per CU must fit
    static std::atomic<uint64_t> pattr[1];  /* per instantiation and device */
    dyn_lds_once(pattr[0], (const void*)k_hex8_clp<M, EXP_MODE, CW, PW>, 80 * 1024);
    const int grid = a.cl.grid < a.cl.n_wg ? a.cl.grid : a.cl.n_wg;
    hipLaunchKernelGGL((k_hex8_clp<M, EXP_MODE, CW, PW>), dim3(grid), dim3((CW + PW) * 64), pbytes, a.stream, a.m, k, a.cl.desc, a.cl.ntab,
                       a.cl.eid, a.cl.pair, a.cl.pslot, a.u, a.aux, a.elem, a.val, a.rhs, a.cl.n_wg, (int)((a.cl.max_row_doubles + 1) & ~(size_t)1), a.opt_ablate);
    return hipGetLastError();
  }
  // quadrature points per round and workgroup barrier: the model's choice (M::HEX_CL_POINTS); "prefetch" = 1 forces one
#define RDC_HEX8_CL(PPR)                                                                                                              \
  {                                                                                                                                   \
    const size_t bytes = hex8_cl_lds_bytes<M>(CW, PW, a.cl.max_row_doubles, PPR);                                                     \
    static std::atomic<uint64_t> attr[1];  /* per instantiation and device */ \
    dyn_lds_once(attr[0], (const void*)k_hex8_cl<M, EXP_MODE, CW, PW, PPR>, 80 * 1024); \
    hipLaunchKernelGGL((k_hex8_cl<M, EXP_MODE, CW, PW, PPR>), dim3(a.cl.n_wg), dim3((CW + PW) * 64), bytes, a.stream, a.m, k, a.cl.desc, a.cl.ntab, \
                       a.cl.eid, a.cl.pair, a.cl.pslot, a.u, a.aux, a.elem, a.val, a.rhs, a.opt_ablate);                             \
  }
  if constexpr (M::HEX_CL_POINTS == 2) { if (a.opt_pf == 1) RDC_HEX8_CL(1) else RDC_HEX8_CL(2) }
  else RDC_HEX8_CL(1)
#undef RDC_HEX8_CL
  return hipGetLastError();
}

// ---- five unknowns: one equation row at a time ----------------------------------------------------------------------------
// The 5 x 5 x 8 accumulator of a pair does not fit the register file (the pair kernels evaluate such models one equation row
// at a time too, redoing the whole per-point set-up five times per pair).  Here the workgroup makes NV passes over the
// quadrature points: in pass A the producer hands out the point records again (cheap next to the consumers' work), the
// consumers accumulate row A only (NV x 8 accumulators), add it into an LDS image of equation row A of the cluster's nodes
// and the image leaves as one run of NV * len doubles per node.  The image overlays the point buffers.
template <class M, int EXP_MODE, int CW, int PW>
__global__ void __launch_bounds__((CW + PW) * 64, 2)
k_hex8_cl_rows(const MeshDev m, const typename M::K k, const HostPrepCl::Desc* __restrict__ desc, const HostPrepCl::Node* __restrict__ ntab,
               const uint32_t* __restrict__ eid, const uint32_t* __restrict__ pair, const uint32_t* __restrict__ pslot,
               const double* __restrict__ u, const double* __restrict__ aux, const double* __restrict__ elem,
               double* __restrict__ val, double* __restrict__ rhs) {
  constexpr int NV = M::NV, NA = (M::NAUX > 0 ? M::NAUX : 1);
  constexpr int MAXP = CW * 64, MAXE = PW * 64, MAXN = CW * 8, NT = (CW + PW) * 64, NW = CW + PW;
  using R = Hex8Rec<M>;
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int w = blockIdx.x;
  const int tid = (int)(((threadIdx.x >> 6) + ((blockIdx.x >> 3) % NW)) % NW) * 64 + (int)(threadIdx.x & 63);
  const HostPrepCl::Desc d = desc[w];
  const int nimg = (int)d.row_doubles;            // image of ONE equation row: sum of NV * len
  double* const img = lds;
  double* const lrhs = lds + ((nimg + 1) & ~1);   // one rhs entry per owned node and pass
  auto zero_image = [&]() {
    v2d_t* z = reinterpret_cast<v2d_t*>(lds);
    const v2d_t zero = {0.0, 0.0};
    for (int x = tid; x < (((nimg + 1) & ~1) + (int)d.nown + 1) / 2; x += NT) z[x] = zero;
  };
  // a half-wave per node: equation row A of the node is NV * len consecutive doubles of the CSR array
  auto copy_out = [&](int A) {
    for (int a = tid >> 5; a < (int)d.nown; a += NT / 32) {
      const HostPrepCl::Node nd = ntab[(size_t)w * MAXN + a];
      const int nn = NV * (int)nd.len;
      double* dst = val + (int64_t)(NV * NV) * nd.bptr + (int64_t)A * nn;
      for (int x = tid & 31; x < nn; x += 32) __builtin_nontemporal_store(img[nd.off + x], dst + x);
      if ((tid & 31) == 0) rhs[(int64_t)NV * nd.node + A] = lrhs[a];
    }
  };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  if (tid >= MAXP) {
    // ---- producer --------------------------------------------------------------------------------------------------------------
    const int pl = tid - MAXP;
    const uint32_t e = eid[(size_t)w * MAXE + pl];
    const bool plive = e != 0xFFFFFFFFu;
    double X[8][3], U[8][NV], AX[8][NA];
    if (plive) {
#pragma unroll
      for (int n = 0; n < 8; n++) {
        const int64_t I = m.conn[(int64_t)e * 8 + n];
#pragma unroll
        for (int c = 0; c < 3; c++) X[n][c] = m.xyz[3 * I + c];
#pragma unroll
        for (int v = 0; v < NV; v++) U[n][v] = u[NV * I + v];
#pragma unroll
        for (int v = 0; v < NA; v++) AX[n][v] = (M::NAUX > 0 && (M::AUX_LOCAL_NODE < 0 || n == M::AUX_LOCAL_NODE)) ? aux[(int64_t)M::NAUX * I + (M::NAUX > 0 ? v : 0)] : 0.0;
      }
    }
    const double* ED = M::NELEM > 0 ? elem + (int64_t)e * M::NELEM : nullptr;
#pragma unroll 1
    for (int A = 0; A < NV; A++) {
      if (plive) hex8_cl_produce<M>(k, X, U, AX, ED, 0, lds + pl * R::STRIDE);
      __syncthreads();
#pragma unroll 1
      for (int q = 0; q < 8; q++) {
        if (plive && q + 1 < 8) hex8_cl_produce<M>(k, X, U, AX, ED, q + 1, lds + (((q + 1) & 1) * MAXE + pl) * R::STRIDE);
        __syncthreads();
      }
      zero_image();
      lds_barrier();
      lds_barrier();                // consumers: atomics of row A
      copy_out(A);
      lds_barrier();                // the image has been read: the next pass may overwrite it
    }
    return;
  }
  // ---- consumers ------------------------------------------------------------------------------------------------------------------
  int le = 0, li = 0, na = 0;
  const uint32_t pr = pair[(size_t)w * MAXP + tid];
  const bool cvalid = pr != 0xFFFFFFFFu;
  if (cvalid) { le = (int)(pr & 0xFF); li = (int)((pr >> 8) & 0xFF); na = (int)((pr >> 16) & 0xFF); }
  uint32_t sl0 = 0, sl1 = 0;
  int off = 0;
  if (cvalid) {
    sl0 = pslot[((size_t)w * MAXP + tid) * 2];
    sl1 = pslot[((size_t)w * MAXP + tid) * 2 + 1];
    off = (int)ntab[(size_t)w * MAXN + na].off;
  }
  auto pass = [&](auto tagA) {
    constexpr int A = decltype(tagA)::value;
    double acc[NV][8], fe = 0.0;
#pragma unroll
    for (int b = 0; b < NV; b++)
#pragma unroll
      for (int j = 0; j < 8; j++) acc[b][j] = 0.0;
    __syncthreads();                // producers: point 0
#pragma unroll 1
    for (int q = 0; q < 8; q++) {
      if (cvalid) hex8_cl_consume_row<M, EXP_MODE, A>(k, lds + ((q & 1) * MAXE + le) * R::STRIDE, q, li, acc, fe);
      __syncthreads();
    }
    zero_image();
    lds_barrier();
    if (cvalid) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int s = (int)(((j < 4 ? sl0 : sl1) >> (8 * (j & 3))) & 0xFF);
#pragma unroll
        for (int b = 0; b < NV; b++)
          if (hex8_cl_block<M>(A, b))   // structurally zero blocks stay the zeros of the image
            __hip_atomic_fetch_add(img + off + NV * s + b, acc[b][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __hip_atomic_fetch_add(lrhs + na, fe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    lds_barrier();
    copy_out(A);
    lds_barrier();
  };
  pass(std::integral_constant<int, 0>{});
  pass(std::integral_constant<int, 1>{});
  pass(std::integral_constant<int, 2>{});
  if constexpr (NV > 3) pass(std::integral_constant<int, (NV > 3 ? 3 : 0)>{});
  if constexpr (NV > 4) pass(std::integral_constant<int, (NV > 4 ? 4 : 0)>{});
}

template <class M, int EXP_MODE>
static hipError_t launch_hex8_cl_rows(const LaunchArgs& a, const typename M::K& k) {
  constexpr int CW = 3, PW = 1;
  if (a.cl.cw != CW || a.cl.pw != PW) return hipErrorInvalidValue;
  const size_t points = (size_t)2 * PW * 64 * Hex8Rec<M>::STRIDE;
  const size_t image = ((a.cl.max_row_doubles + 1) & ~(size_t)1) + (size_t)CW * 8 + 2;
  const size_t bytes = sizeof(double) * (points > image ? points : image);
  static std::atomic<uint64_t> attr[1];  /* per instantiation and device */
    dyn_lds_once(attr[0], (const void*)k_hex8_cl_rows<M, EXP_MODE, CW, PW>, 80 * 1024);
  hipLaunchKernelGGL((k_hex8_cl_rows<M, EXP_MODE, CW, PW>), dim3(a.cl.n_wg), dim3((CW + PW) * 64), bytes, a.stream, a.m, k, a.cl.desc, a.cl.ntab,
                     a.cl.eid, a.cl.pair, a.cl.pslot, a.u, a.aux, a.elem, a.val, a.rhs);
  return hipGetLastError();
}

}  // namespace rdc
#endif
