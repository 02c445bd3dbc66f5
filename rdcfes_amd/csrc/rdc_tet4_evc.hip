// rdc_tet4_evc.hip — element-visit kernel in coefficient form for three-unknown TET4 models; instantiated for Ripf (all terms
// on), the one model whose per-element part is heavy enough to win against the pair kernel (EvcEligible, rdc_internal.h).  Formulation: rdc_tet4_evc.h; lists: rdc_prep_ev.cpp (the lists of k_tet4_ev).
//
// One workgroup = one CLUSTER of <= 16 owned nodes and the <= 256 elements touching it.
//   phase 0  zero the entry slice, load the lists, LDS-DMA the node records of the cluster's closure
//   phase 1  one thread per element visit: tet4_visit() -> ds_add_f64 of the entries of the rows it owns; slice layout
//            [entry of the block][slot][node] (bank = row node: the host placed the visits so that the 16 lanes of an LDS
//            pass have different row nodes)
//   phase 2  one thread per node block: its entries -> LDS image of the CSR rows of the cluster's nodes (zeros for the
//            structurally zero blocks); rhs entries straight to memory
//   phase 3  the image leaves with 16-byte non-temporal stores, one contiguous CSR segment per node
// Every CSR value is written exactly once; no global atomics, no colours.  Sums are order-dependent in the last bits.
#include "rdc_internal.h"
#include "rdc_tet4_ev.h"
#include "rdc_tet4_evc.h"

namespace rdc {

template <class M> struct EvcRec {   // node record of k_pack_nodes<M> (rdc_tet4_fast.hip): xyz | u | aux | pad
  static constexpr int RAW = 3 + M::NV + M::NAUX;
  static constexpr int N = (RAW + 1) & ~1;
};

template <class M>
struct EvcSink {
  double* p[4][4];   // LDS address of entry 0 of block (node i, node j)
  double* pr[4];     // LDS address of rhs entry 0 of node i
  __device__ __forceinline__ void ke(int a, int b, int i, int j, double v) {
    __hip_atomic_fetch_add(p[i][j] + evc_index<M>(a, b) * ev::NBP, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ void fe(int a, int i, double v) {
    __hip_atomic_fetch_add(pr[i] + a * ev::MAXN, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
};

template <class M, int EXP_MODE, int MINW>
__global__ void __launch_bounds__(256, MINW)
k_tet4_evc(const HostPrepEv::Desc* __restrict__ desc, const uint32_t* __restrict__ nlist, const uint32_t* __restrict__ vloc,
           const uint32_t* __restrict__ vslot, const HostPrepEv::Node* __restrict__ ntab, const uint32_t* __restrict__ wg_perm,
           const typename M::K k, const double* __restrict__ rec, double* __restrict__ val, double* __restrict__ rhs,
           const int nls, const int wg_begin) {
  constexpr int BLOCK = 256, NV = M::NV, NA = (M::NAUX > 0 ? M::NAUX : 1), NP = EvcRec<M>::N / 2, NE = evc_blocks<M>();
  constexpr int NBP = ev::NBP, MAXN = ev::MAXN;
  extern __shared__ __attribute__((aligned(16))) double lds[];   // [S: NE x NBP | R: NV x MAXN | records: NP x nls x 16 B]
  __shared__ HostPrepEv::Node snode[MAXN];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  int w = (int)blockIdx.x + wg_begin;
  if (wg_perm) w = (int)wg_perm[w];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double* const R = lds + NE * NBP;
  double* const recs = R + ((NV * MAXN + 1) & ~1);
  // phase 0: zero [S | R] (16-byte stores), list loads, LDS-DMA of the node records
  {
    double2* z = reinterpret_cast<double2*>(lds);
    for (int x = tid; x < (NE * NBP + ((NV * MAXN + 1) & ~1)) / 2; x += BLOCK) z[x] = make_double2(0.0, 0.0);
  }
  const int rounds = nls >> 6;
  uint32_t nid = 0;
  if (wv < rounds) nid = nlist[(size_t)w * nls + wv * 64 + lane];
  const uint32_t pl = vloc[(size_t)w * BLOCK + tid];
  const uint2 sl = reinterpret_cast<const uint2*>(vslot)[(size_t)w * BLOCK + tid];
  const HostPrepEv::Desc d = desc[w];
  if (tid < MAXN) snode[tid] = ntab[(size_t)w * MAXN + tid];
  if (wv < rounds) {
    const char* src = reinterpret_cast<const char*>(rec) + (size_t)nid * (NP * 16);
#pragma unroll
    for (int p = 0; p < NP; p++)
      __builtin_amdgcn_global_load_lds((glb_ptr)(src + p * 16), (lds_ptr)(recs + (p * nls + wv * 64) * 2), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // phase 1: element visits
  if (pl != 0xFFFFFFFFu) {
    double X[4][3], U[4][NV], AX[4][NA];
    int li[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      li[j] = (pl >> (8 * j)) & 0xFF;
      double rr[2 * NP];
#pragma unroll
      for (int p = 0; p < NP; p++) {
        const double2 v2 = reinterpret_cast<const double2*>(recs)[p * nls + li[j]];
        rr[2 * p] = v2.x; rr[2 * p + 1] = v2.y;
      }
      X[j][0] = rr[0]; X[j][1] = rr[1]; X[j][2] = rr[2];
#pragma unroll
      for (int v = 0; v < NV; v++) U[j][v] = rr[3 + v];
#pragma unroll
      for (int v = 0; v < NA; v++) AX[j][v] = M::NAUX > 0 ? rr[(3 + NV + v) % (2 * NP)] : 0.0;
    }
    const int nown = (int)d.nown;
    const int r = (li[0] < nown) + (li[1] < nown) + (li[2] < nown) + (li[3] < nown);   // the owned vertices come first
    EvcSink<M> sink;
    const uint32_t sw[4] = {sl.x & 0xFFFFu, sl.x >> 16, sl.y & 0xFFFFu, sl.y >> 16};   // four 4-bit column slots per row
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int a = (i < r) ? li[i] : 0;   // list position of an owned vertex == its cluster index
      sink.pr[i] = R + a;
#pragma unroll
      for (int j = 0; j < 4; j++) sink.p[i][j] = lds + a + MAXN * (int)((sw[i] >> (4 * j)) & 0xF);   // block (a, slot): slot * 16 + a
    }
    tet4_visit<M, EXP_MODE>(k, X, U, AX, r, sink);
  }
  __syncthreads();
  // phase 2: node block tid = slot * 16 + node
  double e[NE];
  const int bn = tid & (MAXN - 1), bs = tid >> 4;
  const bool has = bn < (int)d.nown && bs < (int)snode[bn < (int)d.nown ? bn : 0].len;
  if (has) {
#pragma unroll
    for (int m = 0; m < NE; m++) e[m] = lds[m * NBP + tid];
  }
  if (tid < (int)d.nown * NV) {   // rhs: R[a][node] -> rhs[node * NV + a]
    const int n = tid / NV, a = tid - n * NV;
    rhs[(size_t)snode[n].node * NV + a] = R[a * MAXN + n];
  }
  __syncthreads();   // every entry has been read: the image may overwrite the slice
  if (has) {
    const int lenv = NV * (int)snode[bn].len;
    double* dst = lds + snode[bn].obase + NV * bs;
#pragma unroll
    for (int a = 0; a < NV; a++)
#pragma unroll
      for (int b = 0; b < NV; b++) dst[a * lenv + b] = evc_block<M>(a, b) ? e[evc_index<M>(a, b) % NE] : 0.0;
  }
  __syncthreads();
  // phase 3: one contiguous CSR segment per node; the image has the 16-byte phase of its segment in memory
  for (int n = wv; n < (int)d.nown; n += 4) {
    const HostPrepEv::Node nd = snode[n];
    const int cnt = NV * NV * (int)nd.len;
    double* out = val + (size_t)(NV * NV) * nd.bptr;     // out[x] <-> img[x]
    const double* img = lds + nd.obase;
    const int sh = (int)(nd.obase & 1);                  // == (NV^2 * bptr) & 1 by construction
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    const int npair = (cnt - sh) >> 1;
    const v2d_t* src = reinterpret_cast<const v2d_t*>(img + sh);
    v2d_t* dstg = reinterpret_cast<v2d_t*>(out + sh);
    for (int x = lane; x < npair; x += 64) __builtin_nontemporal_store(src[x], dstg + x);
    if (sh && lane == 0) __builtin_nontemporal_store(img[0], out);
    if (((cnt - sh) & 1) && lane == 1) __builtin_nontemporal_store(img[cnt - 1], out + cnt - 1);
  }
}

// the node records have been packed by the caller (launch_fast_impl, rdc_tet4_fast.hip)
template <class M>
hipError_t launch_tet4_evc(const LaunchArgs& a, const typename M::K& k) {
  const EvDev& E = a.ev;
  const int wg_count = E.wg_count < 0 ? E.n_wg - E.wg_begin : E.wg_count;
  if (wg_count <= 0) return hipSuccess;
  constexpr int NE = evc_blocks<M>(), NP = EvcRec<M>::N / 2;
  const size_t acc = (size_t)NE * ev::NBP + ((M::NV * ev::MAXN + 1) & ~1) + (size_t)NP * E.nls * 2;
  const size_t lds_bytes = sizeof(double) * (acc > E.max_out_doubles ? acc : E.max_out_doubles);
#define RDC_EVC(MODE, MINW)                                                                                                \
  hipLaunchKernelGGL((k_tet4_evc<M, MODE, MINW>), dim3(wg_count), dim3(256), lds_bytes, a.stream, E.desc, E.nlist, E.vloc, \
                     E.vslot, E.ntab, E.wg_perm, k, a.packed, a.val, a.rhs, E.nls, E.wg_begin)
  if (a.opt_evc_occ == 3) { if (a.exp_mode == M::FAST_EXP_MODE) RDC_EVC(M::FAST_EXP_MODE, 3); else RDC_EVC(0, 3); }
  else { if (a.exp_mode == M::FAST_EXP_MODE) RDC_EVC(M::FAST_EXP_MODE, 2); else RDC_EVC(0, 2); }
#undef RDC_EVC
  return hipGetLastError();
}

template hipError_t launch_tet4_evc<Ripf>(const LaunchArgs&, const Ripf::K&);

}  // namespace rdc
