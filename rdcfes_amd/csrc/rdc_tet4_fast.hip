// rdc_tet4_fast.hip — TET4-specialised kernels (PIHNA / RIPF / HCC) built on tet4_row0.
//
//   k_pack_nodes        (xyz | old solution | aux) -> one 16-byte-aligned record per node, so a
//                       node gather is one contiguous 48-80 B read instead of three scattered ones
//   k_tet4_rowgather    row-owner gather (see rdc_kernels.h::k_rowgather) with the factored row
//   k_tet4_coloured     coloured read-modify-write scatter with the factored row
//
// Replaces the element loop src/pihna.C:383-756 (src/ripf.C:410-671, src/coupled_hcc.C:463-646).
#include "rdc_internal.h"
#include "rdc_tet4_fast.h"
#include "rdc_tet4_pihna_moments.h"

#include <type_traits>

namespace rdc {

template <class M> struct Rec {
  static constexpr int NA = M::NAUX;
  static constexpr int RAW = 3 + M::NV + NA;
  static constexpr int N = (RAW + 1) & ~1;  // doubles per node record, even => 16-byte aligned
};

// records of nodes [node_begin, node_end)
template <class M>
__global__ void k_pack_nodes(int64_t node_begin, int64_t node_end, const double* __restrict__ xyz, const double* __restrict__ u,
                             const double* __restrict__ aux, double* __restrict__ rec, int* __restrict__ ticket) {
  constexpr int N = Rec<M>::N, NV = M::NV, NA = M::NAUX;
  if (ticket && blockIdx.x == 0 && threadIdx.x == 0) *ticket = 0;   // cluster counter of the resident element-visit kernel that follows on the stream
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = node_begin * N + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < node_end * N; t += stride) {
    const int64_t n = t / N;
    const int c = (int)(t - n * N);
    double v = 0.0;
    if (c < 3) v = xyz[3 * n + c];
    else if (c < 3 + NV) v = u[n * NV + (c - 3)];
    else if (c < 3 + NV + NA) v = aux[n * (NA > 0 ? NA : 1) + (c - 3 - NV)];
    rec[t] = v;
  }
}

template <class M>
__device__ __forceinline__ void load_rec(const double* __restrict__ rec, int64_t n, double (&X)[3], double (&U)[M::NV],
                                         double (&A)[M::NAUX > 0 ? M::NAUX : 1]) {
  constexpr int N = Rec<M>::N, NV = M::NV, NA = M::NAUX;
  const double2* p = reinterpret_cast<const double2*>(rec + n * N);
  double r[N];
#pragma unroll
  for (int x = 0; x < N / 2; x++) { const double2 v = p[x]; r[2 * x] = v.x; r[2 * x + 1] = v.y; }
  X[0] = r[0]; X[1] = r[1]; X[2] = r[2];
#pragma unroll
  for (int v = 0; v < NV; v++) U[v] = r[3 + v];
  if (NA > 0) {
#pragma unroll
    for (int v = 0; v < NA; v++) A[v] = r[3 + NV + v];
  } else {
    A[0] = 0.0;
  }
}

template <class M> RDC_HD constexpr bool block_nonzero(int a, int b) {
  bool any = M::hasA(a, b) || M::hasD(a, b);
  for (int g = 0; g < M::NG; g++) any = any || M::hasB(a, b, g);
  return any;
}

// ---- row gather -----------------------------------------------------------------------------
// ABL (tuning ablations, results are WRONG for ABL != 0): 1 = contributions summed into a register
// instead of LDS atomics, 2 = additionally skip the row evaluation (loads + zero + flush only)
template <class M, int ABL>
struct LdsSink {
  double* row;       // LDS row slice of the owner node
  double* lrhs;      // LDS rhs entries of the owner node
  int stride;        // NV * len   (doubles between equation rows)
  int off[4];        // NV * slot of the rotated column j
  double dummy;
  __device__ __forceinline__ void ke(int a, int b, int j, double v) {
    if (!block_nonzero<M>(a, b)) return;  // LDS slice is pre-zeroed
    if (ABL == 0)
      __hip_atomic_fetch_add(row + a * stride + off[j] + b, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      dummy += v;
  }
  __device__ __forceinline__ void fe(int a, double v) {
    if (ABL == 0) __hip_atomic_fetch_add(lrhs + a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else dummy += v;
  }
};

template <class M, int EXP_MODE, int BLOCK, int MINW, int ABL>
__global__ void __launch_bounds__(BLOCK, MINW)
k_tet4_rowgather(const MeshDev m, const typename M::K k, const double* __restrict__ rec, double* __restrict__ val,
                 double* __restrict__ rhs) {
  constexpr int NV = M::NV, NW = BLOCK / 64;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int w = blockIdx.x;
  const int64_t n0 = m.wg_node_ptr[w], n1 = m.wg_node_ptr[w + 1];
  const int64_t bb0 = m.bptr[n0];
  const int64_t vb0 = (int64_t)NV * NV * bb0;
  const int nval = (int)((int64_t)NV * NV * m.bptr[n1] - vb0);
  const int nrhs = (int)(n1 - n0) * NV;
  double* lrhs = lds + nval;
  for (int x = threadIdx.x; x < nval + nrhs; x += BLOCK) lds[x] = 0.0;
  __syncthreads();
  const int64_t p0 = m.node_pair_ptr[n0];
  const int npairs = (int)(m.node_pair_ptr[n1] - p0);
  // lane l of wave v takes pair l*NW + v: the ~24 pairs of one node are spread over all waves,
  // which divides the same-address multiplicity of the LDS adds by NW
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int base = 0; base < npairs; base += BLOCK) {
    const int idx = base + lane * NW + wv;
    if (idx >= npairs) continue;
    const int64_t p = p0 + idx;
    const int64_t e = m.pair_elem[p];
    const int i = m.pair_local[p];
    double X[4][3], U[4][NV], AX[4][M::NAUX > 0 ? M::NAUX : 1];
    int64_t I = 0;
    LdsSink<M, ABL> sink;
    sink.dummy = 0.0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int jo = j ^ i;  // original local index of rotated column j
      const int64_t n = m.conn[e * 4 + jo];
      if (j == 0) I = n;
      load_rec<M>(rec, n, X[j], U[j], AX[j]);
      sink.off[j] = NV * (int)m.eslot[e * 16 + i * 4 + jo];
    }
    const int64_t b0 = m.bptr[I];
    sink.stride = NV * (int)(m.bptr[I + 1] - b0);
    sink.row = lds + (int)(NV * NV * (b0 - bb0));
    sink.lrhs = lrhs + (int)(I - n0) * NV;
    if (ABL < 2) tet4_row0<M, EXP_MODE>(k, X, U, AX, sink);
    else sink.dummy = X[0][0] + X[1][1] + X[2][2] + X[3][0] + U[0][0] + U[1][1] + U[2][2] + U[3][NV - 1] + sink.off[0] +
                      sink.off[1] + sink.off[2] + sink.off[3];
    if (ABL != 0) sink.row[sink.stride + (threadIdx.x & 3)] = sink.dummy;  // keep the work alive
  }
  __syncthreads();
  // every value of this slice leaves the chip exactly once: streaming (non-temporal) stores
  double* out = val + vb0;
  for (int x = threadIdx.x; x < nval; x += BLOCK) __builtin_nontemporal_store(lds[x], out + x);
  double* orhs = rhs + n0 * NV;
  for (int x = threadIdx.x; x < nrhs; x += BLOCK) orhs[x] = lrhs[x];
}

// ---- row gather, flat work lists ------------------------------------------------------------
// Same algorithm as k_tet4_rowgather, but everything a thread needs comes from one workgroup
// descriptor and one 32-byte pair record that are addressed by blockIdx / threadIdx alone, so the
// dependent load chain is two levels deep (descriptor | pair record -> node records) instead of
// five (wg_node_ptr -> bptr / node_pair_ptr -> pair_elem -> conn / eslot -> records, bptr[I]).
// LDS sink of k_tet4_rg3.  Off-diagonal blocks: atomic adds into the row slice.  The diagonal
// block and the rhs of a node receive one contribution from EVERY pair of the node, i.e. the ~6
// lanes of a wave that share the node would hit the same address -- measured at 51 CU-cycles per
// wave-instruction against 6.3 for distinct consecutive addresses (tools/lds_atomic_bench.hip).
// They therefore go to private accumulators laid out [value][node * COPIES + copy]: within one
// wave-instruction every lane has its own (consecutive) address.
template <class M, int ABL_ = 0>
struct LdsSink3 {
  static constexpr int ABL = ABL_;  // diagnostic builds only: 1 = plain stores instead of atomics, 2 = no LDS traffic
  double sink = 0.0;
  double* row;     // row slice of the owner node
  double* dacc;    // private diagonal accumulators of this lane: dacc[v * NS], NS a compile-time constant
  static constexpr int ns = HostPrep::RG3_DIAG_SLOTS;
  int stride;      // NV * len
  int off[4];      // NV * slot of rotated column j (off[0] unused)
  __device__ __forceinline__ void ke(int a, int b, int j, double v) {
    if (!block_nonzero<M>(a, b)) return;  // accumulators are pre-zeroed
    double* p = (j == 0) ? dacc + (a * M::NV + b) * ns : row + a * stride + off[j] + b;
    if (ABL == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (ABL == 1) *p = v;
    else sink += v;
  }
  __device__ __forceinline__ void fe(int a, double v) {
    double* p = dacc + (M::NV * M::NV + a) * ns;
    if (ABL == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (ABL == 1) *p = v;
    else sink += v;
  }
};

template <class M, int EXP_MODE, int BLOCK, int MINW, bool STAMP = false>
__global__ void __launch_bounds__(BLOCK, MINW)
k_tet4_rg3(const HostPrep::WgDesc* __restrict__ desc, const uint32_t* __restrict__ pair_rec,
           const uint16_t* __restrict__ pair_aux, const uint16_t* __restrict__ node_tab, const typename M::K k,
           const double* __restrict__ rec, double* __restrict__ val, double* __restrict__ rhs, const int xcd_remap,
           long long* __restrict__ stamps) {
  constexpr int NV = M::NV, NW = BLOCK / 64, NC = HostPrep::rg3_diag_copies(BLOCK), NDV = NV * NV + NV;
  constexpr int ns = HostPrep::RG3_DIAG_SLOTS, MAXN = ns / NC;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ uint2 ntab[MAXN];  // node_tab entries of the workgroup's nodes (used in the fold phase)
  long long ts[6];
  if (STAMP) ts[0] = __builtin_amdgcn_s_memtime();
  // workgroups are dealt round-robin to the 8 XCDs (observed, used for speed only): give each XCD a
  // contiguous range of the node ordering so neighbouring workgroups share node records in ONE L2
  // (bijection for any grid size: XCD x owns q + (x < r) consecutive workgroups)
  int w = blockIdx.x;
  if (xcd_remap) {
    const int q = gridDim.x >> 3, r = gridDim.x & 7, x = blockIdx.x & 7;
    w = x * q + (x < r ? x : r) + (blockIdx.x >> 3);
  }
  // lane l of wave v takes pair l*NW + v (spreads the pairs of one node over the waves)
  const int idx = (threadIdx.x & 63) * NW + (threadIdx.x >> 6);
  const uint4 pr = reinterpret_cast<const uint4*>(pair_rec)[(int64_t)w * BLOCK + idx];
  const uint4 ax = reinterpret_cast<const uint4*>(pair_aux)[(int64_t)w * BLOCK + idx];
  const HostPrep::WgDesc d = desc[w];
  const bool valid = pr.x != 0xFFFFFFFFu;
  double X[4][3], U[4][NV], AX[4][M::NAUX > 0 ? M::NAUX : 1];
  if (valid) {
    load_rec<M>(rec, pr.x, X[0], U[0], AX[0]);
    load_rec<M>(rec, pr.y, X[1], U[1], AX[1]);
    load_rec<M>(rec, pr.z, X[2], U[2], AX[2]);
    load_rec<M>(rec, pr.w, X[3], U[3], AX[3]);
  }
  if ((int)threadIdx.x < d.nnodes) ntab[threadIdx.x] = reinterpret_cast<const uint2*>(node_tab)[d.n0 + threadIdx.x];
  // zero the accumulators while the loads are in flight: [row slice | private diagonals]
  const int nval = d.nb * NV * NV, ntot = nval + NDV * ns;
  for (int x = threadIdx.x; x < ntot; x += BLOCK) lds[x] = 0.0;
  __syncthreads();
  if (STAMP) ts[1] = __builtin_amdgcn_s_memtime();
  if (valid) {
    LdsSink3<M> sink;
    sink.row = lds + (ax.x & 0xFFFF);
    sink.stride = (int)(ax.x >> 16);
    // the host gives the pairs of a node that share a wave distinct copy indices (pair_aux[3])
    sink.dacc = lds + nval + (int)((ax.y & 0xFFFF) / NV) * NC + (int)(ax.y >> 16);
    sink.off[0] = 0; sink.off[1] = (int)(ax.z >> 16);
    sink.off[2] = (int)(ax.w & 0xFFFF); sink.off[3] = (int)(ax.w >> 16);
    tet4_row0<M, EXP_MODE>(k, X, U, AX, sink);
  }
  if (STAMP) ts[2] = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (STAMP) ts[3] = __builtin_amdgcn_s_memtime();
  // fold the private copies: diagonal block values into the row slice, rhs straight to memory
  for (int x = threadIdx.x; x < d.nnodes * NDV; x += BLOCK) {
    const int v = x / d.nnodes, n = x - v * d.nnodes;  // consecutive threads -> consecutive nodes (stride NC doubles)
    const double* src = lds + nval + v * ns + n * NC;
    double sum = 0.0;
#pragma unroll
    for (int c = 0; c < NC; c++) sum += src[c];
    if (v < NV * NV) {
      const uint2 nt = ntab[n];  // {rowoff | stride << 16, diagoff}
      const int a = v / NV, b = v - a * NV;
      lds[(nt.x & 0xFFFF) + a * (int)(nt.x >> 16) + (nt.y & 0xFFFF) + b] = sum;  // nothing else writes the diagonal
    } else {
      rhs[(int64_t)(d.n0 + n) * NV + (v - NV * NV)] = sum;
    }
  }
  __syncthreads();
  if (STAMP) ts[4] = __builtin_amdgcn_s_memtime();
  double* out = val + d.vb0;
  for (int x = threadIdx.x; x < nval; x += BLOCK) __builtin_nontemporal_store(lds[x], out + x);
  if (STAMP) {
    ts[5] = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && stamps) {
      long long* o = stamps + ((int64_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 6;
#pragma unroll
      for (int x = 0; x < 6; x++) o[x] = ts[x];
    }
  }
}

// ---- row gather with LDS-staged node records (default) ------------------------------------------
// In-kernel stamps of k_tet4_rg3 (tools/stamp_report.py) put 38% of a wave's lifetime into the initial
// load phase: 256 pairs x 4 nodes x 4 pieces = 4,096 scattered 16-byte requests per workgroup through
// the L1/TA, for only ~84 distinct node records.  Here wave r gathers list entries [64r, 64r+64) of the
// workgroup's node list straight into LDS (LDS-DMA, no VGPRs: ~340 requests), and the pairs read their
// four records from LDS through 8-bit list indices.  Everything after that is k_tet4_rg3.
template <class M, int EXP_MODE, int BLOCK, int MINW, bool STAMP = false, int ABL = 0>
__global__ void __launch_bounds__(BLOCK, MINW)
k_tet4_rg5(const HostPrep::WgDesc* __restrict__ desc, const uint32_t* __restrict__ pair_loc,
           const uint16_t* __restrict__ pair_aux, const uint32_t* __restrict__ nlist,
           const uint16_t* __restrict__ node_tab, const typename M::K k, const double* __restrict__ rec,
           double* __restrict__ val, double* __restrict__ rhs, const int nl_stride, const int acc_doubles,
           long long* __restrict__ stamps, const int pf_dist, const int xcd_remap,
           const uint32_t* __restrict__ pair_eid, const double* __restrict__ elem, const int wg_begin,
           const int stagger) {
  constexpr int NV = M::NV, NW = BLOCK / 64, NC = HostPrep::rg3_diag_copies(BLOCK), NDV = NV * NV + NV;
  constexpr int ns = HostPrep::RG3_DIAG_SLOTS, MAXN = ns / NC, NP = Rec<M>::N / 2;
  extern __shared__ __attribute__((aligned(16))) double lds[];  // [accumulators | node records: NP x nl_stride x 16 B]
  __shared__ uint2 ntab[MAXN];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  long long ts[6], tx[3] = {0, 0, 0};
  if (STAMP) ts[0] = __builtin_amdgcn_s_memtime();
  // Diagnostic knob (rdc_set_option "stagger", default 0): delays the SECOND workgroup of every CU (its LDS allocation
  // does not start at 0) once, in the first dispatch round.  It tests whether the two co-resident workgroups run in
  // lock-step phases (kernel time = data-movement skeleton + arithmetic); measured: the
  // offset changes nothing (2.59 ms for 0..24 K cycles), so phase alignment is not what keeps the two from overlapping.
  if (stagger > 0 && blockIdx.x < 2048u) {
    const unsigned lds_base = __builtin_amdgcn_s_getreg((7 << 11) | 6) ;  // HW_REG_LDS_ALLOC.LDS_BASE (bits 7:0)
    if (lds_base != 0)
      for (int i = 0; i < stagger; i++) __builtin_amdgcn_s_sleep(16);  // 16 x 64 cycles each
  }
  int w = blockIdx.x;
  if (xcd_remap) {  // contiguous range of work items per XCD (see k_tet4_rg3); speed only
    const int q = gridDim.x >> 3, r = gridDim.x & 7, x = blockIdx.x & 7;
    w = x * q + (x < r ? x : r) + (blockIdx.x >> 3);
  }
  w += wg_begin;  // sub-range launches (interior rows before the halo exchange has landed, the rest after)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int idx = lane * NW + wv;
  double* const recs = lds + acc_doubles;
  // Zero the whole accumulator area first: it needs nothing from memory (acc_doubles is a launch constant, a few
  // percent more than this workgroup's slice), so these LDS stores -- which queue behind the co-resident
  // workgroup's atomics -- overlap the ~2,300-cycle latency of the list loads below.  16-byte stores.
  {
    double2* z = reinterpret_cast<double2*>(lds);
    for (int x = threadIdx.x; x < acc_doubles / 2; x += BLOCK) z[x] = make_double2(0.0, 0.0);
  }
  // level 1 (independent): node ids of this wave's round, pair record, descriptor
  const int rounds = nl_stride >> 6;
  uint32_t nid = 0;
  if (wv < rounds) nid = nlist[(size_t)w * nl_stride + wv * 64 + lane];
  const uint32_t pl = pair_loc[(size_t)w * BLOCK + idx];
  const uint4 ax = reinterpret_cast<const uint4*>(pair_aux)[(size_t)w * BLOCK + idx];
  const HostPrep::WgDesc d = desc[w];
  // The work lists are read exactly once, so the level-1 loads above are HBM misses.  Touch the lists of
  // the workgroup that will run pf_dist dispatch slots later on this XCD (workgroups are dealt round-robin
  // over the 8 XCDs; pf_dist is a multiple of 8): by then they are L2 hits.  Fire-and-forget loads whose
  // results are never used -- written in asm so that no wait is generated for them.  Speed only.
  if (pf_dist > 0 && w + pf_dist < (int)gridDim.x) {
    const size_t wp = (size_t)(w + pf_dist);
    unsigned t0, t1, t2;
    const void* p0 = pair_loc + wp * BLOCK + threadIdx.x;
    const void* p1 = reinterpret_cast<const uint4*>(pair_aux) + wp * BLOCK + threadIdx.x;
    const void* p2 = nlist + wp * nl_stride + (threadIdx.x < (unsigned)nl_stride ? threadIdx.x : 0);
    asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %4, off\n\tglobal_load_dword %2, %5, off"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2) : "v"(p0), "v"(p1), "v"(p2) : "memory");
  }
  if (STAMP) {  // diagnostic only: serialise the levels to time them
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    tx[0] = __builtin_amdgcn_s_memtime();
  }
  // level 2: gather the distinct node records into LDS
  if (wv < rounds) {
    const char* src = reinterpret_cast<const char*>(rec) + (size_t)nid * (NP * 16);
#pragma unroll
    for (int p = 0; p < NP; p++)
      __builtin_amdgcn_global_load_lds((glb_ptr)(src + p * 16), (lds_ptr)(recs + (p * nl_stride + wv * 64) * 2), 16, 0, 0);
  }
  if ((int)threadIdx.x < d.nnodes) ntab[threadIdx.x] = reinterpret_cast<const uint2*>(node_tab)[d.n0 + threadIdx.x];
  // The row slice starts one double into LDS when its first CSR value sits at an odd index, so that LDS and global
  // memory have the same 16-byte phase and the store phase can move 16 bytes per lane; the private diagonal
  // accumulators follow at the next even index (16-byte reads in the fold).
  const int nval = d.nb * NV * NV, sh = (int)(d.vb0 & 1), dbase = (nval + sh + 1) & ~1;
  double* const sl = lds + sh;
  if (STAMP) tx[1] = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA has landed ...
  if (STAMP) tx[2] = __builtin_amdgcn_s_memtime();
  __syncthreads();                                   // ... and so has everybody else's
  if (STAMP) ts[1] = __builtin_amdgcn_s_memtime();
  if (pl != 0xFFFFFFFFu && !(ABL >= 3 && ABL <= 5)) {  // diagnostic builds 3..5: no compute phase
    double X[4][3], U[4][NV], AX[4][M::NAUX > 0 ? M::NAUX : 1];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int li = (pl >> (8 * j)) & 0xFF;
      double r[2 * NP];
#pragma unroll
      for (int p = 0; p < NP; p++) {
        const double2 v2 = reinterpret_cast<const double2*>(recs)[p * nl_stride + li];
        r[2 * p] = v2.x; r[2 * p + 1] = v2.y;
      }
      X[j][0] = r[0]; X[j][1] = r[1]; X[j][2] = r[2];
#pragma unroll
      for (int v = 0; v < NV; v++) U[j][v] = r[3 + v];
      if (M::NAUX > 0) {
#pragma unroll
        for (int v = 0; v < M::NAUX; v++) AX[j][v] = r[3 + NV + v];
      } else {
        AX[j][0] = 0.0;
      }
    }
    LdsSink3<M, ABL> sink;
    sink.row = sl + (ax.x & 0xFFFF);
    sink.stride = (int)(ax.x >> 16);
    sink.dacc = lds + dbase + (int)((ax.y & 0xFFFF) / NV) * NC + (int)(ax.y >> 16);
    sink.off[0] = 0; sink.off[1] = (int)(ax.z >> 16);
    sink.off[2] = (int)(ax.w & 0xFFFF); sink.off[3] = (int)(ax.w >> 16);
    const double* ED = nullptr;  // per-element inputs (ADPM tract vector): one more 4-byte list entry per pair
    if (M::NELEM > 0 || M::AUX_LOCAL_NODE >= 0) {
      const uint32_t pe = pair_eid[(size_t)w * BLOCK + idx];
      if (M::NELEM > 0) ED = elem + (size_t)(pe & 0x3FFFFFFFu) * M::NELEM;
      if (M::AUX_LOCAL_NODE >= 0) {  // aux field read at local node 1 of the element only (src/proteas.C:481)
        const int pos1 = (int)(pe >> 30);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int v = 0; v < (M::NAUX > 0 ? M::NAUX : 1); v++) AX[j][v] = (j == pos1) ? AX[j][v] : 0.0;
      }
    }
    tet4_row0<M, EXP_MODE>(k, X, U, AX, sink, ED);
    if (ABL == 2 && sink.sink == 1.2345e300) rhs[0] = sink.sink;  // keeps the arithmetic alive
  }
  if (STAMP) ts[2] = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (STAMP) ts[3] = __builtin_amdgcn_s_memtime();
  if (!(ABL == 4 || ABL == 5))  // diagnostic builds: no fold
  for (int x = threadIdx.x; x < d.nnodes * NDV; x += BLOCK) {
    const int v = x / d.nnodes, n = x - v * d.nnodes;
    const double2* src = reinterpret_cast<const double2*>(lds + dbase + v * ns + n * NC);  // NC is even
    double sum = 0.0;
#pragma unroll
    for (int c = 0; c < NC / 2; c++) { const double2 t = src[c]; sum += t.x; sum += t.y; }
    if (v < NV * NV) {
      const uint2 nt = ntab[n];
      const int a = v / NV, b = v - a * NV;
      sl[(nt.x & 0xFFFF) + a * (int)(nt.x >> 16) + (nt.y & 0xFFFF) + b] = sum;
    } else {
      rhs[(int64_t)(d.n0 + n) * NV + (v - NV * NV)] = sum;
    }
  }
  __syncthreads();
  if (STAMP) ts[4] = __builtin_amdgcn_s_memtime();
  double* out = val + d.vb0;   // out[g] <-> sl[g]; out + sh and lds + 2 * sh are 16-byte aligned
  {
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    const int npair = (nval - sh) >> 1;
    const v2d_t* src = reinterpret_cast<const v2d_t*>(lds + 2 * sh);
    v2d_t* dst = reinterpret_cast<v2d_t*>(out + sh);
    if (!(ABL == 5 || ABL == 6))  // diagnostic builds: no store phase
    for (int x = threadIdx.x; x < npair; x += BLOCK) __builtin_nontemporal_store(src[x], dst + x);
    if (sh && threadIdx.x == 0) __builtin_nontemporal_store(sl[0], out);
    if (((nval - sh) & 1) && threadIdx.x == 64) __builtin_nontemporal_store(sl[nval - 1], out + nval - 1);
  }
  if (STAMP) {
    ts[5] = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && stamps) {
      long long* o = stamps + ((int64_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 6;
#pragma unroll
      for (int x = 0; x < 6; x++) o[x] = ts[x];
      // sub-stamps of phase 0 go behind the main table
      long long* o2 = stamps + (int64_t)gridDim.x * NW * 6 + ((int64_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 3;
#pragma unroll
      for (int x = 0; x < 3; x++) o2[x] = tx[x] - ts[0];
    }
  }
}

// ---- k_tet4_rg5 walking over work items with a TAIL PREFETCH (experimental, rdc_set_option("kernel", 6)) ---------
// Vector-memory operations of a wave complete in issue order, and a new workgroup's list loads sit behind the store
// burst of the workgroup that just finished on its CU.  Here a workgroup keeps its CU slot and, right after the compute
// barrier of item w, issues the list loads of item w + gridDim.x (registers are free then; they land during the fold),
// then the LDS-DMA of that item's node records (the record area is dead once the compute phase has gathered from it),
// and only then the stores of item w.  Models without per-element inputs only.
template <class M, int EXP_MODE, int BLOCK>
__global__ void __launch_bounds__(BLOCK, 2)
k_tet4_rg6(const HostPrep::WgDesc* __restrict__ desc, const uint32_t* __restrict__ pair_loc,
           const uint16_t* __restrict__ pair_aux, const uint32_t* __restrict__ nlist,
           const uint16_t* __restrict__ node_tab, const typename M::K k, const double* __restrict__ rec,
           double* __restrict__ val, double* __restrict__ rhs, const int nl_stride, const int acc_doubles,
           const int w_begin, const int w_end) {
  constexpr int NV = M::NV, NW = BLOCK / 64, NC = HostPrep::rg3_diag_copies(BLOCK), NDV = NV * NV + NV;
  constexpr int ns = HostPrep::RG3_DIAG_SLOTS, MAXN = ns / NC, NP = Rec<M>::N / 2;
  extern __shared__ __attribute__((aligned(16))) double lds[];  // [accumulators | node records: NP x nl_stride x 16 B]
  __shared__ uint2 ntab[MAXN];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int idx = lane * NW + wv;
  const int rounds = nl_stride >> 6;
  double* const recs = lds + acc_doubles;
  int w = w_begin + (int)blockIdx.x;
  if (w >= w_end) return;
  // prologue: lists and records of the first item
  uint32_t nid = 0;
  if (wv < rounds) nid = nlist[(size_t)w * nl_stride + wv * 64 + lane];
  uint32_t pl = pair_loc[(size_t)w * BLOCK + idx];
  uint4 ax = reinterpret_cast<const uint4*>(pair_aux)[(size_t)w * BLOCK + idx];
  // only four fields of the 64-byte descriptor are kept (SGPRs are short: the model constants take ~80 of them, and a
  // spilled register means scratch traffic, which shares vmcnt with the DMA and the stores)
  struct Item { int n0, nnodes, nb; int64_t vb0; };
  Item d = {desc[w].n0, desc[w].nnodes, desc[w].nb, desc[w].vb0};
  uint2 nt = make_uint2(0u, 0u);
  if ((int)threadIdx.x < d.nnodes) nt = reinterpret_cast<const uint2*>(node_tab)[d.n0 + threadIdx.x];
  if (wv < rounds) {
    const char* src = reinterpret_cast<const char*>(rec) + (size_t)nid * (NP * 16);
#pragma unroll
    for (int p = 0; p < NP; p++)
      __builtin_amdgcn_global_load_lds((glb_ptr)(src + p * 16), (lds_ptr)(recs + (p * nl_stride + wv * 64) * 2), 16, 0, 0);
  }
  for (;;) {
    const int wn = w + (int)gridDim.x;
    const bool more = wn < w_end;
    Item d2 = d;
    if (more) { d2.n0 = desc[wn].n0; d2.nnodes = desc[wn].nnodes; d2.nb = desc[wn].nb; d2.vb0 = desc[wn].vb0; }  // scalar loads, long before they are needed
    {  // zero the accumulators: the previous item's store phase has read them (barrier at the loop end)
      double2* z = reinterpret_cast<double2*>(lds);
      for (int x = threadIdx.x; x < acc_doubles / 2; x += BLOCK) z[x] = make_double2(0.0, 0.0);
    }
    if ((int)threadIdx.x < d.nnodes) ntab[threadIdx.x] = nt;
    const int nval = d.nb * NV * NV, sh = (int)(d.vb0 & 1), dbase = (nval + sh + 1) & ~1;
    double* const sl = lds + sh;
    // This item's records have landed.  vmcnt(0) also waits for the previous item's stores (issued after the DMA); a
    // counted wait (all but the wave's store instructions) + raw barrier + LDS reads the compiler cannot see would let
    // them drain under the compute phase -- tried at the end of round 1, not yet correct.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (pl != 0xFFFFFFFFu) {
      double X[4][3], U[4][NV], AX[4][M::NAUX > 0 ? M::NAUX : 1];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int li = (pl >> (8 * j)) & 0xFF;
        double r[2 * NP];
#pragma unroll
        for (int p = 0; p < NP; p++) {
          const double2 v2 = reinterpret_cast<const double2*>(recs)[p * nl_stride + li];
          r[2 * p] = v2.x; r[2 * p + 1] = v2.y;
        }
        X[j][0] = r[0]; X[j][1] = r[1]; X[j][2] = r[2];
#pragma unroll
        for (int v = 0; v < NV; v++) U[j][v] = r[3 + v];
        if (M::NAUX > 0) {
#pragma unroll
          for (int v = 0; v < M::NAUX; v++) AX[j][v] = r[3 + NV + v];
        } else {
          AX[j][0] = 0.0;
        }
      }
      LdsSink3<M, 0> sink;
      sink.row = sl + (ax.x & 0xFFFF);
      sink.stride = (int)(ax.x >> 16);
      sink.dacc = lds + dbase + (int)((ax.y & 0xFFFF) / NV) * NC + (int)(ax.y >> 16);
      sink.off[0] = 0; sink.off[1] = (int)(ax.z >> 16);
      sink.off[2] = (int)(ax.w & 0xFFFF); sink.off[3] = (int)(ax.w >> 16);
      tet4_row0<M, EXP_MODE>(k, X, U, AX, sink, nullptr);
    }
    __syncthreads();
    // ---- tail prefetch: lists of the next item, issued before this item's stores -------------------------------
    uint32_t nid2 = 0, pl2 = 0xFFFFFFFFu;
    uint4 ax2 = make_uint4(0u, 0u, 0u, 0u);
    uint2 nt2 = make_uint2(0u, 0u);
    if (more) {
      if (wv < rounds) nid2 = nlist[(size_t)wn * nl_stride + wv * 64 + lane];
      pl2 = pair_loc[(size_t)wn * BLOCK + idx];
      ax2 = reinterpret_cast<const uint4*>(pair_aux)[(size_t)wn * BLOCK + idx];
      if ((int)threadIdx.x < d2.nnodes) nt2 = reinterpret_cast<const uint2*>(node_tab)[d2.n0 + threadIdx.x];
    }
    // fold of this item (as k_tet4_rg5)
    for (int x = threadIdx.x; x < d.nnodes * NDV; x += BLOCK) {
      const int v = x / d.nnodes, n = x - v * d.nnodes;
      const double2* src = reinterpret_cast<const double2*>(lds + dbase + v * ns + n * NC);  // NC is even
      double sum = 0.0;
#pragma unroll
      for (int c = 0; c < NC / 2; c++) { const double2 t = src[c]; sum += t.x; sum += t.y; }
      if (v < NV * NV) {
        const uint2 q = ntab[n];
        const int a = v / NV, b = v - a * NV;
        sl[(q.x & 0xFFFF) + a * (int)(q.x >> 16) + (q.y & 0xFFFF) + b] = sum;
      } else {
        rhs[(int64_t)(d.n0 + n) * NV + (v - NV * NV)] = sum;
      }
    }
    __syncthreads();
    if (more) {  // records of the next item into the (now dead) record area
      // every prefetched list value is consumed HERE, while no LDS-DMA is in flight: with one in flight the compiler
      // would wait vmcnt(0) -- stores included -- at their first use
      asm volatile("" :: "v"(nid2), "v"(pl2), "v"(ax2.x), "v"(ax2.y), "v"(ax2.z), "v"(ax2.w), "v"(nt2.x), "v"(nt2.y));
      if (wv < rounds) {
        const char* src = reinterpret_cast<const char*>(rec) + (size_t)nid2 * (NP * 16);
#pragma unroll
        for (int p = 0; p < NP; p++)
          __builtin_amdgcn_global_load_lds((glb_ptr)(src + p * 16), (lds_ptr)(recs + (p * nl_stride + wv * 64) * 2), 16, 0, 0);
      }
    }
    {  // stores of this item
      double* out = val + d.vb0;
      typedef double v2d_t __attribute__((ext_vector_type(2)));
      const int npair = (nval - sh) >> 1;
      const v2d_t* src = reinterpret_cast<const v2d_t*>(lds + 2 * sh);
      v2d_t* dst = reinterpret_cast<v2d_t*>(out + sh);
      for (int x = threadIdx.x; x < npair; x += BLOCK) __builtin_nontemporal_store(src[x], dst + x);
      if (sh && threadIdx.x == 0) __builtin_nontemporal_store(sl[0], out);
      if (((nval - sh) & 1) && threadIdx.x == 64) __builtin_nontemporal_store(sl[nval - 1], out + nval - 1);
    }
    if (!more) break;
    // the LDS reads of the store phase are done before the next item zeroes the slice; raw barrier: __syncthreads()
    // would drain vmcnt (the stores just issued and the DMA) here
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    w = wn; pl = pl2; ax = ax2; d = d2; nt = nt2;
  }
}

// ---- persistent, software-pipelined row gather -----------------------------------------------
// Same arithmetic and LDS accumulation as k_tet4_rg5, but a workgroup loops over work items (w = blockIdx.x,
// += gridDim.x) and EVERYTHING an item needs arrives by LDS-DMA (global_load_lds: no VGPR destination) while earlier
// items are being evaluated, in a three-stage pipeline:
//   during item w      : node list of item w + 2G            -> nlbuf[(it + 2) % 3]
//                        records (gathered through the node list of w + G, read from nlbuf[(it + 1) % 3]), pair
//                        lists, descriptor and diagonal-slot table of item w + G -> item buffer (it + 1) & 1
// No value loaded from global memory is held in a VGPR across the compute phase: the compute phase already uses all
// 256 registers, and a spill costs far more here than elsewhere -- scratch traffic shares vmcnt with the DMA, so
// every reload would wait for the whole prefetch (the first version of this kernel did exactly that: 3.8 ms).
template <class M, int EXP_MODE, int BLOCK, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW)
k_tet4_rg4(const HostPrep::WgDesc* __restrict__ desc, const uint32_t* __restrict__ pair_loc,
           const uint16_t* __restrict__ pair_aux, const uint32_t* __restrict__ nlist,
           const uint16_t* __restrict__ wg_ntab, const typename M::K k, const double* __restrict__ rec,
           double* __restrict__ val, double* __restrict__ rhs, const int nwg, const int nl_stride,
           const int acc_doubles, const int stagger) {
  constexpr int NV = M::NV, NW = BLOCK / 64, NC = HostPrep::rg3_diag_copies(BLOCK), NDV = NV * NV + NV;
  constexpr int ns = HostPrep::RG3_DIAG_SLOTS, NP = Rec<M>::N / 2;  // NP 16-byte pieces per node record
  constexpr int NTAB = 16;                                           // diagonal-slot table entries per item (uint2)
  extern __shared__ __attribute__((aligned(16))) double lds[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  // LDS map: [accumulators] [2 x (records | pair_aux | pair_loc | descriptor | ntab)] [3 x node list]
  const int rec_doubles = NP * nl_stride * 2;
  const int buf_doubles = rec_doubles + BLOCK * 2 + BLOCK / 2 + 8 + NTAB;
  double* const bufs = lds + acc_doubles;
  uint32_t* const nlbuf = reinterpret_cast<uint32_t*>(bufs + 2 * buf_doubles);  // [3][nl_stride]
  // thread-derived values are re-derived from an opaque copy of tt in every iteration (and the time-step
  // factor every coefficient is multiplied with is made opaque, too): otherwise the compiler hoists the address
  // arithmetic and the products of the model constants out of the item loop, runs out of registers and spills them
  int tt = threadIdx.x, lane = tt & 63, wv = tt >> 6;
  int idx = lane * NW + wv;  // pair slot of this thread (see k_tet4_rg3)
  const int rounds = nl_stride >> 6;
  const int G = gridDim.x;

  auto dma_nlist = [&](int item, int slot) {
    if (wv < rounds) {
      const char* src = reinterpret_cast<const char*>(nlist + (size_t)item * nl_stride + wv * 64 + lane);
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(nlbuf + slot * nl_stride + wv * 64), 4, 0, 0);
    }
  };
  // all LDS-DMA of one work item: records (wave r gathers list entries [64r, 64r+64)), pair lists, descriptor, ntab
  auto prefetch = [&](int item, uint32_t nid, int b) {
    double* base = bufs + b * buf_doubles;
    if (wv < rounds) {
      const char* src = reinterpret_cast<const char*>(rec) + (size_t)nid * (NP * 16);
#pragma unroll
      for (int p = 0; p < NP; p++)
        __builtin_amdgcn_global_load_lds((glb_ptr)(src + p * 16), (lds_ptr)(base + (p * nl_stride + wv * 64) * 2), 16, 0, 0);
    }
    const char* ax = reinterpret_cast<const char*>(pair_aux) + ((size_t)item * BLOCK + tt) * 16;
    __builtin_amdgcn_global_load_lds((glb_ptr)ax, (lds_ptr)(base + rec_doubles + wv * 128), 16, 0, 0);
    const char* pl = reinterpret_cast<const char*>(pair_loc) + ((size_t)item * BLOCK + tt) * 4;
    __builtin_amdgcn_global_load_lds((glb_ptr)pl, (lds_ptr)(base + rec_doubles + BLOCK * 2 + wv * 32), 4, 0, 0);
    if (tt < 4) {  // the 64-byte workgroup descriptor
      const char* dd = reinterpret_cast<const char*>(desc + item) + tt * 16;
      __builtin_amdgcn_global_load_lds((glb_ptr)dd, (lds_ptr)(base + rec_doubles + BLOCK * 2 + BLOCK / 2), 16, 0, 0);
    } else if (tt >= 64 && tt < 64 + NTAB / 2) {  // 16 x 8 bytes of diagonal-slot table (wave 1)
      const char* nt = reinterpret_cast<const char*>(wg_ntab) + ((size_t)item * NTAB) * 8 + (tt - 64) * 16;
      __builtin_amdgcn_global_load_lds((glb_ptr)nt, (lds_ptr)(base + rec_doubles + BLOCK * 2 + BLOCK / 2 + 8), 16, 0, 0);
    }
  };

  int w = blockIdx.x;
  if (w >= nwg) return;
  // optional: delay the second resident workgroup of every CU (dispatched in the second half of the grid) so the
  // two do not run their phases in lock step
  if (stagger > 0 && (int)blockIdx.x >= (int)gridDim.x / 2)
    for (int x = 0; x < stagger; x++) __builtin_amdgcn_s_sleep(127);
  // ---- prologue: first item's data, second item's node list; accumulators start at zero -----------------
  {
    const uint32_t nid0 = (wv < rounds) ? nlist[(size_t)w * nl_stride + wv * 64 + lane] : 0u;
    prefetch(w, nid0, 0);
    if (w + G < nwg) dma_nlist(w + G, 1);
    double2* z = reinterpret_cast<double2*>(lds);
    for (int x = tt; x < acc_doubles / 2; x += BLOCK) z[x] = make_double2(0.0, 0.0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int it = 0; w < nwg; it++, w += G) {
    tt = threadIdx.x;
    asm volatile("" : "+v"(tt));
    lane = tt & 63; wv = tt >> 6; idx = lane * NW + wv;
    typename M::K kk = k;
    asm volatile("" : "+s"(kk.DT2));
    const int b = it & 1;
    const double* base = bufs + b * buf_doubles;
    // descriptor fields as scalars (an LDS read lands in VGPRs)
    struct { int64_t vb0; int n0, nnodes, nb; } d;
    {
      const HostPrep::WgDesc* dp = reinterpret_cast<const HostPrep::WgDesc*>(base + rec_doubles + BLOCK * 2 + BLOCK / 2);
      const int64_t vb = dp->vb0;
      d.vb0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(vb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(vb & 0xFFFFFFFF));
      d.n0 = __builtin_amdgcn_readfirstlane(dp->n0);
      d.nnodes = __builtin_amdgcn_readfirstlane(dp->nnodes);
      d.nb = __builtin_amdgcn_readfirstlane(dp->nb);
    }
    const uint2* ntab = reinterpret_cast<const uint2*>(base + rec_doubles + BLOCK * 2 + BLOCK / 2 + 8);
    const int wn = w + G;
    if (wn < nwg) {  // buffer b^1 and list slot (it+2)%3 were last read before the barriers that ended earlier iterations
      const uint32_t nid = (wv < rounds) ? nlbuf[((it + 1) % 3) * nl_stride + wv * 64 + lane] : 0u;
      prefetch(wn, nid, b ^ 1);
      if (wn + G < nwg) dma_nlist(wn + G, (it + 2) % 3);
    }
    const uint4 ax = reinterpret_cast<const uint4*>(base + rec_doubles)[idx];
    const uint32_t pl = reinterpret_cast<const uint32_t*>(base + rec_doubles + BLOCK * 2)[idx];
    const int nval = d.nb * NV * NV, sh = (int)(d.vb0 & 1), dbase = (nval + sh + 1) & ~1;  // see k_tet4_rg5
    double* const sl = lds + sh;
    if (pl != 0xFFFFFFFFu) {
      double X[4][3], U[4][NV], AX[4][M::NAUX > 0 ? M::NAUX : 1];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int li = (pl >> (8 * j)) & 0xFF;
        double r[2 * NP];
#pragma unroll
        for (int p = 0; p < NP; p++) {
          const double2 v2 = reinterpret_cast<const double2*>(base)[p * nl_stride + li];
          r[2 * p] = v2.x; r[2 * p + 1] = v2.y;
        }
        X[j][0] = r[0]; X[j][1] = r[1]; X[j][2] = r[2];
#pragma unroll
        for (int v = 0; v < NV; v++) U[j][v] = r[3 + v];
        if (M::NAUX > 0) {
#pragma unroll
          for (int v = 0; v < M::NAUX; v++) AX[j][v] = r[3 + NV + v];
        } else {
          AX[j][0] = 0.0;
        }
      }
      LdsSink3<M> sink;
      sink.row = sl + (ax.x & 0xFFFF);
      sink.stride = (int)(ax.x >> 16);
      sink.dacc = lds + dbase + (int)((ax.y & 0xFFFF) / NV) * NC + (int)(ax.y >> 16);
      sink.off[0] = 0; sink.off[1] = (int)(ax.z >> 16);
      sink.off[2] = (int)(ax.w & 0xFFFF); sink.off[3] = (int)(ax.w >> 16);
      tet4_row0<M, EXP_MODE>(kk, X, U, AX, sink);
    }
    // the DMA of the next items was issued a whole compute phase ago, and the stores of the previous item before
    // that: this wait does not stall
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // fold the private copies (and clear them for the next item)
    for (int x = tt; x < d.nnodes * NDV; x += BLOCK) {
      const int v = x / d.nnodes, n = x - v * d.nnodes;
      double2* src = reinterpret_cast<double2*>(lds + dbase + v * ns + n * NC);
      double sum = 0.0;
#pragma unroll
      for (int c = 0; c < NC / 2; c++) { const double2 t = src[c]; sum += t.x; sum += t.y; src[c] = make_double2(0.0, 0.0); }
      if (v < NV * NV) {
        const uint2 nt = ntab[n];
        const int a = v / NV, bb = v - a * NV;
        sl[(nt.x & 0xFFFF) + a * (int)(nt.x >> 16) + (nt.y & 0xFFFF) + bb] = sum;
      } else {
        rhs[(int64_t)(d.n0 + n) * NV + (v - NV * NV)] = sum;
      }
    }
    __syncthreads();
    double* out = val + d.vb0;  // out[g] <-> sl[g]; accumulators are cleared by their last reader: no separate zero pass
    {
      typedef double v2d_t __attribute__((ext_vector_type(2)));
      const int npair = (nval - sh) >> 1;
      v2d_t* src = reinterpret_cast<v2d_t*>(lds + 2 * sh);
      v2d_t* dst = reinterpret_cast<v2d_t*>(out + sh);
      const v2d_t zero2 = {0.0, 0.0};
      for (int x = tt; x < npair; x += BLOCK) { __builtin_nontemporal_store(src[x], dst + x); src[x] = zero2; }
      if (sh && tt == 0) { __builtin_nontemporal_store(sl[0], out); sl[0] = 0.0; }
      if (((nval - sh) & 1) && tt == 64) { __builtin_nontemporal_store(sl[nval - 1], out + nval - 1); sl[nval - 1] = 0.0; }
    }
    __syncthreads();
  }
}

// ---- staged row gather (the default TET4 path) ----------------------------------------------
// One thread per (row node, incident element) pair evaluates one equation row at a time into an
// LDS stage buffer with plain stores; after a barrier the same workgroup sums, for every node
// block of its rows, the staged contributions in a FIXED order (bitwise reproducible) and writes
// the CSR values straight to HBM with streaming stores.  Compared with k_tet4_rowgather: no LDS
// atomics (an FP64 ds_add costs ~45 issue cycles on gfx950), no LDS row slice, no zero / flush
// passes, and only two dependent load levels (descriptor | pair record  ->  node records).
// stage row of a pair: [column 0 = diagonal block: NV values][rhs value][columns 1..3: NV values each]
struct StageSink {
  double* my;  // stage row of this thread
  __device__ __forceinline__ void ke(int, int b, int j, double v) { my[j == 0 ? b : j * NVs + 1 + b] = v; }
  __device__ __forceinline__ void fe(int, double v) { my[NVs] = v; }
  int NVs;
};

template <class M, int EXP_MODE, int BLOCK, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW)
k_tet4_rg2(const HostPrep::WgDesc* __restrict__ desc, const uint32_t* __restrict__ pair_rec,
           const HostPrep::Chunk* __restrict__ chunk, const HostPrep::StoreDesc* __restrict__ sdesc,
           const uint16_t* __restrict__ contrib, const typename M::K k, const double* __restrict__ rec,
           double* __restrict__ val, double* __restrict__ rhs, const int dbg) {
  constexpr int NV = M::NV, STRIDE = (4 * NV + 1) | 1, SLOT = NV + 1;  // odd stride: conflict-free ds_write_b64
  __shared__ double stage[BLOCK * STRIDE + SLOT];  // + SLOT: the gather reads SLOT values from the last column too
  __shared__ double outbuf[BLOCK * SLOT];  // partial sums of one row pass: NV values + rhs per slot, nout <= BLOCK
  __shared__ uint2 clist2[BLOCK];          // 4 contribution entries (uint16) per pair
  __shared__ uint2 lchunk[BLOCK];
  __shared__ uint2 lsdesc[BLOCK];
  const uint16_t* clist = reinterpret_cast<const uint16_t*>(clist2);
  const int w = blockIdx.x, tid = threadIdx.x;
  // level-1 loads: independent of each other
  const uint4 pr = reinterpret_cast<const uint4*>(pair_rec)[(int64_t)w * BLOCK + tid];
  const HostPrep::WgDesc d = desc[w];
  const bool valid = pr.x != 0xFFFFFFFFu;
  // level-2 loads
  Tet4Pre<M> P;
  if (valid) {
    double X[4][3], U[4][NV], AX[4][M::NAUX > 0 ? M::NAUX : 1];
    load_rec<M>(rec, pr.x, X[0], U[0], AX[0]);
    load_rec<M>(rec, pr.y, X[1], U[1], AX[1]);
    load_rec<M>(rec, pr.z, X[2], U[2], AX[2]);
    load_rec<M>(rec, pr.w, X[3], U[3], AX[3]);
    tet4_prepare<M, EXP_MODE>(k, X, U, AX, P);
  }
  if (tid < d.np) clist2[tid] = reinterpret_cast<const uint2*>(contrib + d.c0)[tid];
  if (tid < d.nch) lchunk[tid] = reinterpret_cast<const uint2*>(chunk)[d.ch0 + tid];
  if (tid < d.nb) lsdesc[tid] = reinterpret_cast<const uint2*>(sdesc)[d.bb0 + tid];
  const int nvals = d.nb * NV;
  StageSink sink;
  sink.my = stage + tid * STRIDE;
  sink.NVs = NV;
#pragma unroll
  for (int a = 0; a < NV; a++) {
    // launder the time-step factor: every coefficient of the row depends on it, so the compiler
    // cannot hoist row a's arithmetic above the previous barrier
    typename M::K kk = k;
    asm volatile("" : "+s"(kk.DT2));
    if (valid && !(dbg & 1)) tet4_row<M>(kk, P, a, sink);
    __syncthreads();
    // ---- gather: one chunk (<= RG2_CHUNK contributions to one node block) per thread ----------
    if (tid < d.nch && !(dbg & 2)) {
      const uint2 c = lchunk[tid];  // {cbeg | cnt << 16, dst}
      const int cbeg = c.x & 0xFFFF, cnt = c.x >> 16, dst = c.y & 0xFFFF;
      double acc[SLOT];
#pragma unroll
      for (int b = 0; b < SLOT; b++) acc[b] = 0.0;
#pragma unroll
      for (int x = 0; x < HostPrep::RG2_CHUNK; x++)
        if (x < cnt) {
          const double* src = stage + clist[cbeg + x];
#pragma unroll
          for (int b = 0; b < SLOT; b++) acc[b] += src[b];  // 6th value: rhs (meaningful for diagonal blocks only)
        }
#pragma unroll
      for (int b = 0; b < SLOT; b++) outbuf[dst * SLOT + b] = acc[b];
    }
    __syncthreads();
    // ---- store: consecutive threads -> consecutive CSR values, streaming stores -----------------
    for (int x = tid; x < nvals && !(dbg & 4); x += BLOCK) {
      const int ob = x / NV, b = x - ob * NV;
      const uint2 sd = lsdesc[ob];  // {outoff | len << 16 | nextra << 24, extra | diag << 16 | node << 24}
      const int outoff = sd.x & 0xFFFF, len = (sd.x >> 16) & 0xFF, nextra = sd.x >> 24, extra = sd.y & 0xFFFF;
      double v = outbuf[ob * SLOT + b];
      for (int e = 0; e < nextra; e++) v += outbuf[(extra + e) * SLOT + b];
      __builtin_nontemporal_store(v, val + d.vb0 + outoff + a * NV * len + b);
      if (b == 0 && ((sd.y >> 16) & 0xFF)) {  // diagonal block: its slot also carries the node's rhs entry
        double r = outbuf[ob * SLOT + NV];
        for (int e = 0; e < nextra; e++) r += outbuf[(extra + e) * SLOT + NV];
        rhs[(int64_t)(d.n0 + (int)(sd.y >> 24)) * NV + a] = r;
      }
    }
    // (the next barrier, after the next row's compute, orders these outbuf reads before its reuse)
  }
}

// ---- coloured -------------------------------------------------------------------------------
template <class M>
struct RmwSink {
  double* row;
  double* prhs;
  int64_t stride;
  int64_t off[4];
  bool first[4], first_rhs;
  __device__ __forceinline__ void ke(int a, int b, int j, double v) {
    double* p = row + a * stride + off[j] + b;
    *p = first[j] ? v : (*p + v);
  }
  __device__ __forceinline__ void fe(int a, double v) {
    double* p = prhs + a;
    *p = first_rhs ? v : (*p + v);
  }
};

template <class M, int EXP_MODE>
__global__ void __launch_bounds__(256, 2)  // two waves per SIMD: 292 registers (one wave) ran 2x slower than a few spills
k_tet4_coloured(const MeshDev m, const typename M::K k, int64_t first, int64_t count, const double* __restrict__ rec,
                const double* __restrict__ elem, double* __restrict__ val, double* __restrict__ rhs) {
  constexpr int NV = M::NV;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int64_t e = m.elem_order[first + t];
  const uint64_t fm = m.first_mask[e];
  const uint32_t fr = m.first_rhs[e];
#pragma unroll 1
  for (int i = 0; i < 4; i++) {
    const int64_t I = m.conn[e * 4 + i];
    if (I >= m.n_owned) continue;
    double X[4][3], U[4][NV], AX[4][M::NAUX > 0 ? M::NAUX : 1];
    RmwSink<M> sink;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int jo = j ^ i;
      const int64_t n = m.conn[e * 4 + jo];
      load_rec<M>(rec, n, X[j], U[j], AX[j]);
      if (M::AUX_LOCAL_NODE >= 0 && jo != M::AUX_LOCAL_NODE) {
#pragma unroll
        for (int v = 0; v < (M::NAUX > 0 ? M::NAUX : 1); v++) AX[j][v] = 0.0;
      }
      sink.off[j] = NV * (int64_t)m.eslot[e * 16 + i * 4 + jo];
      sink.first[j] = (fm >> (i * 4 + jo)) & 1ull;
    }
    const int64_t b0 = m.bptr[I];
    sink.stride = NV * (m.bptr[I + 1] - b0);
    sink.row = val + NV * NV * b0;
    sink.prhs = rhs + I * NV;
    sink.first_rhs = (fr >> i) & 1u;
    tet4_row0<M, EXP_MODE>(k, X, U, AX, sink, M::NELEM > 0 ? elem + e * M::NELEM : nullptr);
  }
}

// Node records.  Two-part assembly (rdc_assembly.h, "part"): part 1 packs only the OWNED nodes -- its rows read
// nothing else, and the ghost rows of u may be being rewritten by the halo exchange on another stream -- and
// records an event; part 2 (possibly on another stream) waits for that event, packs only the GHOST nodes and then
// reads both.  No record is written while a kernel of the other part may read it.
template <class M>
static hipError_t pack_nodes(const LaunchArgs& a) {
  int64_t nb = 0, ne = a.m.n_node;
  if (a.pack_part == 1) ne = a.m.n_owned;
  else if (a.pack_part == 2) { nb = a.m.n_owned; if (a.pack_event) (void)hipStreamWaitEvent(a.stream, a.pack_event, 0); }
  const int64_t total = (ne - nb) * Rec<M>::N;
  if (total > 0) {
    int64_t grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL((k_pack_nodes<M>), dim3((unsigned)grid), dim3(256), 0, a.stream, nb, ne, a.m.xyz, a.u, a.aux, a.packed, a.ev_ticket ? a.ev_ticket + (a.pack_part == 2 ? 16 : 0) : nullptr);
  } else if (a.ev_ticket) {   // nothing to pack (a part without ghost nodes): the cluster counter of the launch that follows is still reset
    (void)hipMemsetAsync(a.ev_ticket + (a.pack_part == 2 ? 16 : 0), 0, sizeof(int), a.stream);
  }
  if (a.pack_part == 1 && a.pack_event) (void)hipEventRecord(a.pack_event, a.stream);
  return hipGetLastError();
}

hipError_t pack_nodes_pihna(const LaunchArgs& a) { return pack_nodes<Pihna>(a); }

template <class M, int EXP_MODE>
static hipError_t launch_fast_impl(const LaunchArgs& a, const typename M::K& k) {
  {
    const hipError_t pe = pack_nodes<M>(a);
    if (pe != hipSuccess) return pe;
  }
  // the timed region starts after the (tiny) record pack: it brackets the dominant kernel only
  if (a.ev_start) (void)hipEventRecord(a.ev_start, a.stream);
  // element visits in coefficient form (rdc_tet4_evc.hip) for the models that profit, when the context has the lists
  if constexpr (EvcEligible<M>::value)
  if (a.use_ev && a.ev.n_wg > 0 && a.strategy == RDC_SCATTER_ROWGATHER) return launch_tet4_evc<M>(a, k);
  // models with per-element inputs (M::NELEM > 0) or a local-node aux mask exist only as k_tet4_rg5 and k_tet4_coloured
  if constexpr (M::NELEM == 0 && M::AUX_LOCAL_NODE < 0)
  if (a.strategy == RDC_SCATTER_ROWGATHER && a.rg2.n_wg > 0 && a.rg2.pair_aux && a.rg2.nlist && a.rg2.wg_ntab && a.rg2.block == 256 &&
      a.opt_kernel == 4) {
    constexpr int BLOCK = 256;
    const int nl = a.rg2.nl_stride;
    const int acc_doubles = (int)((a.rg2.lds_bytes / sizeof(double) + 3) & ~(size_t)1);
    const size_t lds_bytes = sizeof(double) * ((size_t)acc_doubles + 2 * ((size_t)(Rec<M>::N / 2) * nl * 2 + BLOCK * 2 + BLOCK / 2 + 8 + 16) +
                                                3 * (size_t)nl / 2);
    int grid = a.opt_grid > 0 ? a.opt_grid : 512;
    if (grid > a.rg2.n_wg) grid = a.rg2.n_wg;
#define RDC_RG4(MINW)                                                                                              \
  hipLaunchKernelGGL((k_tet4_rg4<M, EXP_MODE, BLOCK, MINW>), dim3(grid), dim3(BLOCK), lds_bytes, a.stream, a.rg2.desc,    \
                     a.rg2.pair_loc, a.rg2.pair_aux, a.rg2.nlist, a.rg2.wg_ntab, k, a.packed, a.val, a.rhs, a.rg2.n_wg, nl, \
                     acc_doubles, a.opt_pf)
    if (a.opt_occ == 1) RDC_RG4(1); else RDC_RG4(2);
#undef RDC_RG4
    return hipGetLastError();
  }
  if constexpr (M::NELEM == 0 && M::AUX_LOCAL_NODE < 0 &&
                (std::is_same<M, PihnaNoCellTransportMoments>::value || std::is_same<M, PihnaNoCellTransport>::value))
  if (a.strategy == RDC_SCATTER_ROWGATHER && a.rg2.n_wg > 0 && a.rg2.pair_aux && a.rg2.nlist && a.rg2.block == 256 && a.opt_kernel == 6) {
    constexpr int BLOCK = 256;
    const int nl = a.rg2.nl_stride;
    const int acc_doubles = (int)((a.rg2.lds_bytes / sizeof(double) + 3) & ~(size_t)1);
    const size_t lds_bytes = sizeof(double) * ((size_t)acc_doubles + (size_t)(Rec<M>::N / 2) * nl * 2);
    const int wg_begin = a.rg2.wg_begin, wg_count = a.rg2.wg_count < 0 ? a.rg2.n_wg - a.rg2.wg_begin : a.rg2.wg_count;
    if (wg_count <= 0) return hipSuccess;
    int grid = a.opt_grid > 0 ? a.opt_grid : 512;   // two resident workgroups per CU
    if (grid > wg_count) grid = wg_count;
    hipLaunchKernelGGL((k_tet4_rg6<M, EXP_MODE, BLOCK>), dim3(grid), dim3(BLOCK), lds_bytes, a.stream, a.rg2.desc, a.rg2.pair_loc,
                       a.rg2.pair_aux, a.rg2.nlist, a.rg2.node_tab, k, a.packed, a.val, a.rhs, nl, acc_doubles, wg_begin,
                       wg_begin + wg_count);
    return hipGetLastError();
  }
  if (a.strategy == RDC_SCATTER_ROWGATHER && a.rg2.n_wg > 0 && a.rg2.pair_aux && a.rg2.nlist && a.rg2.block == 256 &&
      (a.opt_kernel == 0 || M::NELEM > 0 || M::AUX_LOCAL_NODE >= 0)) {
    constexpr int BLOCK = 256;
    const int nl = a.rg2.nl_stride;
    const int acc_doubles = (int)((a.rg2.lds_bytes / sizeof(double) + 3) & ~(size_t)1);  // + slice phase shift + diagonal alignment
    // + opt_ldspad KB of unused LDS: diagnostic, lowers the number of co-resident workgroups (one per CU from ~55 KB)
    const size_t lds_bytes = sizeof(double) * ((size_t)acc_doubles + (size_t)(Rec<M>::N / 2) * nl * 2) + (size_t)a.opt_ldspad * 1024;
    const int wg_begin = a.rg2.wg_begin, wg_count = a.rg2.wg_count < 0 ? a.rg2.n_wg - a.rg2.wg_begin : a.rg2.wg_count;
    if (wg_count <= 0) return hipSuccess;
#define RDC_RG5(MINW, ST)                                                                                          \
  hipLaunchKernelGGL((k_tet4_rg5<M, EXP_MODE, BLOCK, MINW, ST>), dim3(wg_count), dim3(BLOCK), lds_bytes, a.stream,     \
                     a.rg2.desc, a.rg2.pair_loc, a.rg2.pair_aux, a.rg2.nlist, a.rg2.node_tab, k, a.packed, a.val, a.rhs, \
                     nl, acc_doubles, a.stamps, a.opt_pf, a.opt_xcd, a.rg2.pair_eid, a.elem, wg_begin, a.opt_stagger)
    if constexpr ((std::is_same<M, PihnaNoCellTransport>::value || std::is_same<M, PihnaNoCellTransportMoments>::value) && EXP_MODE == 3) {
      // diagnostic builds (timing only, results are wrong): LDS atomics replaced by plain stores / removed
      if (a.opt_ablate >= 1 && a.opt_ablate <= 6) {
#define RDC_ABL(X)                                                                                                        \
  hipLaunchKernelGGL((k_tet4_rg5<M, EXP_MODE, BLOCK, 2, false, X>), dim3(wg_count), dim3(BLOCK), lds_bytes, a.stream,      \
                     a.rg2.desc, a.rg2.pair_loc, a.rg2.pair_aux, a.rg2.nlist, a.rg2.node_tab, k, a.packed, a.val, a.rhs, \
                     nl, acc_doubles, a.stamps, a.opt_pf, a.opt_xcd, a.rg2.pair_eid, a.elem, wg_begin, a.opt_stagger)
        switch (a.opt_ablate) {
          case 1: RDC_ABL(1); break;   // plain LDS stores instead of atomics
          case 2: RDC_ABL(2); break;   // no LDS accumulation traffic
          case 3: RDC_ABL(3); break;   // no compute phase
          case 4: RDC_ABL(4); break;   // no compute, no fold
          case 5: RDC_ABL(5); break;   // loads + zero + barriers only
          default: RDC_ABL(6); break;  // everything but the store phase
        }
#undef RDC_ABL
        return hipGetLastError();
      }
    }
    if constexpr (std::is_same<M, PihnaNoCellTransportSlim>::value || std::is_same<M, PihnaNoCellTransportMoments>::value) {
      if (a.opt_occ == 3) { RDC_RG5(3, false); return hipGetLastError(); }
    }
    if (a.stamps && (std::is_same<M, PihnaNoCellTransport>::value || std::is_same<M, PihnaNoCellTransportMoments>::value) && EXP_MODE == 3) RDC_RG5(2, true);
    else if (a.opt_occ == 1) RDC_RG5(1, false);
    else RDC_RG5(2, false);
#undef RDC_RG5
    return hipGetLastError();
  }
  if constexpr (M::NELEM == 0 && M::AUX_LOCAL_NODE < 0)
  if (a.strategy == RDC_SCATTER_ROWGATHER && a.rg2.n_wg > 0 && a.rg2.pair_aux && (a.opt_kernel == 0 || a.opt_kernel == 3)) {
#define RDC_RG3(BLOCK, MINW)                                                                                       \
  hipLaunchKernelGGL((k_tet4_rg3<M, EXP_MODE, BLOCK, MINW>), dim3(a.rg2.n_wg), dim3(BLOCK), a.rg2.lds_bytes, a.stream, \
                     a.rg2.desc, a.rg2.pair_rec, a.rg2.pair_aux, a.rg2.node_tab, k, a.packed, a.val, a.rhs, a.opt_xcd,  \
                     (long long*)nullptr)
    if (a.stamps && a.rg2.block == 256 && std::is_same<M, PihnaNoCellTransport>::value && EXP_MODE == 3) {
      // diagnostic build with s_memtime stamps per phase (tools/stamp_report.py); never the timed kernel
      hipLaunchKernelGGL((k_tet4_rg3<M, EXP_MODE, 256, 2, true>), dim3(a.rg2.n_wg), dim3(256), a.rg2.lds_bytes, a.stream,
                         a.rg2.desc, a.rg2.pair_rec, a.rg2.pair_aux, a.rg2.node_tab, k, a.packed, a.val, a.rhs, a.opt_xcd,
                         a.stamps);
      return hipGetLastError();
    }
    if (a.rg2.block == 128) {  // MINW counts waves per SIMD: 2 means four 128-thread workgroups per CU
      if (a.opt_occ == 1) RDC_RG3(128, 1); else RDC_RG3(128, 2);
    } else {
      if (a.opt_occ == 1) RDC_RG3(256, 1);
      else if (a.opt_occ == 3) RDC_RG3(256, 3);
      else RDC_RG3(256, 2);
    }
#undef RDC_RG3
    return hipGetLastError();
  }
  if constexpr (M::NELEM == 0 && M::AUX_LOCAL_NODE < 0)
  if (a.strategy == RDC_SCATTER_ROWGATHER && a.rg2.n_wg > 0 && a.opt_kernel == 2 && a.rg2.block == 256) {
    constexpr int BLOCK = 256;
#define RDC_RG2(MINW)                                                                                              \
  hipLaunchKernelGGL((k_tet4_rg2<M, EXP_MODE, BLOCK, MINW>), dim3(a.rg2.n_wg), dim3(BLOCK), 0, a.stream, a.rg2.desc, \
                     a.rg2.pair_rec, a.rg2.chunk, a.rg2.sdesc, a.rg2.contrib, k, a.packed, a.val, a.rhs, a.opt_ablate)
    if (a.opt_occ == 1) RDC_RG2(1);
    else if (a.opt_occ == 3) RDC_RG2(3);
    else RDC_RG2(2);
#undef RDC_RG2
    return hipGetLastError();
  }
  if constexpr (M::NELEM == 0 && M::AUX_LOCAL_NODE < 0)
  if (a.strategy == RDC_SCATTER_ROWGATHER) {
    constexpr int BLOCK = 256;
    if (a.n_wg > 0) {
#define RDC_RG(MINW, ABL)                                                                                          \
  hipLaunchKernelGGL((k_tet4_rowgather<M, EXP_MODE, BLOCK, MINW, ABL>), dim3(a.n_wg), dim3(BLOCK), a.lds_bytes, a.stream, \
                     a.m, k, a.packed, a.val, a.rhs)
      // tuning variants exist for the PIHNA / cubic-exponent instantiation only
      const bool lab = (std::is_same<M, Pihna>::value || std::is_same<M, PihnaNoCellTransport>::value) && EXP_MODE == 3;
      const int key = lab ? a.opt_occ * 10 + a.opt_ablate : 20;
      if (lab) {
        switch (key) {
          case 10: RDC_RG(1, 0); break;
          case 11: RDC_RG(1, 1); break;
          case 12: RDC_RG(1, 2); break;
          case 21: RDC_RG(2, 1); break;
          case 22: RDC_RG(2, 2); break;
          case 30: RDC_RG(3, 0); break;
          case 31: RDC_RG(3, 1); break;
          default: RDC_RG(2, 0); break;
        }
      } else {
        RDC_RG(2, 0);
      }
#undef RDC_RG
    }
    return hipGetLastError();
  }
  for (int c = 0; c < a.n_colours; c++) {
    const int64_t first = a.colour_ptr[c], count = a.colour_ptr[c + 1] - first;
    if (count <= 0) continue;
    const int64_t grid = (count + 255) / 256;
    hipLaunchKernelGGL((k_tet4_coloured<M, EXP_MODE>), dim3((unsigned)grid), dim3(256), 0, a.stream, a.m, k, first, count,
                       a.packed, a.elem, a.val, a.rhs);
  }
  return hipGetLastError();
}

template <class M>
hipError_t launch_tet4_fast(const LaunchArgs& a, const typename M::K& k) {
  if (a.exp_mode == M::FAST_EXP_MODE) return launch_fast_impl<M, M::FAST_EXP_MODE>(a, k);
  return launch_fast_impl<M, 0>(a, k);
}

template hipError_t launch_tet4_fast<Pihna>(const LaunchArgs&, const Pihna::K&);
template hipError_t launch_tet4_fast<PihnaNoCellTransport>(const LaunchArgs&, const PihnaNoCellTransport::K&);
template hipError_t launch_tet4_fast<PihnaNoCellTransportSlim>(const LaunchArgs&, const PihnaNoCellTransportSlim::K&);
template hipError_t launch_tet4_fast<PihnaNoCellTransportMoments>(const LaunchArgs&, const PihnaNoCellTransportMoments::K&);
template hipError_t launch_tet4_fast<Ripf>(const LaunchArgs&, const Ripf::K&);
template hipError_t launch_tet4_fast<RipfReduced>(const LaunchArgs&, const RipfReduced::K&);
template hipError_t launch_tet4_fast<Hcc>(const LaunchArgs&, const Hcc::K&);
template hipError_t launch_tet4_fast<HccMassOnly>(const LaunchArgs&, const HccMassOnly::K&);
template hipError_t launch_tet4_fast<Adpm>(const LaunchArgs&, const Adpm::K&);
template hipError_t launch_tet4_fast<AdpmDecayOnly>(const LaunchArgs&, const AdpmDecayOnly::K&);
template hipError_t launch_tet4_fast<Proteas>(const LaunchArgs&, const Proteas::K&);

}  // namespace rdc
