// rdc_kernels.h — generic (any model x TET4/HEX8) HIP kernels of the assembly path.
//
//   rd_row           one row-node of the element matrices, all quadrature points   (device fn)
//   k_coloured       element-parallel, one launch per colour, plain RMW, no atomics
//   k_rowgather      row-owner gather: a workgroup owns a run of consecutive nodes, accumulates
//                    their complete CSR rows in LDS and streams them out once
//
// The reference loop these replace: src/pihna.C:383-756 (and its ripf / coupled_hcc twins).
#ifndef RDC_KERNELS_H
#define RDC_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "rdc_row.h"

namespace rdc {

// device view of the mesh / pattern (all pointers device memory)
struct MeshDev {
  int64_t n_elem, n_node, n_owned;
  const uint32_t* conn;       // [n_elem][NEN]
  const double* xyz;          // [n_node][3]
  const int64_t* bptr;        // [n_owned+1] node-block row pointer
  const uint16_t* eslot;      // [n_elem][NEN*NEN] slot of node j in the row of node i
  // coloured scatter
  const uint32_t* elem_order; // elements sorted by colour
  const uint64_t* first_mask; // [n_elem] bit (i*NEN+j): this element is the first writer of block (i,j)
  const uint8_t* first_rhs;   // [n_elem] bit i: first writer of the rhs entries of node i
  // row gather
  const uint32_t* pair_elem;  // [n_pairs] incident element of a (node, element) pair, grouped by node
  const uint8_t* pair_local;  // [n_pairs] local index of the node inside that element
  const int64_t* node_pair_ptr;  // [n_owned+1]
  const int32_t* wg_node_ptr;    // [n_wg+1] consecutive owned nodes per workgroup
};

template <class M, int NEN>
__device__ __forceinline__ void load_element(const MeshDev& m, int64_t e, const double* __restrict__ u,
                                             const double* __restrict__ aux, uint32_t (&nd)[NEN],
                                             double (&X)[NEN][3], double (&U)[NEN][M::NV],
                                             double (&AX)[NEN][M::NAUX > 0 ? M::NAUX : 1]) {
#pragma unroll
  for (int i = 0; i < NEN; i++) nd[i] = m.conn[e * NEN + i];
#pragma unroll
  for (int i = 0; i < NEN; i++) {
    const int64_t n = nd[i];
#pragma unroll
    for (int d = 0; d < 3; d++) X[i][d] = m.xyz[3 * n + d];
#pragma unroll
    for (int v = 0; v < M::NV; v++) U[i][v] = u[n * M::NV + v];
    if (M::NAUX > 0) {
#pragma unroll
      for (int v = 0; v < M::NAUX; v++)  // PROTEAS reads its aux field at one local node only (AUX_LOCAL_NODE)
        AX[i][v] = (M::AUX_LOCAL_NODE < 0 || i == M::AUX_LOCAL_NODE) ? aux[n * M::NAUX + v] : 0.0;
    } else {
      AX[i][0] = 0.0;
    }
  }
}

// ---- coloured scatter: elements [first, first+count) of elem_order share no node ----------
template <class M, int NEN, int EXP_MODE>
__global__ void __launch_bounds__(256)
k_coloured(const MeshDev m, const typename M::K k, int64_t first, int64_t count,
           const double* __restrict__ u, const double* __restrict__ aux, const double* __restrict__ elem,
           double* __restrict__ val, double* __restrict__ rhs) {
  constexpr int NV = M::NV;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int64_t e = m.elem_order[first + t];
  uint32_t nd[NEN];
  double X[NEN][3], U[NEN][NV], AX[NEN][M::NAUX > 0 ? M::NAUX : 1];
  load_element<M, NEN>(m, e, u, aux, nd, X, U, AX);
  const uint64_t fm = m.first_mask[e];
  const uint32_t fr = m.first_rhs[e];
#pragma unroll 1
  for (int i = 0; i < NEN; i++) {
    const int64_t I = m.conn[e * NEN + i];
    if (I >= m.n_owned) continue;  // row assembled by the owner partition
    double acc[NV][NV][NEN], fe[NV];
    rd_row<M, NEN, EXP_MODE>(k, X, U, AX, i, acc, fe, M::NELEM > 0 ? elem + e * M::NELEM : nullptr);
    const int64_t b0 = m.bptr[I];
    const int64_t len = m.bptr[I + 1] - b0;
    double* row = val + (int64_t)NV * NV * b0;
#pragma unroll
    for (int j = 0; j < NEN; j++) {
      const int64_t s = m.eslot[e * (NEN * NEN) + i * NEN + j];
      const bool first_writer = (fm >> (i * NEN + j)) & 1ull;
#pragma unroll
      for (int a = 0; a < NV; a++)
#pragma unroll
        for (int b = 0; b < NV; b++) {
          double* p = row + a * NV * len + NV * s + b;
          *p = first_writer ? acc[a][b][j] : (*p + acc[a][b][j]);
        }
    }
    const bool first_r = (fr >> i) & 1u;
#pragma unroll
    for (int a = 0; a < NV; a++) {
      double* p = rhs + I * NV + a;
      *p = first_r ? fe[a] : (*p + fe[a]);
    }
  }
}

// one equation row A of a (node, element) pair over all quadrature points, added into the LDS row slice
template <class M, int NEN, int EXP_MODE, int A>
__device__ __forceinline__ void rowsplit_row(const MeshDev& m, const typename M::K& k, const double (&X)[NEN][3],
                                             const double (&U)[NEN][M::NV], const double (&AX)[NEN][M::NAUX > 0 ? M::NAUX : 1],
                                             int64_t e, int i, const double* ED, double* row, int len, double* lrhs) {
  constexpr int NV = M::NV;
  double acc[NV][NEN], fe = 0.0;
#pragma unroll
  for (int b = 0; b < NV; b++)
#pragma unroll
    for (int j = 0; j < NEN; j++) acc[b][j] = 0.0;
#pragma unroll 1
  for (int q = 0; q < Ref<NEN>::NQP; q++) {
    RowPoint<M, NEN> P;
    rd_point_setup<M, NEN, EXP_MODE>(k, X, U, AX, q, i, ED, P);
    rd_point_accum_row<M, NEN, A>(P, acc, fe);
  }
#pragma unroll
  for (int j = 0; j < NEN; j++) {
    const int s = m.eslot[e * (NEN * NEN) + i * NEN + j];
#pragma unroll
    for (int b = 0; b < NV; b++)
      __hip_atomic_fetch_add(row + A * NV * len + NV * s + b, acc[b][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __hip_atomic_fetch_add(lrhs + A, fe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <class M, int NEN, int EXP_MODE, int... As>
__device__ __forceinline__ void rowsplit_rows(std::integer_sequence<int, As...>, const MeshDev& m, const typename M::K& k,
                                              const double (&X)[NEN][3], const double (&U)[NEN][M::NV],
                                              const double (&AX)[NEN][M::NAUX > 0 ? M::NAUX : 1], int64_t e, int i,
                                              const double* ED, double* row, int len, double* lrhs) {
  (rowsplit_row<M, NEN, EXP_MODE, As>(m, k, X, U, AX, e, i, ED, row, len, lrhs), ...);
}

// ---- row gather ---------------------------------------------------------------------------
// One workgroup owns the consecutive owned nodes [wg_node_ptr[w], wg_node_ptr[w+1]); their CSR
// rows form ONE contiguous slice of val[].  One thread per (node, incident element) pair
// evaluates the node's row of that element and adds it into the slice held in LDS; the slice is
// then written once with coalesced streaming stores.  Every HBM byte of the matrix is written
// exactly once, nothing is read back, no global atomics, no colours.
template <class M, int NEN, int EXP_MODE, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_rowgather(const MeshDev m, const typename M::K k, const double* __restrict__ u,
            const double* __restrict__ aux, const double* __restrict__ elem, double* __restrict__ val,
            double* __restrict__ rhs) {
  constexpr int NV = M::NV;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int w = blockIdx.x;
  const int64_t n0 = m.wg_node_ptr[w], n1 = m.wg_node_ptr[w + 1];
  const int64_t vb0 = (int64_t)NV * NV * m.bptr[n0];
  const int nval = (int)((int64_t)NV * NV * m.bptr[n1] - vb0);
  const int nrhs = (int)(n1 - n0) * NV;
  // the slice starts one double into LDS when its first CSR value has an odd index: LDS and global memory then share
  // the 16-byte phase, and the zero / copy-out phases move 16 bytes per lane (the launch adds 16 bytes of LDS)
  const int sh = (int)(vb0 & 1);
  double* const sl = lds + sh;
  double* lrhs = sl + nval;
  {
    double2* z = reinterpret_cast<double2*>(lds);
    for (int x = threadIdx.x; x < (sh + nval + nrhs + 1) / 2; x += BLOCK) z[x] = make_double2(0.0, 0.0);
  }
  __syncthreads();
  const int64_t p0 = m.node_pair_ptr[n0], p1 = m.node_pair_ptr[n1];
  // The pairs of one node are consecutive in the list and all add into the same diagonal block: consecutive
  // lanes on consecutive pairs would serialise every ds_add_f64 on that address (8-way on HEX8, 24-way on
  // TET4; tools/lds_atomic_bench.hip).  Transposed lane mapping: the 16 lanes of an LDS pass take pairs 8
  // apart, i.e. (mostly) different row nodes, whose rows never share an address.
  constexpr int GRP = BLOCK / 8;
  const int64_t tperm = (int64_t)(threadIdx.x % GRP) * 8 + threadIdx.x / GRP;
  for (int64_t p = p0 + tperm; p < p1; p += BLOCK) {
    const int64_t e = m.pair_elem[p];
    const int i = m.pair_local[p];
    uint32_t nd[NEN];
    double X[NEN][3], U[NEN][NV], AX[NEN][M::NAUX > 0 ? M::NAUX : 1];
    load_element<M, NEN>(m, e, u, aux, nd, X, U, AX);
    const double* ED = M::NELEM > 0 ? elem + e * M::NELEM : nullptr;
    const int64_t I = m.conn[e * NEN + i];
    const int64_t b0 = m.bptr[I];
    const int len = (int)(m.bptr[I + 1] - b0);
    double* row = sl + ((int64_t)NV * NV * b0 - vb0);
    if constexpr (NV * NV * NEN > 128) {
      // The full NV x NV x NEN accumulator (200 doubles for five unknowns on HEX8) does not fit the register file
      // (hundreds of bytes of scratch per lane, 39 ms for 0.5 M hexes): one equation row at a time instead -- the
      // per-point set-up is redone NV times, everything stays in registers.
      rowsplit_rows<M, NEN, EXP_MODE>(std::make_integer_sequence<int, NV>{}, m, k, X, U, AX, e, i, ED, row, len,
                                      lrhs + (I - n0) * NV);
      continue;
    }
    double acc[NV][NV][NEN], fe[NV];
    rd_row<M, NEN, EXP_MODE>(k, X, U, AX, i, acc, fe, ED);
#pragma unroll
    for (int j = 0; j < NEN; j++) {
      const int s = m.eslot[e * (NEN * NEN) + i * NEN + j];
#pragma unroll
      for (int a = 0; a < NV; a++)
#pragma unroll
        for (int b = 0; b < NV; b++)
          __hip_atomic_fetch_add(row + a * NV * len + NV * s + b, acc[a][b][j], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int a = 0; a < NV; a++)
      __hip_atomic_fetch_add(lrhs + (I - n0) * NV + a, fe[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  double* out = val + vb0;  // out[g] <-> sl[g]; out + sh and lds + 2 * sh are 16-byte aligned
  {
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    const int npair = (nval - sh) >> 1;
    const v2d_t* src = reinterpret_cast<const v2d_t*>(lds + 2 * sh);
    v2d_t* dst = reinterpret_cast<v2d_t*>(out + sh);
    for (int x = threadIdx.x; x < npair; x += BLOCK) __builtin_nontemporal_store(src[x], dst + x);
    if (sh && threadIdx.x == 0) out[0] = sl[0];
    if (((nval - sh) & 1) && threadIdx.x == 64 % BLOCK) out[nval - 1] = sl[nval - 1];
  }
  double* orhs = rhs + n0 * NV;
  for (int x = threadIdx.x; x < nrhs; x += BLOCK) orhs[x] = lrhs[x];
}

// ---- node-staged row gather (HEX8) ------------------------------------------------------------------
// Same decomposition as k_rowgather, but the coordinates / unknowns / aux values of the workgroup's distinct
// nodes are first copied into LDS ([node][3 + NV + NAUX] doubles) and every quadrature point re-reads the eight
// nodes of its element from there instead of holding them in registers for the whole loop (96 VGPRs on HEX8 with
// three unknowns): the generic evaluator then fits two waves per SIMD.
template <class M, int NEN, int EXP_MODE, int BLOCK>
__global__ void __launch_bounds__(BLOCK, 2)
k_rowgather_staged(const MeshDev m, const typename M::K k, const double* __restrict__ u, const double* __restrict__ aux,
                   const double* __restrict__ elem, const int64_t* __restrict__ nl_ptr, const uint32_t* __restrict__ nlist,
                   const uint16_t* __restrict__ ploc, const int tab_off, double* __restrict__ val,
                   double* __restrict__ rhs) {
  constexpr int NV = M::NV, NA = (M::NAUX > 0 ? M::NAUX : 1), REC = 3 + NV + (M::NAUX > 0 ? M::NAUX : 0);
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int w = blockIdx.x;
  const int64_t n0 = m.wg_node_ptr[w], n1 = m.wg_node_ptr[w + 1];
  const int64_t vb0 = (int64_t)NV * NV * m.bptr[n0];
  const int nval = (int)((int64_t)NV * NV * m.bptr[n1] - vb0);
  const int nrhs = (int)(n1 - n0) * NV;
  const int sh = (int)(vb0 & 1);  // see k_rowgather
  double* const sl = lds + sh;
  double* lrhs = sl + nval;
  double* tab = lds + tab_off;
  {
    double2* z = reinterpret_cast<double2*>(lds);
    for (int x = threadIdx.x; x < (sh + nval + nrhs + 1) / 2; x += BLOCK) z[x] = make_double2(0.0, 0.0);
  }
  {
    const int64_t l0 = nl_ptr[w];
    const int nn = (int)(nl_ptr[w + 1] - l0);
    for (int x = threadIdx.x; x < nn; x += BLOCK) {
      const int64_t n = nlist[l0 + x];
      double* r = tab + x * REC;
      r[0] = m.xyz[3 * n]; r[1] = m.xyz[3 * n + 1]; r[2] = m.xyz[3 * n + 2];
#pragma unroll
      for (int v = 0; v < NV; v++) r[3 + v] = u[n * NV + v];
      if (M::NAUX > 0) {
#pragma unroll
        for (int v = 0; v < M::NAUX; v++) r[3 + NV + v] = aux[n * M::NAUX + v];
      }
    }
  }
  __syncthreads();
  const int64_t p0 = m.node_pair_ptr[n0], p1 = m.node_pair_ptr[n1];
  constexpr int GRP = BLOCK / 8;
  const int64_t tperm = (int64_t)(threadIdx.x % GRP) * 8 + threadIdx.x / GRP;  // see k_rowgather
  for (int64_t p = p0 + tperm; p < p1; p += BLOCK) {
    const int64_t e = m.pair_elem[p];
    const int i = m.pair_local[p];
    int li[NEN];
    {
      const uint4 a0 = reinterpret_cast<const uint4*>(ploc)[p * NEN / 8];
      li[0] = a0.x & 0xFFFF; li[1] = a0.x >> 16; li[2] = a0.y & 0xFFFF; li[3] = a0.y >> 16;
      if (NEN == 8) { li[4 % NEN] = a0.z & 0xFFFF; li[5 % NEN] = a0.z >> 16; li[6 % NEN] = a0.w & 0xFFFF; li[7 % NEN] = a0.w >> 16; }
    }
    double acc[NV][NV][NEN], fe[NV];
    rd_row_zero<M, NEN>(acc, fe);
#pragma unroll 1
    for (int q = 0; q < Ref<NEN>::NQP; q++) {
      int off = 0;
      asm volatile("" : "+v"(off));  // opaque per iteration: keeps the node reads inside the loop (not hoisted back into registers)
      double X[NEN][3], U[NEN][NV], AX[NEN][NA];
#pragma unroll
      for (int l = 0; l < NEN; l++) {
        const double* r = tab + (li[l] * REC + off);
        X[l][0] = r[0]; X[l][1] = r[1]; X[l][2] = r[2];
#pragma unroll
        for (int v = 0; v < NV; v++) U[l][v] = r[3 + v];
        if (M::NAUX > 0) {
#pragma unroll
          for (int v = 0; v < M::NAUX; v++) AX[l][v] = (M::AUX_LOCAL_NODE < 0 || l == M::AUX_LOCAL_NODE) ? r[3 + NV + v] : 0.0;
        } else {
          AX[l][0] = 0.0;
        }
      }
      rd_row_point<M, NEN, EXP_MODE>(k, X, U, AX, q, i, acc, fe, M::NELEM > 0 ? elem + e * M::NELEM : nullptr);
    }
    const int64_t I = m.conn[e * NEN + i];
    const int64_t b0 = m.bptr[I];
    const int len = (int)(m.bptr[I + 1] - b0);
    double* row = sl + ((int64_t)NV * NV * b0 - vb0);
#pragma unroll
    for (int j = 0; j < NEN; j++) {
      const int s = m.eslot[e * (NEN * NEN) + i * NEN + j];
#pragma unroll
      for (int a = 0; a < NV; a++)
#pragma unroll
        for (int b = 0; b < NV; b++)
          __hip_atomic_fetch_add(row + a * NV * len + NV * s + b, acc[a][b][j], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int a = 0; a < NV; a++)
      __hip_atomic_fetch_add(lrhs + (I - n0) * NV + a, fe[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  double* out = val + vb0;  // out[g] <-> sl[g]; out + sh and lds + 2 * sh are 16-byte aligned
  {
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    const int npair = (nval - sh) >> 1;
    const v2d_t* src = reinterpret_cast<const v2d_t*>(lds + 2 * sh);
    v2d_t* dst = reinterpret_cast<v2d_t*>(out + sh);
    for (int x = threadIdx.x; x < npair; x += BLOCK) __builtin_nontemporal_store(src[x], dst + x);
    if (sh && threadIdx.x == 0) out[0] = sl[0];
    if (((nval - sh) & 1) && threadIdx.x == 64 % BLOCK) out[nval - 1] = sl[nval - 1];
  }
  double* orhs = rhs + n0 * NV;
  for (int x = threadIdx.x; x < nrhs; x += BLOCK) orhs[x] = lrhs[x];
}

// PIHNA save_solution volume sums (src/pihna.C:898-958): per-workgroup partial sums -> part[blockIdx.x][4]
template <int NEN>
static __global__ void __launch_bounds__(256)
k_pihna_volumes(const MeshDev m, int64_t n_elem, const double* __restrict__ u, const rdc_pihna_ranges r,
                double* __restrict__ part) {
  __shared__ double red[4][256];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_elem; e += (int64_t)gridDim.x * 256) {
    double X[NEN][3];
    bool ok[4] = {true, true, true, true};
#pragma unroll
    for (int i = 0; i < NEN; i++) {
      const int64_t n = m.conn[e * NEN + i];
#pragma unroll
      for (int d = 0; d < 3; d++) X[i][d] = m.xyz[3 * n + d];
      const double n_ = u[5 * n], c_ = u[5 * n + 1], h_ = u[5 * n + 2], v_ = u[5 * n + 3];
      const double ch = c_ + h_, T_ = (n_ + c_ + h_ + v_) / r.cells_max_capacity;
      ok[0] = ok[0] && (ch >= r.active_tumor_min && ch <= r.active_tumor_max);
      ok[1] = ok[1] && (n_ >= r.necrotic_min && n_ <= r.necrotic_max);
      ok[2] = ok[2] && (v_ >= r.vascularity_min && v_ <= r.vascularity_max);
      ok[3] = ok[3] && (T_ >= r.total_cell_min && T_ <= r.total_cell_max);
    }
    double vol = 0.0;  // elem->volume(): sum of JxW (exact for TET4 and for the trilinear HEX8 map)
#pragma unroll 1
    for (int q = 0; q < Ref<NEN>::NQP; q++) {
      double N[NEN], G[NEN][3], W;
      fe_point<NEN>(X, q, N, G, W);
      vol += W;
    }
#pragma unroll
    for (int x = 0; x < 4; x++) acc[x] += ok[x] ? vol : 0.0;
  }
#pragma unroll
  for (int x = 0; x < 4; x++) red[x][threadIdx.x] = acc[x];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
#pragma unroll
      for (int x = 0; x < 4; x++) red[x][threadIdx.x] += red[x][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < 4) part[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// RIPF save_solution volume sums (src/ripf.C:812-858): per-workgroup partial sums -> part[blockIdx.x][2]
template <int NEN>
static __global__ void __launch_bounds__(256)
k_ripf_volumes(const MeshDev m, int64_t n_elem, const double* __restrict__ u, const rdc_ripf_ranges r,
               double* __restrict__ part) {
  __shared__ double red[2][256];
  double acc[2] = {0.0, 0.0};
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_elem; e += (int64_t)gridDim.x * 256) {
    double X[NEN][3];
    bool ok[2] = {true, true};
#pragma unroll
    for (int i = 0; i < NEN; i++) {
      const int64_t n = m.conn[e * NEN + i];
#pragma unroll
      for (int d = 0; d < 3; d++) X[i][d] = m.xyz[3 * n + d];
      const double HU = u[3 * n], cc = u[3 * n + 1], fb = u[3 * n + 2];
      ok[0] = ok[0] && (HU >= r.cc_HU_min && HU <= r.cc_HU_max && cc >= r.cc_min);
      ok[1] = ok[1] && (HU >= r.fb_HU_min && HU <= r.fb_HU_max && fb >= r.fb_min);
    }
    double vol = 0.0;
#pragma unroll 1
    for (int q = 0; q < Ref<NEN>::NQP; q++) {
      double N[NEN], G[NEN][3], W;
      fe_point<NEN>(X, q, N, G, W);
      vol += W;
    }
    acc[0] += ok[0] ? vol : 0.0;
    acc[1] += ok[1] ? vol : 0.0;
  }
  red[0][threadIdx.x] = acc[0];
  red[1][threadIdx.x] = acc[1];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x < 2) part[blockIdx.x * 2 + threadIdx.x] = red[threadIdx.x][0];
}

// ADPM save_solution (src/adpm.C:747-813).  slot[e] = index of the element's subdomain in the parcellation (-1: not
// listed), bit 30 set on the last element of its region: that one writes the region's concentrations (upstream
// assigns, so the last element wins).  Thresholded volumes: LDS sums per region -> part[blockIdx.x][n_ids][2].
template <int NEN>
static __global__ void __launch_bounds__(256)
k_adpm_parcellation(const MeshDev m, int64_t n_elem, const double* __restrict__ u, const rdc_adpm_ranges r,
                    const int32_t* __restrict__ slot, int n_ids, double* __restrict__ part, double* __restrict__ conc) {
  extern __shared__ double sums[];  // [n_ids][2]
  for (int i = threadIdx.x; i < 2 * n_ids; i += 256) sums[i] = 0.0;
  __syncthreads();
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_elem; e += (int64_t)gridDim.x * 256) {
    const int32_t sf = slot[e];
    if (sf < 0) continue;
    const int sl = sf & 0x3fffffff;
    double X[NEN][3], Ab[NEN], Ta[NEN];
    bool okA = true, okT = true;
#pragma unroll
    for (int i = 0; i < NEN; i++) {
      const int64_t n = m.conn[e * NEN + i];
#pragma unroll
      for (int d = 0; d < 3; d++) X[i][d] = m.xyz[3 * n + d];
      Ab[i] = u[3 * n + 1];
      Ta[i] = u[3 * n + 2];
      okA = okA && (Ab[i] >= r.A_b_min && Ab[i] <= r.A_b_max);
      okT = okT && (Ta[i] >= r.Tau_min && Ta[i] <= r.Tau_max);
    }
    double vol = 0.0, avA = 0.0, avT = 0.0;
#pragma unroll 1
    for (int q = 0; q < Ref<NEN>::NQP; q++) {
      double N[NEN], G[NEN][3], W;
      fe_point<NEN>(X, q, N, G, W);
      double a = 0.0, t = 0.0;
#pragma unroll
      for (int l = 0; l < NEN; l++) { a += N[l] * Ab[l]; t += N[l] * Ta[l]; }
      vol += W; avA += W * a; avT += W * t;
    }
    if (okA) atomicAdd(&sums[2 * sl], vol);
    if (okT) atomicAdd(&sums[2 * sl + 1], vol);
    if (sf & 0x40000000) { conc[2 * sl] = avA / vol; conc[2 * sl + 1] = avT / vol; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * n_ids; i += 256) part[(size_t)blockIdx.x * 2 * n_ids + i] = sums[i];
}

// RIPF check_solution (src/ripf.C:675-775): clamp, time-derivative system, fractionation schedule, prev := unclamped,
// aux record of the next assembly; per-workgroup maxima of the total dose go to wg_max[blockIdx.x].
static __global__ void __launch_bounds__(256)
k_ripf_check(int64_t n, double dt_r, double HU_min, double HU_max, double broad_frac, double focus_frac, int day,
             double* __restrict__ sol, double* __restrict__ prev, double* __restrict__ td, double* __restrict__ rt,
             double* __restrict__ aux, double* __restrict__ wg_max) {
  __shared__ double red[256];
  double mx = -1.0;  // :705
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double s0 = sol[3 * i], s1 = sol[3 * i + 1], s2 = sol[3 * i + 2];
    double HU = s0, cc = s1, fb = s2;
    if (HU < HU_min) HU = HU_min; else if (HU > HU_max) HU = HU_max;  // :722
    if (cc < 0.0) cc = 0.0;
    if (fb < 0.0) fb = 0.0;
    const double t0 = (HU - prev[3 * i]) * dt_r, t1 = (cc - prev[3 * i + 1]) * dt_r, t2 = (fb - prev[3 * i + 2]) * dt_r;
    sol[3 * i] = HU; sol[3 * i + 1] = cc; sol[3 * i + 2] = fb;
    td[3 * i] = t0; td[3 * i + 1] = t1; td[3 * i + 2] = t2;
    prev[3 * i] = s0; prev[3 * i + 1] = s1; prev[3 * i + 2] = s2;      // :769 copies the unclamped global solution
    const double rb = rt[3 * i], rf = rt[3 * i + 1];
    const double total_frac = broad_frac + focus_frac;
    double tot;
    if (day < broad_frac) tot = rb / broad_frac * (day + 1);
    else if (day < total_frac) tot = rf / focus_frac * ((day + 1) - broad_frac) + rb;
    else tot = rb + rf;
    rt[3 * i + 2] = tot;
    aux[3 * i] = t1; aux[3 * i + 1] = t2; aux[3 * i + 2] = tot;
    mx = tot > mx ? tot : mx;                                           // std::max(RT_total_max, RT_total_), :761
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x + s] > red[threadIdx.x] ? red[threadIdx.x + s] : red[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) wg_max[blockIdx.x] = red[0];
}

// check_solution clamp (src/pihna.C:785-790), in place
static __global__ void k_clamp_nonnegative(double* __restrict__ u, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double x = u[i];
    if (x < 0.0) u[i] = 0.0;
  }
}

}  // namespace rdc
#endif
