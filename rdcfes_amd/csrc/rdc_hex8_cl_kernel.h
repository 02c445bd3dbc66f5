// rdc_hex8_cl_kernel.h — the kernel of rdc_hex8_cl.h (HIP only; included by rdc_launch.h).
#ifndef RDC_HEX8_CL_KERNEL_H
#define RDC_HEX8_CL_KERNEL_H
#include "rdc_hex8_cl.h"
#include "rdc_internal.h"

namespace rdc {

// block (a, b) of the model is structurally non-zero (any of the A / B / D coefficient masks)
template <class M>
constexpr bool hex8_cl_block(int a, int b) {
  bool nz = M::hasA(a, b) || M::hasD(a, b);
  for (int g = 0; g < M::NG; g++) nz = nz || M::hasB(a, b, g);
  return nz;
}

template <class M, int EXP_MODE, int CW, int PW>
__global__ void __launch_bounds__((CW + PW) * 64, 2)
k_hex8_cl(const MeshDev m, const typename M::K k, const HostPrepCl::Desc* __restrict__ desc, const HostPrepCl::Node* __restrict__ ntab,
          const uint32_t* __restrict__ eid, const uint32_t* __restrict__ pair, const uint32_t* __restrict__ pslot,
          const double* __restrict__ u, const double* __restrict__ aux, const double* __restrict__ elem,
          double* __restrict__ val, double* __restrict__ rhs) {
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
  auto copy_out = [&]() {
    for (int a = tid >> 5; a < (int)d.nown; a += NT / 32) {
      const HostPrepCl::Node nd = ntab[(size_t)w * MAXN + a];
      const int nn = NV * NV * (int)nd.len;
      double* dst = val + (int64_t)(NV * NV) * nd.bptr;
      for (int x = tid & 31; x < nn; x += 32) __builtin_nontemporal_store(img[nd.off + x], dst + x);
      if ((tid & 31) < NV) rhs[(int64_t)NV * nd.node + (tid & 31)] = lrhs[NV * a + (tid & 31)];
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
    if (plive) hex8_cl_produce<M, EXP_MODE>(k, X, U, AX, ED, 0, lds + pl * R::STRIDE);
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < 8; q++) {   // one point ahead of the consumers
      if (plive && q + 1 < 8) hex8_cl_produce<M, EXP_MODE>(k, X, U, AX, ED, q + 1, lds + (((q + 1) & 1) * MAXE + pl) * R::STRIDE);
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
  __syncthreads();                  // producers: point 0
#pragma unroll 1
  for (int q = 0; q < 8; q++) {
    if (cvalid) hex8_cl_consume<M>(k, lds + ((q & 1) * MAXE + le) * R::STRIDE, q, li, acc, fe);
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

template <class M>
inline size_t hex8_cl_lds_bytes(int cw, int pw, size_t max_row_doubles) {
  const size_t points = (size_t)2 * pw * 64 * Hex8Rec<M>::STRIDE;
  const size_t image = ((max_row_doubles + 1) & ~(size_t)1) + (size_t)M::NV * cw * 8;
  return sizeof(double) * (points > image ? points : image);
}

template <class M, int EXP_MODE>
static hipError_t launch_hex8_cl(const LaunchArgs& a, const typename M::K& k) {
  constexpr int CW = 3, PW = 1;
  if (a.cl.cw != CW || a.cl.pw != PW) return hipErrorInvalidValue;
  const size_t bytes = hex8_cl_lds_bytes<M>(CW, PW, a.cl.max_row_doubles);
  static bool attr = false;  // per instantiation
  if (!attr) { (void)hipFuncSetAttribute((const void*)k_hex8_cl<M, EXP_MODE, CW, PW>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr = true; }
  hipLaunchKernelGGL((k_hex8_cl<M, EXP_MODE, CW, PW>), dim3(a.cl.n_wg), dim3((CW + PW) * 64), bytes, a.stream, a.m, k, a.cl.desc, a.cl.ntab,
                     a.cl.eid, a.cl.pair, a.cl.pslot, a.u, a.aux, a.elem, a.val, a.rhs);
  return hipGetLastError();
}

}  // namespace rdc
#endif
