// rdc_solid_cl.hip — fused residual + tangent kernel of the SolidSystem on HEX8 (default for tangent requests).
//
// Replaces SolidSystem::element_time_derivative (src/solid_system.C:146-271) with the Hyperelastic law
// (src/hyperelastic.h:25-87, src/hyperlastic_inline.h:3-189); algebra in the header of rdc_solid.hip.
//
// The two-pass form (rdc_solid.hip) writes every element matrix to HBM and reads it back (2 x 9.2 GB on H(126), half of
// its 7.2 ms).  Here the element matrices never leave the CU:
//   * a workgroup owns a CLUSTER of <= CW * 8 owned nodes (rdc_prep_cl.cpp) and produces their CSR rows completely;
//   * CW consumer waves: one lane per (owned node, incident element) pair, 72 accumulators = that row of the element
//     matrix, summed over the eight quadrature points;
//   * PW producer waves: one lane per element touching the cluster; per quadrature point it evaluates the shape
//     gradients, F, M, Q, sigma ONCE for all the pairs of the element and hands the 47 doubles over through LDS.
//     The roles are split by WAVE so that the producers' working set (coordinates of eight nodes, the 3 x 3 chain) does
//     not compete with the 144 accumulator registers of the consumers; both fit 256 VGPRs, two workgroups per CU.
//     The point buffer is double-buffered: producers fill point q + 1 while consumers accumulate point q, one
//     workgroup barrier per point;
//   * epilogue: the consumers add their 72 values into an LDS image of the CSR rows of all owned nodes (ds_add_f64;
//     the image overlays the point buffers; pair order chosen on the host so that the 16 lanes of an LDS pass hit
//     different rows), and the image leaves as one run of 9 * len consecutive doubles per node.  Every CSR value is
//     written exactly once, no global atomics, nothing is read back.
// The summation order inside a node block depends on the LDS atomics' arrival order: results are reproducible to
// rounding, not bitwise (the two-pass form, solid_kernel = 2, is).
#include "rdc_solid.h"

namespace rdc {

namespace {
constexpr int PSTRIDE = 49;  // 47 doubles per (element, point) record; odd stride => records of different elements spread over the banks
typedef double cl_v2d __attribute__((ext_vector_type(2)));
}

template <int CW, int PW, bool SYM>
__global__ void __launch_bounds__((CW + PW) * 64, (CW + PW) * 64 <= 256 ? 2 : 1)
k_solid_cl(const MeshDev m, const HostPrepCl::Desc* __restrict__ desc, const HostPrepCl::Node* __restrict__ ntab,
           const uint32_t* __restrict__ eid, const uint32_t* __restrict__ pair, const uint32_t* __restrict__ pslot,
           const double* __restrict__ Xu, const double* __restrict__ fibre, const int32_t* __restrict__ elem_material,
           const rdc_solid_material* __restrict__ materials, const double pseudo_time, double* __restrict__ val,
           double* __restrict__ rhs, const int diag /* timing diagnostics (tools/solid_ab.py), bit mask: 1 = consumers idle, 2 = producers idle, 4 = image not zeroed, 8 = no atomics, 16 = no copy-out, 32 = no element loads; 64 = copy-out with 8-byte stores (results unchanged) */) {
  constexpr int MAXP = CW * 64, MAXE = PW * 64, MAXN = CW * 8, NT = (CW + PW) * 64;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int w = blockIdx.x;
  // Roles rotate over the waves from workgroup to workgroup: the hardware places wave i of every workgroup on SIMD i,
  // so with fixed roles the producers of the two workgroups of a CU (the longer instruction stream) would share a SIMD.
  constexpr int NW = CW + PW;
  const int tid = (int)(((threadIdx.x >> 6) + ((blockIdx.x >> 3) % NW)) % NW) * 64 + (int)(threadIdx.x & 63);
  const HostPrepCl::Desc d = desc[w];
  const bool producer = tid >= MAXP;
  const int nimg = (int)d.row_doubles;
  double* const img = lds;
  double* const lrhs = lds + ((nimg + 1) & ~1);
  // the image of the CSR rows of all owned nodes overlays the point buffers once the points are consumed
  auto zero_image = [&]() {
    cl_v2d* z = reinterpret_cast<cl_v2d*>(lds);
    const cl_v2d zero = {0.0, 0.0};
    for (int x = tid; x < (((nimg + 1) & ~1) + 3 * (int)d.nown + 1) / 2; x += NT) z[x] = zero;
  };
  // a half-wave per node: its three rows are 9 * len consecutive doubles of the CSR array, written with 16-byte stores (the
  // node's image segment has the 16-byte phase of its CSR segment: rdc_prep_cl.cpp)
  auto copy_out = [&]() {
    for (int a = tid >> 5; a < (int)d.nown; a += NT / 32) {
      const HostPrepCl::Node nd = ntab[(size_t)w * MAXN + a];
      const int n9 = 9 * (int)nd.len, l32 = tid & 31;
      double* dst = val + 9 * (int64_t)nd.bptr;
      const double* src = img + nd.off;
      const int sh = (int)(nd.off & 1), npair = (n9 - sh) >> 1;
      const cl_v2d* s2 = reinterpret_cast<const cl_v2d*>(src + sh);
      cl_v2d* d2 = reinterpret_cast<cl_v2d*>(dst + sh);
      if (diag & 64) {   // 8-byte stores (timing comparison; same values)
        for (int k = l32; k < n9; k += 32) __builtin_nontemporal_store(src[k], dst + k);
      } else {
        for (int k = l32; k < npair; k += 32) __builtin_nontemporal_store(s2[k], d2 + k);
        if (sh && l32 == 0) __builtin_nontemporal_store(src[0], dst);
        if (((n9 - sh) & 1) && l32 == 1) __builtin_nontemporal_store(src[n9 - 1], dst + n9 - 1);
      }
      if (l32 < 3) rhs[3 * (int64_t)nd.node + l32] = lrhs[3 * a + l32];
    }
  };
  // workgroup barrier that orders LDS accesses only (no wait for the global stores in flight)
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // The two roles are separate code paths with the SAME sequence of workgroup barriers (the branch is uniform per wave),
  // so that the register allocator never has to hold the consumers' accumulators and the producers' element together.
  if (producer) {
    // ================= producer: the element of this lane, loaded once ======================================================
    const int pl = tid - MAXP;
    const uint32_t e = eid[(size_t)w * MAXE + pl];
    const bool plive = e != 0xFFFFFFFFu;
    double X[8][3], XU[8][3];
    double mu = 0.0, lame = 0.0, Kf = 0.0, A[3] = {0.0, 0.0, 0.0}, lam[3] = {1.0, 1.0, 1.0}, rlam = 1.0;
    if (plive && !(diag & 32)) {
#pragma unroll
      for (int n = 0; n < 8; n++) {
        const int64_t I = m.conn[(int64_t)e * 8 + n];
#pragma unroll
        for (int c = 0; c < 3; c++) { X[n][c] = m.xyz[3 * I + c]; XU[n][c] = Xu[3 * I + c]; }
      }
      const rdc_solid_material mat = materials[elem_material[e]];   // src/solid_system.C:183-190
      mu = 0.5 * mat.Young / (1.0 + mat.Poisson);                   // hyperlastic_inline.h:21-24
      lame = mat.Young * mat.Poisson / ((1.0 + mat.Poisson) * (1.0 - 2.0 * mat.Poisson));
      const double K = mat.FibreStiffness;
      if (K > 0.0) {                                                // hyperelastic.h:46
        const double f0 = fibre[3 * (int64_t)e], f1 = fibre[3 * (int64_t)e + 1], f2 = fibre[3 * (int64_t)e + 2];
        const double nrm = sqrt(f0 * f0 + f1 * f1 + f2 * f2);
        A[0] = f0 / nrm; A[1] = f1 / nrm; A[2] = f2 / nrm;
        Kf = K;
      }
#pragma unroll
      for (int c = 0; c < 3; c++) lam[c] = 1.0 + pseudo_time * mat.rate[c];  // solid_system.C:232-234
      rlam = 1.0 / (lam[0] * lam[1] * lam[2]);
    }
    // one quadrature point of the element -> record in buffer b
    auto produce = [&](int q, int b) {
      double N[8], G[8][3], W;
      fe_point<8>(X, q, N, G, W);
      double gX[3][3];  // gradX[d][c] = sum_l dphi_l[c] * X_l[d], solid_system.C:221-229
#pragma unroll
      for (int dd = 0; dd < 3; dd++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
          double s = 0.0;
#pragma unroll
          for (int l = 0; l < 8; l++) s += G[l][c] * XU[l][dd];
          gX[dd][c] = s;
        }
      // F = gradX^-1
      const double c00 = gX[1][1] * gX[2][2] - gX[1][2] * gX[2][1];
      const double c01 = gX[1][2] * gX[2][0] - gX[1][0] * gX[2][2];
      const double c02 = gX[1][0] * gX[2][1] - gX[1][1] * gX[2][0];
      const double s = 1.0 / (gX[0][0] * c00 + gX[0][1] * c01 + gX[0][2] * c02);
      double F[3][3];
      F[0][0] = c00 * s;
      F[0][1] = (gX[0][2] * gX[2][1] - gX[0][1] * gX[2][2]) * s;
      F[0][2] = (gX[0][1] * gX[1][2] - gX[0][2] * gX[1][1]) * s;
      F[1][0] = c01 * s;
      F[1][1] = (gX[0][0] * gX[2][2] - gX[0][2] * gX[2][0]) * s;
      F[1][2] = (gX[0][2] * gX[1][0] - gX[0][0] * gX[1][2]) * s;
      F[2][0] = c02 * s;
      F[2][1] = (gX[0][1] * gX[2][0] - gX[0][0] * gX[2][1]) * s;
      F[2][2] = (gX[0][0] * gX[1][1] - gX[0][1] * gX[1][0]) * s;
      // det F = 1 / det gradX exactly (F = gradX^-1); the reference evaluates the determinant of F itself, equal to rounding
      const double dgX = gX[0][0] * c00 + gX[0][1] * c01 + gX[0][2] * c02;
      const double Jr = dgX;                 // 1 / det F
      const double Je = s * rlam;            // det F / (lam0 lam1 lam2)
      double M[3][3], fa[3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        fa[i] = F[i][0] * A[0] + F[i][1] * A[1] + F[i][2] * A[2];
#pragma unroll
        for (int j = 0; j < 3; j++)
          M[i][j] = F[i][0] * lam[0] * gX[0][j] + F[i][1] * lam[1] * gX[1][j] + F[i][2] * lam[2] * gX[2][j];
      }
      // beta = Je dW/dJe and alpha = beta + Je^2 d2W/dJe^2 of hyperlastic_inline.h:42,47 in closed form (no divisions):
      //   dW/dJe = -mu/Je + lame/2 (Je - 1/Je),  d2W/dJe^2 = mu/Je^2 + lame/2 (1 + 1/Je^2)
      const double Je2 = Je * Je;
      const double beta = 0.5 * lame * (Je2 - 1.0) - mu;
      const double alpha = lame * Je2;
      double* pd = lds + (b * MAXE + pl) * PSTRIDE;
#pragma unroll
      for (int n = 0; n < 8; n++)
#pragma unroll
        for (int c = 0; c < 3; c++) pd[3 * n + c] = G[n][c];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) pd[24 + 3 * i + j] = M[i][j];
      const int V[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {1, 2}, {0, 2}};  // hyperelastic.h:15-20
#pragma unroll
      for (int p = 0; p < 6; p++) {
        const int i = V[p][0], j = V[p][1];
        const double qq = M[i][0] * M[j][0] + M[i][1] * M[j][1] + M[i][2] * M[j][2];
        const double bb = F[i][0] * F[j][0] + F[i][1] * F[j][1] + F[i][2] * F[j][2];
        pd[33 + p] = qq;
        pd[39 + p] = (mu * bb + beta * qq - Kf * fa[i] * fa[j]) * Jr * W;  // sigma * JxW
      }
      pd[45] = alpha * Jr * W;
      pd[46] = beta * Jr * W;
    };
    if (plive && !(diag & 2)) produce(0, 0);
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < 8; q++) {   // one point ahead of the consumers
      if (plive && q + 1 < 8 && !(diag & 2)) produce(q + 1, (q + 1) & 1);
      __syncthreads();
    }
    if (!(diag & 4)) zero_image();
    lds_barrier();
    lds_barrier();                  // consumers: atomics
    if (!(diag & 16)) copy_out();
    return;
  }
  // ================= consumer: one (owned node, element) pair per lane ========================================================
  double acc[8][3][3], re[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < 8; j++)
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) acc[j][r][c] = 0.0;
  int le = 0, li = 0, na = 0;
  const uint32_t pr = pair[(size_t)w * MAXP + tid];
  const bool cvalid = pr != 0xFFFFFFFFu;
  if (cvalid) { le = (int)(pr & 0xFF); li = (int)((pr >> 8) & 0xFF); na = (int)((pr >> 16) & 0xFF); }
  // one quadrature point: row li of the element matrix against all columns
  auto consume = [&](int b) {
    const double* pd = lds + (b * MAXE + le) * PSTRIDE;
    const double* pm = pd + 24;
    const double gi[3] = {pd[3 * li], pd[3 * li + 1], pd[3 * li + 2]};
    double M[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int c = 0; c < 3; c++) M[a][c] = pm[3 * a + c];
    const double Q[3][3] = {{pm[9], pm[12], pm[14]}, {pm[12], pm[10], pm[13]}, {pm[14], pm[13], pm[11]}};
    const double S[3][3] = {{pm[15], pm[18], pm[20]}, {pm[18], pm[16], pm[19]}, {pm[20], pm[19], pm[17]}};
    const double aW = pm[21], bW = pm[22];
    double ap[3], bn[3], sg[3], mi[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      ap[r] = aW * (Q[r][0] * gi[0] + Q[r][1] * gi[1] + Q[r][2] * gi[2]);
      bn[r] = bW * (M[0][r] * gi[0] + M[1][r] * gi[1] + M[2][r] * gi[2]);
      sg[r] = S[r][0] * gi[0] + S[r][1] * gi[1] + S[r][2] * gi[2];
      mi[r] = M[r][0] * gi[0] + M[r][1] * gi[1] + M[r][2] * gi[2];
      re[r] += sg[r];   // B_i sigma_voigt * JxW, hyperelastic.h:52-66
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const double gj[3] = {pd[3 * j], pd[3 * j + 1], pd[3 * j + 2]};
      if (SYM && j < li) {
        // use_symmetry: block (i, j) with j < i is the transpose of block (j, i), solid_system.C:252-262
        double apj[3], bnj[3], sgj[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
          apj[r] = aW * (Q[r][0] * gj[0] + Q[r][1] * gj[1] + Q[r][2] * gj[2]);
          bnj[r] = bW * (M[0][r] * gj[0] + M[1][r] * gj[1] + M[2][r] * gj[2]);
          sgj[r] = S[r][0] * gj[0] + S[r][1] * gj[1] + S[r][2] * gj[2];
        }
        const double kap = bnj[0] * gi[0] + bnj[1] * gi[1] + bnj[2] * gi[2];
        const double gam = sgj[0] * gi[0] + sgj[1] * gi[1] + sgj[2] * gi[2];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) {
            double v = fma(apj[c], gi[r], acc[j][r][c]);
            v = fma(-M[c][r], kap, v);
            v = fma(-mi[c], bnj[r], v);
            acc[j][r][c] = (r == c) ? v + gam : v;
          }
        continue;
      }
      const double kap = bn[0] * gj[0] + bn[1] * gj[1] + bn[2] * gj[2];
      const double gam = sg[0] * gj[0] + sg[1] * gj[1] + sg[2] * gj[2];
      double mj[3];
#pragma unroll
      for (int r = 0; r < 3; r++) mj[r] = M[r][0] * gj[0] + M[r][1] * gj[1] + M[r][2] * gj[2];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
          double v = fma(ap[r], gj[c], acc[j][r][c]);
          v = fma(-M[r][c], kap, v);
          v = fma(-mj[r], bn[c], v);
          acc[j][r][c] = (r == c) ? v + gam : v;
        }
    }
  };
  __syncthreads();                  // producers: point 0
#pragma unroll 1
  for (int q = 0; q < 8; q++) {
    if (cvalid && !(diag & 1)) consume(q & 1);
    __syncthreads();
  }
  // ---- epilogue: rows added into the LDS image, image copied out ------------------------------------------------------------
  if (!(diag & 4)) zero_image();
  uint32_t sl0 = 0, sl1 = 0;
  int off = 0, len3 = 0;
  if (cvalid) {
    sl0 = pslot[((size_t)w * MAXP + tid) * 2];
    sl1 = pslot[((size_t)w * MAXP + tid) * 2 + 1];
    const HostPrepCl::Node nd = ntab[(size_t)w * MAXN + na];
    off = (int)nd.off;
    len3 = 3 * (int)nd.len;
  }
  lds_barrier();
  if (cvalid && !(diag & 8)) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int s = (int)(((j < 4 ? sl0 : sl1) >> (8 * (j & 3))) & 0xFF);
      double* p = img + off + 3 * s;
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
          __hip_atomic_fetch_add(p + r * len3 + c, acc[j][r][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
      __hip_atomic_fetch_add(lrhs + 3 * na + c, re[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  lds_barrier();
  if (!(diag & 16)) copy_out();
}

size_t solid_cl_lds_bytes(int cw, int pw, size_t max_row_doubles) {
  const size_t points = (size_t)2 * pw * 64 * PSTRIDE;
  const size_t image = ((max_row_doubles + 1) & ~(size_t)1) + (size_t)3 * cw * 8;
  return sizeof(double) * (points > image ? points : image);
}

template <int CW, int PW>
static hipError_t launch_cl(const SolidArgs& a) {
  const size_t bytes = solid_cl_lds_bytes(CW, PW, a.cl.max_row_doubles);
  if (a.params.use_symmetry) {
    static std::atomic<uint64_t> attr[1];  /* per instantiation and device */
    dyn_lds_once(attr[0], (const void*)k_solid_cl<CW, PW, true>, 160 * 1024);
    hipLaunchKernelGGL((k_solid_cl<CW, PW, true>), dim3(a.cl.n_wg), dim3((CW + PW) * 64), bytes, a.stream, a.m, a.cl.desc, a.cl.ntab, a.cl.eid,
                       a.cl.pair, a.cl.pslot, a.Xu, a.fibre, a.elem_material, a.materials, a.params.pseudo_time, a.val, a.rhs, a.store_mode);
  } else {
    static std::atomic<uint64_t> attr[1];  /* per instantiation and device */
    dyn_lds_once(attr[0], (const void*)k_solid_cl<CW, PW, false>, 160 * 1024);
    hipLaunchKernelGGL((k_solid_cl<CW, PW, false>), dim3(a.cl.n_wg), dim3((CW + PW) * 64), bytes, a.stream, a.m, a.cl.desc, a.cl.ntab, a.cl.eid,
                       a.cl.pair, a.cl.pslot, a.Xu, a.fibre, a.elem_material, a.materials, a.params.pseudo_time, a.val, a.rhs, a.store_mode);
  }
  return hipGetLastError();
}

hipError_t launch_solid_cl(const SolidArgs& a) {
  if (a.cl.cw == 3 && a.cl.pw == 1) return launch_cl<3, 1>(a);
  if (a.cl.cw == 6 && a.cl.pw == 2) return launch_cl<6, 2>(a);
  return hipErrorInvalidValue;
}

}  // namespace rdc
