// rdc_solid.hip — SolidSystem residual / tangent kernels.
//
// Replaces SolidSystem::element_time_derivative (src/solid_system.C:146-271),
// side_time_derivative (:273-371) and the Hyperelastic law (src/hyperelastic.h:25-87,
// src/hyperlastic_inline.h:3-189).
//
// Constitutive algebra.  The reference forms the spatial tangent by an explicit 3^8 push-forward
// of dS/dC (hyperlastic_inline.h:100-149).  With dWdI2 = d2W/dI1^2 = d2W/dI2^2 = d2W/dI4^2 = 0
// (:41-47) and Fp = diag(lambda) (hyperelastic.h:39-44) that sum closes exactly:
//     dS/dCe_IJKL = alpha Ci_IJ Ci_KL - beta (Ci_IK Ci_JL + Ci_IL Ci_JK),     Ci = Ce^-1
//     beta  = Je dW/dJe,   alpha = beta + Je^2 d2W/dJe^2
//     dCe/dC contracts to a 1/(lambda_K lambda_L) scaling, i.e. it turns F into Fe = F Fp^-1 on (K,L)
//     Fe Ci Fe^T = I,   F Ci Fe^T = F Fp F^-1 =: M,   F Ci F^T = M M^T =: Q
//   =>  c_ijkl  = (1/det F) [ alpha Q_ij delta_kl - beta (M_ik M_jl + M_il M_jk) ]
//       sigma   = (1/det F) [ mu F F^T + beta Q - K (F A)(F A)^T ]
// Both keep the reference's quirks: total F and 1/det F in the push-forward although S is built
// from Ce (App. D.6), and the resulting Voigt matrix is NOT major-symmetric when Fp != I.
#include "rdc_solid.h"

namespace rdc {

struct SolidPoint {
  double sigma[3][3];
  double C[6][6];
};

// F^-1 = gradX (d X_d / d x_c), lam = growth stretches, A = unit fibre (or 0)
__device__ __forceinline__ void solid_point(const double (&gX)[3][3], const double (&lam)[3], const double (&A)[3],
                                            double mu, double lame, double K, bool tangent, SolidPoint& o) {
  // F = gradX^-1
  const double c00 = gX[1][1] * gX[2][2] - gX[1][2] * gX[2][1];
  const double c01 = gX[1][2] * gX[2][0] - gX[1][0] * gX[2][2];
  const double c02 = gX[1][0] * gX[2][1] - gX[1][1] * gX[2][0];
  const double dgX = gX[0][0] * c00 + gX[0][1] * c01 + gX[0][2] * c02;
  const double s = 1.0 / dgX;
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
  const double detF = F[0][0] * (F[1][1] * F[2][2] - F[1][2] * F[2][1]) - F[0][1] * (F[1][0] * F[2][2] - F[1][2] * F[2][0]) +
                      F[0][2] * (F[1][0] * F[2][1] - F[1][1] * F[2][0]);
  const double Jr = 1.0 / detF;
  const double Je = detF / (lam[0] * lam[1] * lam[2]);
  // M = F diag(lam) gradX, Q = M M^T, b = F F^T, a = F A
  double M[3][3], Q[3][3], b[3][3], a[3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      M[i][j] = F[i][0] * lam[0] * gX[0][j] + F[i][1] * lam[1] * gX[1][j] + F[i][2] * lam[2] * gX[2][j];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    a[i] = F[i][0] * A[0] + F[i][1] * A[1] + F[i][2] * A[2];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      Q[i][j] = M[i][0] * M[j][0] + M[i][1] * M[j][1] + M[i][2] * M[j][2];
      b[i][j] = F[i][0] * F[j][0] + F[i][1] * F[j][1] + F[i][2] * F[j][2];
    }
  }
  const double dWdJe = (-mu / Je) + (lame / 2.0 * Je - lame / 2.0 / Je);          // hyperlastic_inline.h:42
  const double d2W = (mu / Je / Je) + (lame / 2.0 + lame / 2.0 / Je / Je);        // :47
  const double beta = Je * dWdJe;
  const double alpha = beta + Je * Je * d2W;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) o.sigma[i][j] = (mu * b[i][j] + beta * Q[i][j] - K * a[i] * a[j]) * Jr;
  if (!tangent) return;
  const int V[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {1, 2}, {0, 2}};  // hyperelastic.h:15-20
#pragma unroll
  for (int p = 0; p < 6; p++)
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const int i = V[p][0], j = V[p][1], k = V[q][0], l = V[q][1];
      o.C[p][q] = (alpha * Q[i][j] * (k == l ? 1.0 : 0.0) - beta * (M[i][k] * M[j][l] + M[i][l] * M[j][k])) * Jr;
    }
}

// B_i (3x6), src/hyperlastic_inline.h:3-15: rows r, Voigt columns (00,11,22,01,12,02)
__device__ __forceinline__ void bmat(const double (&g)[3], double (&B)[3][6]) {
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int p = 0; p < 6; p++) B[r][p] = 0.0;
  B[0][0] = g[0]; B[1][1] = g[1]; B[2][2] = g[2];
  B[0][3] = g[1]; B[1][3] = g[0];
  B[1][4] = g[2]; B[2][4] = g[1];
  B[0][5] = g[2]; B[2][5] = g[0];
}

template <int NEN>
__global__ void __launch_bounds__(128)
k_solid_coloured(const MeshDev m, int64_t first, int64_t count, const double* __restrict__ Xu,
                 const double* __restrict__ fibre, const int32_t* __restrict__ elem_material,
                 const rdc_solid_material* __restrict__ materials, double pseudo_time, int use_symmetry,
                 int request_jacobian, double* __restrict__ val, double* __restrict__ rhs) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int64_t e = m.elem_order[first + t];
  double X[NEN][3], XU[NEN][3];
#pragma unroll
  for (int i = 0; i < NEN; i++) {
    const int64_t n = m.conn[e * NEN + i];
#pragma unroll
    for (int d = 0; d < 3; d++) { X[i][d] = m.xyz[3 * n + d]; XU[i][d] = Xu[3 * n + d]; }
  }
  const rdc_solid_material mat = materials[elem_material[e]];   // src/solid_system.C:183-190
  const double mu = 0.5 * mat.Young / (1.0 + mat.Poisson);      // hyperlastic_inline.h:21-24
  const double lame = mat.Young * mat.Poisson / ((1.0 + mat.Poisson) * (1.0 - 2.0 * mat.Poisson));
  const double K = mat.FibreStiffness;                          // 2*dWdI4 = -K
  double A[3] = {0.0, 0.0, 0.0};
  if (K > 0.0) {                                                // hyperelastic.h:46
    const double f0 = fibre[3 * e], f1 = fibre[3 * e + 1], f2 = fibre[3 * e + 2];
    const double nrm = sqrt(f0 * f0 + f1 * f1 + f2 * f2);
    A[0] = f0 / nrm; A[1] = f1 / nrm; A[2] = f2 / nrm;
  }
  const double Kf = (K > 0.0) ? K : 0.0;
  double lam[3];
#pragma unroll
  for (int d = 0; d < 3; d++) lam[d] = 1.0 + pseudo_time * mat.rate[d];  // solid_system.C:232-234
  const uint64_t fm = m.first_mask[e];
  const uint32_t fr = m.first_rhs[e];
#pragma unroll 1
  for (int i = 0; i < NEN; i++) {
    const int64_t I = m.conn[e * NEN + i];
    if (I >= m.n_owned) continue;
    double acc[3][3][NEN], re[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++)
#pragma unroll
        for (int j = 0; j < NEN; j++) acc[a][b][j] = 0.0;
#pragma unroll 1
    for (int q = 0; q < Ref<NEN>::NQP; q++) {
      double N[NEN], G[NEN][3], W;
      fe_point<NEN>(X, q, N, G, W);
      double gX[3][3];  // gradX[d][c] = sum_l dphi_l[c] * X_l[d], solid_system.C:221-229
#pragma unroll
      for (int d = 0; d < 3; d++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
          double s = 0.0;
#pragma unroll
          for (int l = 0; l < NEN; l++) s += G[l][c] * XU[l][d];
          gX[d][c] = s;
        }
      SolidPoint P;
      solid_point(gX, lam, A, mu, lame, Kf, request_jacobian != 0, P);
      double gi[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int n = 0; n < NEN; n++)
        if (n == i) { gi[0] = G[n][0]; gi[1] = G[n][1]; gi[2] = G[n][2]; }
      // residual: B_i sigma_voigt, hyperelastic.h:52-66 (uses sigma(0,1), sigma(1,2), sigma(0,2))
      re[0] += W * (gi[0] * P.sigma[0][0] + gi[1] * P.sigma[0][1] + gi[2] * P.sigma[0][2]);
      re[1] += W * (gi[1] * P.sigma[1][1] + gi[0] * P.sigma[0][1] + gi[2] * P.sigma[1][2]);
      re[2] += W * (gi[2] * P.sigma[2][2] + gi[1] * P.sigma[1][2] + gi[0] * P.sigma[0][2]);
      if (!request_jacobian) continue;
      double Bi[3][6], BC[3][6], BCt[3][6];
      bmat(gi, Bi);
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int p = 0; p < 6; p++) {
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int x = 0; x < 6; x++) { s0 += Bi[r][x] * P.C[x][p]; s1 += Bi[r][x] * P.C[p][x]; }
          BC[r][p] = s0;   // B_i C
          BCt[r][p] = s1;  // B_i C^T  (mirror of the (j,i) block under use_symmetry, solid_system.C:261-262)
        }
      double sg[3];
#pragma unroll
      for (int r = 0; r < 3; r++) sg[r] = gi[0] * P.sigma[0][r] + gi[1] * P.sigma[1][r] + gi[2] * P.sigma[2][r];
#pragma unroll
      for (int j = 0; j < NEN; j++) {
        const double gj[3] = {G[j][0], G[j][1], G[j][2]};
        double Bj[3][6];
        bmat(gj, Bj);
        // geometric term dphi_i . sigma . dphi_j (for the mirrored block: dphi_j . sigma . dphi_i)
        double Gnn;
        const bool mirrored = use_symmetry && (j < i);
        if (!mirrored) Gnn = sg[0] * gj[0] + sg[1] * gj[1] + sg[2] * gj[2];
        else {
          Gnn = 0.0;
#pragma unroll
          for (int r = 0; r < 3; r++) Gnn += gj[r] * (P.sigma[r][0] * gi[0] + P.sigma[r][1] * gi[1] + P.sigma[r][2] * gi[2]);
        }
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) {
            double s0 = 0.0;
#pragma unroll
            for (int p = 0; p < 6; p++) s0 += (mirrored ? BCt[r][p] : BC[r][p]) * Bj[c][p];
            acc[r][c][j] += W * (s0 + (r == c ? Gnn : 0.0));
          }
      }
    }
    const bool first_r = (fr >> i) & 1u;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      double* p = rhs + I * 3 + a;
      *p = first_r ? re[a] : (*p + re[a]);
    }
    // the matrix is (re)written also when no Jacobian is requested so that stale values never
    // survive; it then holds zeros
    const int64_t b0 = m.bptr[I];
    const int64_t len = m.bptr[I + 1] - b0;
    double* row = val + 9 * b0;
#pragma unroll
    for (int j = 0; j < NEN; j++) {
      const int64_t s = m.eslot[e * (NEN * NEN) + i * NEN + j];
      const bool first_writer = (fm >> (i * NEN + j)) & 1ull;
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
          double* p = row + a * 3 * len + 3 * s + b;
          *p = first_writer ? acc[a][b][j] : (*p + acc[a][b][j]);
        }
    }
  }
}

// penalty Dirichlet sides, src/solid_system.C:273-371.  Few entries (boundary only); neighbouring
// sides share nodes, so these adds use FP64 hardware atomics.
template <int NEN>
__global__ void k_solid_sides(const MeshDev m, int64_t n_sides, const int64_t* __restrict__ side_elem,
                              const int32_t* __restrict__ side_id, const double* __restrict__ side_disp,
                              const double* __restrict__ Xu, double pseudo_time, double penalty,
                              int request_jacobian, double* __restrict__ val, double* __restrict__ rhs) {
  const int64_t sidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (sidx >= n_sides) return;
  constexpr int NSN = (NEN == 4) ? 3 : 4;
  const int TS[4][3] = {{0, 2, 1}, {0, 1, 3}, {1, 2, 3}, {2, 0, 3}};
  const int HS[6][4] = {{0, 3, 2, 1}, {0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {3, 0, 4, 7}, {4, 5, 6, 7}};
  const int64_t e = side_elem[sidx];
  const int sd = side_id[sidx];
  int loc[NSN];
  int64_t nd[NSN];
  double x[NSN][3], X0[NSN][3];
  for (int a = 0; a < NSN; a++) {
    loc[a] = (NEN == 4) ? TS[sd][a % 3] : HS[sd][a];
    nd[a] = m.conn[e * NEN + loc[a]];
    for (int d = 0; d < 3; d++) { x[a][d] = m.xyz[3 * nd[a] + d]; X0[a][d] = Xu[3 * nd[a] + d]; }
  }
  const double ratio = pseudo_time * 1.000001;  // :291-292
  double disp[3];
  for (int d = 0; d < 3; d++) disp[d] = side_disp[3 * sidx + d] * ratio;
  for (int q = 0; q < 4; q++) {
    double N[NSN], dN[NSN][2], w;
    if (NEN == 4) {  // QGauss(2, THIRD) on TRI3: centroid -27/96, three (0.2,0.2,0.6)-type 25/96
      const double px = (q == 0) ? (1.0 / 3.0) : (q == 3 ? 0.6 : 0.2);
      const double py = (q == 0) ? (1.0 / 3.0) : (q == 1 ? 0.6 : 0.2);
      w = (q == 0) ? (-27.0 / 96.0) : (25.0 / 96.0);
      N[0] = 1.0 - px - py; N[1] = px; N[2] = py;
      dN[0][0] = -1.0; dN[0][1] = -1.0; dN[1][0] = 1.0; dN[1][1] = 0.0; dN[2][0] = 0.0; dN[2][1] = 1.0;
    } else {         // 2x2 Gauss on QUAD4
      const double g = 0.57735026918962576451;
      const double px = (q & 1) ? g : -g, py = (q & 2) ? g : -g;
      w = 1.0;
      const double cx[4] = {-1, 1, 1, -1}, cy[4] = {-1, -1, 1, 1};
      for (int a = 0; a < NSN; a++) {
        N[a] = 0.25 * (1 + cx[a % 4] * px) * (1 + cy[a % 4] * py);
        dN[a][0] = 0.25 * cx[a % 4] * (1 + cy[a % 4] * py);
        dN[a][1] = 0.25 * (1 + cx[a % 4] * px) * cy[a % 4];
      }
    }
    double t1[3] = {0, 0, 0}, t2[3] = {0, 0, 0}, cur[3] = {0, 0, 0}, org[3] = {0, 0, 0};
    for (int a = 0; a < NSN; a++)
      for (int d = 0; d < 3; d++) {
        t1[d] += x[a][d] * dN[a][0];
        t2[d] += x[a][d] * dN[a][1];
        cur[d] += x[a][d] * N[a];
        org[d] += X0[a][d] * N[a];
      }
    const double cx_ = t1[1] * t2[2] - t1[2] * t2[1], cy_ = t1[2] * t2[0] - t1[0] * t2[2], cz_ = t1[0] * t2[1] - t1[1] * t2[0];
    const double W = sqrt(cx_ * cx_ + cy_ * cy_ + cz_ * cz_) * w;
    double diff[3];
    for (int d = 0; d < 3; d++) diff[d] = cur[d] - org[d] - disp[d];  // :337-339
    for (int a = 0; a < NSN; a++) {
      const int64_t I = nd[a];
      if (I >= m.n_owned) continue;
      for (int di = 0; di < 3; di++) {
        if (diff[di] != diff[di]) continue;  // NaN component = unconstrained, :346
        unsafeAtomicAdd(rhs + I * 3 + di, W * N[a] * diff[di] * penalty);
      }
      if (!request_jacobian) continue;
      const int64_t b0 = m.bptr[I];
      const int64_t len = m.bptr[I + 1] - b0;
      for (int b = 0; b < NSN; b++) {
        const int64_t s = m.eslot[e * (NEN * NEN) + loc[a] * NEN + loc[b]];
        for (int dj = 0; dj < 3; dj++) {
          if (diff[dj] != diff[dj]) continue;  // :358
          unsafeAtomicAdd(val + 9 * b0 + dj * 3 * len + 3 * s + dj, W * N[a] * N[b] * penalty);
        }
      }
    }
  }
}

// ================================================================================================
// Two-pass assembly (default).  The solid tangent is FP64-compute-bound (~26 k FMA per HEX8 element
// after the algebra below, against ~5 kB of compulsory traffic), so the scatter is taken off the
// compute kernel altogether:
//   pass 1  k_solid_elem   every element once: its NEN x NEN blocks of 3x3 -> ke[e][i][j][3][3], fe[e][i][3]
//   pass 2  k_solid_gather every CSR node block once: sum of the element blocks listed for it
//           (SolidGather, fixed order => deterministic), written to its three row pieces.
// No atomics, no colours, no recomputation; the price is one write + one read of the element
// matrices (4.6 kB per HEX8) through HBM, which costs less than the constitutive work it saves.
//
// Element-matrix algebra.  With c_ijkl and sigma from the header comment, and the B-matrices of
// hyperlastic_inline.h:3-15, the block of nodes (i, j) at one quadrature point
//   D_rc = (g_i . sigma . g_j) delta_rc + (B_i C B_j^T)_rc ,  g = grad phi       (hyperelastic.h:68-87)
// is, because c is symmetric in (i,j) and in (k,l) so the Voigt lookup is exact,
//   (B_i C B_j^T)_rc = sum_st g_i[s] g_j[t] c_(rs)(ct)
//                    = [ alpha (Q g_i)_r g_j[c] - beta ( M_rc (n_i . g_j) + (M g_j)_r n_i[c] ) ] / det F ,
//   n_i = M^T g_i .
// Work split inside a workgroup of 128 threads = 128 / NEN elements:
//   phase 1  thread (element, q): shape gradients, F, M, Q, sigma at quadrature point q -> LDS
//            (TET4: gradients are constant, one point with the summed weight)
//   phase 2  thread (element, i): row node i, all columns j, loop over the points read from LDS
//   phase 3  rows staged through LDS so that the element matrices leave as contiguous 16-byte stores.
// ================================================================================================
template <int NEN> struct SolidCfg;
template <> struct SolidCfg<8> { static constexpr int NQ = 8, EPB = 16, PSTRIDE = 49, ESTRIDE = 8 * 49 + 1, ROW = 74, HALVES = 2; };
template <> struct SolidCfg<4> { static constexpr int NQ = 1, EPB = 32, PSTRIDE = 35, ESTRIDE = 35, ROW = 38, HALVES = 1; };
template <int NEN> struct SolidLds {
  using C = SolidCfg<NEN>;
  static constexpr int CSTRIDE = NEN * 6 + 1;
  static constexpr int POINTS = C::EPB * C::ESTRIDE;
  static constexpr int STAGE = (C::EPB / C::HALVES) * NEN * C::ROW;
  static constexpr int COORDS = C::EPB * CSTRIDE;
  static constexpr int DOUBLES = (POINTS > STAGE ? (POINTS > COORDS ? POINTS : COORDS) : (STAGE > COORDS ? STAGE : COORDS));
};

typedef double rdc_v2d __attribute__((ext_vector_type(2)));

// JS = 2 splits the columns of a row between two threads (HEX8: 256 threads, 36 accumulators each): twice
// the waves per LDS byte to hide the LDS / global latencies of the phases.
template <int NEN, bool JAC, bool SYM, int JS>
__global__ void __launch_bounds__(128 * JS)
k_solid_elem(const MeshDev m, const double* __restrict__ Xu, const double* __restrict__ fibre,
             const int32_t* __restrict__ elem_material, const rdc_solid_material* __restrict__ materials,
             double pseudo_time, double* __restrict__ ke, double* __restrict__ fe, const int store_mode) {
  using C = SolidCfg<NEN>;
  using L = SolidLds<NEN>;
  constexpr int NB = NEN * 9;  // doubles of one row (node i, all j) of the element matrix
  __shared__ __attribute__((aligned(16))) double lds[L::DOUBLES];
  constexpr int NT = 128 * JS, NJ = NEN / JS;
  const int tid = threadIdx.x;
  const int jh = tid / 128;                       // column half; waves of half 1 skip phases 0 and 1
  const int el = (tid % 128) / NEN, li = tid % NEN;
  const int64_t e0 = (int64_t)blockIdx.x * C::EPB;
  const int64_t e = e0 + el;
  const bool live = e < m.n_elem;
  // ---- phase 0: node coordinates of the workgroup's elements ---------------------------------
  int64_t I = 0;
  if (live) I = m.conn[e * NEN + li];
  if (live && jh == 0) {
    double* c = lds + el * L::CSTRIDE + li * 6;
#pragma unroll
    for (int d = 0; d < 3; d++) { c[d] = m.xyz[3 * I + d]; c[3 + d] = Xu[3 * I + d]; }
  }
  __syncthreads();
  double X[NEN][3], XU[NEN][3];
  if (live && jh == 0 && li < C::NQ) {
#pragma unroll
    for (int n = 0; n < NEN; n++)
#pragma unroll
      for (int d = 0; d < 3; d++) {
        X[n][d] = lds[el * L::CSTRIDE + n * 6 + d];
        XU[n][d] = lds[el * L::CSTRIDE + n * 6 + 3 + d];
      }
  }
  __syncthreads();
  // ---- phase 1: one quadrature point per thread -----------------------------------------------
  if (live && jh == 0 && li < C::NQ) {
    double N[NEN], G[NEN][3], W;
    if (NEN == 8) fe_point<NEN>(X, li, N, G, W);
    else {  // constant gradients: one evaluation carries the summed weight of the five points
      double w = 0.0;
#pragma unroll
      for (int q = 0; q < Ref<NEN>::NQP; q++) { fe_point<NEN>(X, q, N, G, W); w += W; }
      W = w;
    }
    double gX[3][3];  // gradX[d][c] = sum_l dphi_l[c] * X_l[d], solid_system.C:221-229
#pragma unroll
    for (int d = 0; d < 3; d++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < NEN; l++) s += G[l][c] * XU[l][d];
        gX[d][c] = s;
      }
    const rdc_solid_material mat = materials[elem_material[e]];   // src/solid_system.C:183-190
    const double mu = 0.5 * mat.Young / (1.0 + mat.Poisson);      // hyperlastic_inline.h:21-24
    const double lame = mat.Young * mat.Poisson / ((1.0 + mat.Poisson) * (1.0 - 2.0 * mat.Poisson));
    const double K = mat.FibreStiffness;
    double A[3] = {0.0, 0.0, 0.0};
    if (K > 0.0) {                                                // hyperelastic.h:46
      const double f0 = fibre[3 * e], f1 = fibre[3 * e + 1], f2 = fibre[3 * e + 2];
      const double nrm = sqrt(f0 * f0 + f1 * f1 + f2 * f2);
      A[0] = f0 / nrm; A[1] = f1 / nrm; A[2] = f2 / nrm;
    }
    const double Kf = (K > 0.0) ? K : 0.0;
    double lam[3];
#pragma unroll
    for (int d = 0; d < 3; d++) lam[d] = 1.0 + pseudo_time * mat.rate[d];  // solid_system.C:232-234
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
    const double detF = F[0][0] * (F[1][1] * F[2][2] - F[1][2] * F[2][1]) - F[0][1] * (F[1][0] * F[2][2] - F[1][2] * F[2][0]) +
                        F[0][2] * (F[1][0] * F[2][1] - F[1][1] * F[2][0]);
    const double Jr = 1.0 / detF;
    const double Je = detF / (lam[0] * lam[1] * lam[2]);
    double M[3][3], fa[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      fa[i] = F[i][0] * A[0] + F[i][1] * A[1] + F[i][2] * A[2];
#pragma unroll
      for (int j = 0; j < 3; j++)
        M[i][j] = F[i][0] * lam[0] * gX[0][j] + F[i][1] * lam[1] * gX[1][j] + F[i][2] * lam[2] * gX[2][j];
    }
    const double dWdJe = (-mu / Je) + (lame / 2.0 * Je - lame / 2.0 / Je);          // hyperlastic_inline.h:42
    const double d2W = (mu / Je / Je) + (lame / 2.0 + lame / 2.0 / Je / Je);        // :47
    const double beta = Je * dWdJe;
    const double alpha = beta + Je * Je * d2W;
    double* pd = lds + el * C::ESTRIDE + (NEN == 8 ? li : 0) * C::PSTRIDE;
#pragma unroll
    for (int n = 0; n < NEN; n++)
#pragma unroll
      for (int d = 0; d < 3; d++) pd[3 * n + d] = G[n][d];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) pd[3 * NEN + 3 * i + j] = M[i][j];
    const int V[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {1, 2}, {0, 2}};  // hyperelastic.h:15-20
#pragma unroll
    for (int p = 0; p < 6; p++) {
      const int i = V[p][0], j = V[p][1];
      const double q = M[i][0] * M[j][0] + M[i][1] * M[j][1] + M[i][2] * M[j][2];
      const double b = F[i][0] * F[j][0] + F[i][1] * F[j][1] + F[i][2] * F[j][2];
      pd[3 * NEN + 9 + p] = q;
      pd[3 * NEN + 15 + p] = (mu * b + beta * q - Kf * fa[i] * fa[j]) * Jr * W;  // sigma * JxW
    }
    pd[3 * NEN + 21] = alpha * Jr * W;
    pd[3 * NEN + 22] = beta * Jr * W;
  }
  __syncthreads();
  // ---- phase 2: row node li of element el -------------------------------------------------------
  double acc[NJ][3][3], re[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) acc[j][r][c] = 0.0;
  if (live && I < m.n_owned) {
#pragma unroll 1
    for (int q = 0; q < C::NQ; q++) {
      const double* pd = lds + el * C::ESTRIDE + q * C::PSTRIDE;
      const double* pm = pd + 3 * NEN;
      const double gi[3] = {pd[3 * li], pd[3 * li + 1], pd[3 * li + 2]};
      double M[3][3];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) M[a][b] = pm[3 * a + b];
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
      if (!JAC) continue;
#pragma unroll
      for (int jj = 0; jj < NJ; jj++) {
        const int j = jh * NJ + jj;
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
              double v = fma(apj[c], gi[r], acc[jj][r][c]);
              v = fma(-M[c][r], kap, v);
              v = fma(-mi[c], bnj[r], v);
              acc[jj][r][c] = (r == c) ? v + gam : v;
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
            double v = fma(ap[r], gj[c], acc[jj][r][c]);
            v = fma(-M[r][c], kap, v);
            v = fma(-mj[r], bn[c], v);
            acc[jj][r][c] = (r == c) ? v + gam : v;
          }
      }
    }
  }
  if (live && jh == 0) {
    double* f = fe + (e * NEN + li) * 3;
    f[0] = re[0]; f[1] = re[1]; f[2] = re[2];
  }
  if (!JAC) return;
  // diagnostic store modes (timing only): 1 = each thread stores its own row straight from registers (576-byte
  // stride between lanes), 2 = no element-matrix stores at all
  if (store_mode == 2) return;
  if (store_mode == 1) {
    if (live) {
      rdc_v2d* dst = reinterpret_cast<rdc_v2d*>(ke + (e * NEN + li) * NB + jh * NJ * 9);
      const double* a0 = &acc[0][0][0];
#pragma unroll
      for (int x = 0; x < NJ * 9 / 2; x++) { rdc_v2d v; v.x = a0[2 * x]; v.y = a0[2 * x + 1]; __builtin_nontemporal_store(v, dst + x); }
    }
    return;
  }
  // ---- phase 3: rows -> LDS -> contiguous 16-byte stores -------------------------------------------
  constexpr int EPH = C::EPB / C::HALVES;  // elements per staging pass
#pragma unroll 1
  for (int h = 0; h < C::HALVES; h++) {
    __syncthreads();
    if (el / EPH == h) {
      double* row = lds + ((el % EPH) * NEN + li) * C::ROW + jh * NJ * 9;
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) row[j * 9 + 3 * r + c] = acc[j][r][c];
    }
    __syncthreads();
    const int64_t eb = e0 + (int64_t)h * EPH;
    int64_t nrows = (m.n_elem - eb) * NEN;
    if (nrows > EPH * NEN) nrows = EPH * NEN;
    const int n2 = (int)(nrows > 0 ? nrows : 0) * (NB / 2);  // 16-byte units
    double* dst = ke + eb * NEN * NB;
    for (int x = tid; x < n2; x += NT) {
      const int rw = x / (NB / 2), k2 = x - rw * (NB / 2);
      const rdc_v2d v = *reinterpret_cast<const rdc_v2d*>(lds + rw * C::ROW + 2 * k2);
      __builtin_nontemporal_store(v, reinterpret_cast<rdc_v2d*>(dst) + x);
    }
  }
}

// pass 2: one thread per node block of the owned rows
template <bool JAC>
__global__ void __launch_bounds__(256)
k_solid_gather(int64_t nblocks, const int64_t* __restrict__ bptr, const int32_t* __restrict__ brow,
               const uint32_t* __restrict__ gptr, const uint32_t* __restrict__ gsrc, const double* __restrict__ ke,
               double* __restrict__ val) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  double a[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (JAC) {
    const uint32_t s1 = gptr[b + 1];
    for (uint32_t s = gptr[b]; s < s1; s++) {
      const double* k = ke + (int64_t)gsrc[s] * 9;
#pragma unroll
      for (int x = 0; x < 9; x++) a[x] += k[x];
    }
  }
  // when no Jacobian is requested the matrix is still rewritten (zeros) so stale values never survive
  const int64_t I = brow[b];
  const int64_t b0 = bptr[I], len = bptr[I + 1] - b0;
  double* row = val + 9 * b0 + 3 * (b - b0);
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) __builtin_nontemporal_store(a[3 * r + c], row + r * 3 * len + c);
}

// rhs: one thread per owned node sums the rows of its incident (element, local node) pairs
__global__ void __launch_bounds__(256)
k_solid_gather_rhs(int64_t n_owned, int nen, const int64_t* __restrict__ node_pair_ptr,
                   const uint32_t* __restrict__ pair_elem, const uint8_t* __restrict__ pair_local,
                   const double* __restrict__ fe, double* __restrict__ rhs) {
  const int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (I >= n_owned) return;
  double r0 = 0.0, r1 = 0.0, r2 = 0.0;
  for (int64_t p = node_pair_ptr[I]; p < node_pair_ptr[I + 1]; p++) {
    const double* f = fe + ((int64_t)pair_elem[p] * nen + pair_local[p]) * 3;
    r0 += f[0]; r1 += f[1]; r2 += f[2];
  }
  rhs[3 * I] = r0; rhs[3 * I + 1] = r1; rhs[3 * I + 2] = r2;
}

// pass 2, staged form (default): 256 consecutive node blocks per workgroup, one thread each as above, but
// the nine sums go through LDS so that the three row pieces of the blocks leave as runs of consecutive
// doubles (a node's blocks are consecutive in each of its three CSR rows) instead of 24-byte pieces.
__global__ void __launch_bounds__(256)
k_solid_gather_st(int64_t nblocks, const int64_t* __restrict__ bptr, const int32_t* __restrict__ brow,
                  const uint32_t* __restrict__ gptr, const uint32_t* __restrict__ gsrc, const double* __restrict__ ke,
                  double* __restrict__ val) {
  __shared__ double img[3][3 * 256];
  __shared__ int64_t obase[256];   // value index of (row 0, column 0) of the block
  __shared__ int32_t ostride[256]; // 3 * blocks in the row of its node
  const int64_t B0 = (int64_t)blockIdx.x * 256;
  const int64_t b = B0 + threadIdx.x;
  if (b < nblocks) {
    double a[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const uint32_t s1 = gptr[b + 1];
    for (uint32_t s = gptr[b]; s < s1; s++) {
      const double* k = ke + (int64_t)gsrc[s] * 9;
#pragma unroll
      for (int x = 0; x < 9; x++) a[x] += k[x];
    }
    const int64_t I = brow[b];
    const int64_t nb0 = bptr[I], len = bptr[I + 1] - nb0;
    obase[threadIdx.x] = 9 * nb0 + 3 * (b - nb0);
    ostride[threadIdx.x] = (int32_t)(3 * len);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) img[r][3 * threadIdx.x + c] = a[3 * r + c];
  }
  __syncthreads();
  const int64_t rem = nblocks - B0;
  const int nx = 3 * (int)(rem < 256 ? rem : 256);
#pragma unroll
  for (int r = 0; r < 3; r++)
    for (int x = threadIdx.x; x < nx; x += 256) {
      const int tb = x / 3, c = x - 3 * tb;
      __builtin_nontemporal_store(img[r][x], val + obase[tb] + (int64_t)r * ostride[tb] + c);
    }
}

template <int NEN>
static void launch_two_pass(const SolidArgs& a) {
  const unsigned grid = (unsigned)((a.m.n_elem + SolidCfg<NEN>::EPB - 1) / SolidCfg<NEN>::EPB);
#define RDC_SOLID_ELEM(JAC, SYM, JS)                                                                                \
  hipLaunchKernelGGL((k_solid_elem<NEN, JAC, SYM, JS>), dim3(grid), dim3(128 * JS), 0, a.stream, a.m, a.Xu, a.fibre, \
                     a.elem_material, a.materials, a.params.pseudo_time, a.ke, a.fe, a.store_mode)
  constexpr int JSD = (NEN == 8) ? 2 : 1;
  if (!a.request_jacobian) RDC_SOLID_ELEM(false, false, 1);
  else if (a.params.use_symmetry) RDC_SOLID_ELEM(true, true, JSD);
  else if (a.split == 1) RDC_SOLID_ELEM(true, false, 1);
  else RDC_SOLID_ELEM(true, false, JSD);
#undef RDC_SOLID_ELEM
  const unsigned gb = (unsigned)((a.nblocks + 255) / 256);
  if (a.request_jacobian && a.gather == 0)
    hipLaunchKernelGGL(k_solid_gather_st, dim3(gb), dim3(256), 0, a.stream, a.nblocks, a.m.bptr, a.brow, a.gptr, a.gsrc, a.ke,
                       a.val);
  else if (a.request_jacobian)
    hipLaunchKernelGGL((k_solid_gather<true>), dim3(gb), dim3(256), 0, a.stream, a.nblocks, a.m.bptr, a.brow, a.gptr, a.gsrc,
                       a.ke, a.val);
  else  // the matrix is still rewritten (zeros) so that stale values never survive
    (void)hipMemsetAsync(a.val, 0, (size_t)a.nblocks * 9 * sizeof(double), a.stream);
  const unsigned gn = (unsigned)((a.m.n_owned + 255) / 256);
  hipLaunchKernelGGL(k_solid_gather_rhs, dim3(gn), dim3(256), 0, a.stream, a.m.n_owned, NEN, a.m.node_pair_ptr,
                     a.m.pair_elem, a.m.pair_local, a.fe, a.rhs);
}

// SolidSystem::post_process (src/solid_system.C:394-538): one thread per element.
// out: [n_elem][5] = {pressure, von Mises, F*eta averaged (3)}
template <int NEN>
__global__ void __launch_bounds__(128)
k_solid_post(const MeshDev m, const double* __restrict__ Xu, const double* __restrict__ fibre,
             const int32_t* __restrict__ elem_material, const rdc_solid_material* __restrict__ materials,
             double pseudo_time, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m.n_elem) return;
  double X[NEN][3], XU[NEN][3];
#pragma unroll
  for (int i = 0; i < NEN; i++) {
    const int64_t n = m.conn[e * NEN + i];
#pragma unroll
    for (int d = 0; d < 3; d++) { X[i][d] = m.xyz[3 * n + d]; XU[i][d] = Xu[3 * n + d]; }
  }
  const rdc_solid_material mat = materials[elem_material[e]];
  const double mu = 0.5 * mat.Young / (1.0 + mat.Poisson);
  const double lame = mat.Young * mat.Poisson / ((1.0 + mat.Poisson) * (1.0 - 2.0 * mat.Poisson));
  const double K = mat.FibreStiffness;
  const double eta[3] = {fibre[3 * e], fibre[3 * e + 1], fibre[3 * e + 2]};
  double A[3] = {0.0, 0.0, 0.0};
  if (K > 0.0) {
    const double nrm = sqrt(eta[0] * eta[0] + eta[1] * eta[1] + eta[2] * eta[2]);
    A[0] = eta[0] / nrm; A[1] = eta[1] / nrm; A[2] = eta[2] / nrm;
  }
  const double Kf = (K > 0.0) ? K : 0.0;
  double lam[3];
#pragma unroll
  for (int d = 0; d < 3; d++) lam[d] = 1.0 + pseudo_time * mat.rate[d];
  double sc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, fv[3] = {0.0, 0.0, 0.0};  // 00,11,22,01,12,02
#pragma unroll 1
  for (int q = 0; q < Ref<NEN>::NQP; q++) {
    double N[NEN], G[NEN][3], W;
    fe_point<NEN>(X, q, N, G, W);
    double gX[3][3];
#pragma unroll
    for (int d = 0; d < 3; d++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < NEN; l++) s += G[l][c] * XU[l][d];
        gX[d][c] = s;
      }
    SolidPoint P;
    solid_point(gX, lam, A, mu, lame, Kf, false, P);
    sc[0] += P.sigma[0][0]; sc[1] += P.sigma[1][1]; sc[2] += P.sigma[2][2];
    sc[3] += P.sigma[0][1]; sc[4] += P.sigma[1][2]; sc[5] += P.sigma[0][2];  // upper triangle, :517-519
    // F = gradX^-1 applied to eta: solve gX f = eta by Cramer's rule
    const double c00 = gX[1][1] * gX[2][2] - gX[1][2] * gX[2][1];
    const double c01 = gX[1][2] * gX[2][0] - gX[1][0] * gX[2][2];
    const double c02 = gX[1][0] * gX[2][1] - gX[1][1] * gX[2][0];
    const double s = 1.0 / (gX[0][0] * c00 + gX[0][1] * c01 + gX[0][2] * c02);
    const double F00 = c00 * s, F01 = (gX[0][2] * gX[2][1] - gX[0][1] * gX[2][2]) * s, F02 = (gX[0][1] * gX[1][2] - gX[0][2] * gX[1][1]) * s;
    const double F10 = c01 * s, F11 = (gX[0][0] * gX[2][2] - gX[0][2] * gX[2][0]) * s, F12 = (gX[0][2] * gX[1][0] - gX[0][0] * gX[1][2]) * s;
    const double F20 = c02 * s, F21 = (gX[0][1] * gX[2][0] - gX[0][0] * gX[2][1]) * s, F22 = (gX[0][0] * gX[1][1] - gX[0][1] * gX[1][0]) * s;
    fv[0] += F00 * eta[0] + F01 * eta[1] + F02 * eta[2];
    fv[1] += F10 * eta[0] + F11 * eta[1] + F12 * eta[2];
    fv[2] += F20 * eta[0] + F21 * eta[1] + F22 * eta[2];
  }
  const double inq = 1.0 / (double)Ref<NEN>::NQP;
#pragma unroll
  for (int x = 0; x < 6; x++) sc[x] *= inq;
  // principal-stress invariants: (e0+e1+e2)/3 = tr/3;  e0^2+e1^2+e2^2-e0e1-e0e2-e1e2 = I1^2 - 3 I2
  const double dev = sc[0] * sc[0] + sc[1] * sc[1] + sc[2] * sc[2] - sc[0] * sc[1] - sc[0] * sc[2] - sc[1] * sc[2] +
                     3.0 * (sc[3] * sc[3] + sc[4] * sc[4] + sc[5] * sc[5]);
  double* o = out + 5 * e;
  o[0] = (sc[0] + sc[1] + sc[2]) / 3.0;
  o[1] = sqrt(dev > 0.0 ? dev : 0.0);
  o[2] = fv[0] * inq; o[3] = fv[1] * inq; o[4] = fv[2] * inq;
}

hipError_t launch_solid_post(const SolidArgs& a, double* out) {
  const unsigned grid = (unsigned)((a.m.n_elem + 127) / 128);
  if (a.nen == 4)
    hipLaunchKernelGGL((k_solid_post<4>), dim3(grid), dim3(128), 0, a.stream, a.m, a.Xu, a.fibre, a.elem_material, a.materials,
                       a.params.pseudo_time, out);
  else
    hipLaunchKernelGGL((k_solid_post<8>), dim3(grid), dim3(128), 0, a.stream, a.m, a.Xu, a.fibre, a.elem_material, a.materials,
                       a.params.pseudo_time, out);
  return hipGetLastError();
}

hipError_t launch_solid(const SolidArgs& a) {
  if (a.kernel == 3) {
    if (a.cl.n_wg > 0) {   // a part of a two-part assembly may hold no cluster
      const hipError_t e = launch_solid_cl(a);
      if (e != hipSuccess) return e;
    }
  } else if (a.kernel == 0) {
    if (a.nen == 4) launch_two_pass<4>(a); else launch_two_pass<8>(a);
  } else
  for (int c = 0; c < a.n_colours; c++) {
    const int64_t first = a.colour_ptr[c], count = a.colour_ptr[c + 1] - first;
    if (count <= 0) continue;
    const int block = 128;
    const unsigned grid = (unsigned)((count + block - 1) / block);
    if (a.nen == 4)
      hipLaunchKernelGGL((k_solid_coloured<4>), dim3(grid), dim3(block), 0, a.stream, a.m, first, count, a.Xu, a.fibre,
                         a.elem_material, a.materials, a.params.pseudo_time, a.params.use_symmetry, a.request_jacobian,
                         a.val, a.rhs);
    else
      hipLaunchKernelGGL((k_solid_coloured<8>), dim3(grid), dim3(block), 0, a.stream, a.m, first, count, a.Xu, a.fibre,
                         a.elem_material, a.materials, a.params.pseudo_time, a.params.use_symmetry, a.request_jacobian,
                         a.val, a.rhs);
  }
  if (a.done_record) (void)hipEventRecord(a.done_record, a.stream);
  if (a.n_sides > 0) {
    if (a.sides_wait) (void)hipStreamWaitEvent(a.stream, a.sides_wait, 0);
    const int block = 64;
    const unsigned grid = (unsigned)((a.n_sides + block - 1) / block);
    if (a.nen == 4)
      hipLaunchKernelGGL((k_solid_sides<4>), dim3(grid), dim3(block), 0, a.stream, a.m, a.n_sides, a.side_elem, a.side_id,
                         a.side_disp, a.Xu, a.params.pseudo_time, a.params.displacement_penalty, a.request_jacobian, a.val,
                         a.rhs);
    else
      hipLaunchKernelGGL((k_solid_sides<8>), dim3(grid), dim3(block), 0, a.stream, a.m, a.n_sides, a.side_elem, a.side_id,
                         a.side_disp, a.Xu, a.params.pseudo_time, a.params.displacement_penalty, a.request_jacobian, a.val,
                         a.rhs);
  }
  return hipGetLastError();
}

}  // namespace rdc
