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

hipError_t launch_solid(const SolidArgs& a) {
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
  if (a.n_sides > 0) {
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
