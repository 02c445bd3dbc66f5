// rdc_row.h — evaluation of ONE row-node of the element matrices of a reaction-diffusion model
// over all quadrature points (generic in model and element type).  Host+device so that the
// CPU-only test-suite can exercise exactly the code the kernels run (tests/host_shim.cpp);
// the shipped library only ever instantiates it inside HIP kernels.
#ifndef RDC_ROW_H
#define RDC_ROW_H
#include "rdc_fe.h"

namespace rdc {

template <class M, int NEN>
RDC_HD void rd_row_zero(double (&acc)[M::NV][M::NV][NEN], double (&fe)[M::NV]) {
#pragma unroll
  for (int a = 0; a < M::NV; a++) {
    fe[a] = 0.0;
#pragma unroll
    for (int b = 0; b < M::NV; b++)
#pragma unroll
      for (int j = 0; j < NEN; j++) acc[a][b][j] = 0.0;
  }
}

// ---- everything a quadrature point contributes to the row of local node `irow`, before the accumulation -------
template <class M, int NEN>
struct RowPoint {
  double N[NEN], G[NEN][3], W, Ni, Gi[3], gi[M::NG];
  typename M::C c;
};

template <class M, int NEN, int EXP_MODE>
RDC_HD void rd_point_setup(const typename M::K& k, const double (&X)[NEN][3], const double (&U)[NEN][M::NV],
                           const double (&AX)[NEN][M::NAUX > 0 ? M::NAUX : 1], int q, int irow, const double* ED,
                           RowPoint<M, NEN>& P) {
  constexpr int NV = M::NV, NG = M::NG, NA = (M::NAUX > 0 ? M::NAUX : 1);
  fe_point<NEN>(X, q, P.N, P.G, P.W);
  // old solution, aux fields and gradient fields at the point (src/pihna.C:429-442)
  double uq[NV], aq[NA], GF[NG][3];
#pragma unroll
  for (int v = 0; v < NV; v++) {
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < NEN; l++) s += P.N[l] * U[l][v];
    uq[v] = s;
  }
#pragma unroll
  for (int v = 0; v < NA; v++) {
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < NEN; l++) s += P.N[l] * AX[l][v];
    aq[v] = s;
  }
#pragma unroll
  for (int g = 0; g < NG; g++) {
    const int src = M::grad_src(g);
    if (src >= NV) { GF[g][0] = 0.0; GF[g][1] = 0.0; GF[g][2] = 0.0; continue; }  // filled by grad_post()
#pragma unroll
    for (int d = 0; d < 3; d++) {
      double s = 0.0;
#pragma unroll
      for (int l = 0; l < NEN; l++) s += P.G[l][d] * (src >= 0 ? U[l][(src >= 0 && src < NV) ? src : 0] : AX[l][src < 0 ? (-1 - src) % NA : 0]);
      GF[g][d] = s;
    }
    if (src < 0) {  // RIPF: unit radiotherapy gradient (src/ripf.C:481-484)
      const double l2 = sqrt(GF[g][0] * GF[g][0] + GF[g][1] * GF[g][1] + GF[g][2] * GF[g][2]);
      if (l2 != 0.0) { GF[g][0] /= l2; GF[g][1] /= l2; GF[g][2] /= l2; }
      else { GF[g][0] = 0.0; GF[g][1] = 0.0; GF[g][2] = 0.0; }
    }
  }
  if (M::NELEM > 0) M::grad_post(k, GF, ED);
  typename M::Pt pt;
  M::template point<EXP_MODE>(k, uq, aq, pt);
  M::coef(k, pt, P.c);
  // shape data of the row node
  P.Ni = 0.0; P.Gi[0] = 0.0; P.Gi[1] = 0.0; P.Gi[2] = 0.0;
#pragma unroll
  for (int n = 0; n < NEN; n++)
    if (n == irow) { P.Ni = P.N[n]; P.Gi[0] = P.G[n][0]; P.Gi[1] = P.G[n][1]; P.Gi[2] = P.G[n][2]; }
#pragma unroll
  for (int g = 0; g < NG; g++) P.gi[g] = GF[g][0] * P.Gi[0] + GF[g][1] * P.Gi[1] + GF[g][2] * P.Gi[2];
}

// The accumulation below is written per structurally non-zero coefficient (M::hasA / hasB / hasD / hasRG, the masks of
// the factored TET4 kernels): without fast-math the compiler may not drop `0.0 * x`, so the plain triple product
// over all NV x NV blocks costs 27 multiply-adds per column on a three-unknown model of which HCC needs 11.  The
// quadrature weight is folded into the row-node factors once per point (W*phi_i, W*grad phi_i, W*beta_ab), which
// leaves one FMA per non-zero coefficient and column.

// equation row A only (everything else of coef() is dead code): for element types / models whose full NV x NV x NEN
// accumulator does not fit the register file
template <class M, int NEN, int A>
RDC_HD void rd_point_accum_row(const RowPoint<M, NEN>& P, double (&acc)[M::NV][NEN], double& fe) {
  constexpr int NV = M::NV, NG = M::NG;
  double r = P.c.R[A] * P.Ni;
#pragma unroll
  for (int g = 0; g < NG; g++)
    if (M::hasRG(A, g)) r += P.c.RG[A][g] * P.gi[g];
  fe += P.W * r;
  const double wNi = P.W * P.Ni;
  const double wG[3] = {P.W * P.Gi[0], P.W * P.Gi[1], P.W * P.Gi[2]};
  double bgw[NV];
  bool anyB[NV];
#pragma unroll
  for (int b = 0; b < NV; b++) {
    double bg = 0.0;
    anyB[b] = false;
#pragma unroll
    for (int g = 0; g < NG; g++)
      if (M::hasB(A, b, g)) { bg += P.c.B[A][b][g] * P.gi[g]; anyB[b] = true; }
    bgw[b] = P.W * bg;
  }
#pragma unroll
  for (int j = 0; j < NEN; j++) {
    const double pp = P.N[j] * wNi;
    const double dd = P.G[j][0] * wG[0] + P.G[j][1] * wG[1] + P.G[j][2] * wG[2];
#pragma unroll
    for (int b = 0; b < NV; b++) {
      double v = acc[b][j];
      if (M::hasA(A, b)) v += P.c.A[A][b] * pp;
      if (anyB[b]) v += bgw[b] * P.N[j];
      if (M::hasD(A, b)) v += P.c.D[A][b] * dd;
      acc[b][j] = v;
    }
  }
}

// ---- contribution of quadrature point q to the row of local node `irow` ------------------------------
template <class M, int NEN, int EXP_MODE>
RDC_HD void rd_row_point(const typename M::K& k, const double (&X)[NEN][3], const double (&U)[NEN][M::NV],
                         const double (&AX)[NEN][M::NAUX > 0 ? M::NAUX : 1], int q, int irow,
                         double (&acc)[M::NV][M::NV][NEN], double (&fe)[M::NV], const double* ED) {
  constexpr int NV = M::NV, NG = M::NG;
  RowPoint<M, NEN> P;
  rd_point_setup<M, NEN, EXP_MODE>(k, X, U, AX, q, irow, ED, P);
#pragma unroll
  for (int a = 0; a < NV; a++) {
    double r = P.c.R[a] * P.Ni;
#pragma unroll
    for (int g = 0; g < NG; g++)
      if (M::hasRG(a, g)) r += P.c.RG[a][g] * P.gi[g];
    fe[a] += P.W * r;
  }
  const double wNi = P.W * P.Ni;
  const double wG[3] = {P.W * P.Gi[0], P.W * P.Gi[1], P.W * P.Gi[2]};
  double bgw[NV][NV];
  bool anyB[NV][NV];
#pragma unroll
  for (int a = 0; a < NV; a++)
#pragma unroll
    for (int b = 0; b < NV; b++) {
      double bg = 0.0;
      anyB[a][b] = false;
#pragma unroll
      for (int g = 0; g < NG; g++)
        if (M::hasB(a, b, g)) { bg += P.c.B[a][b][g] * P.gi[g]; anyB[a][b] = true; }
      bgw[a][b] = P.W * bg;
    }
#pragma unroll
  for (int j = 0; j < NEN; j++) {
    const double pp = P.N[j] * wNi;
    const double dd = P.G[j][0] * wG[0] + P.G[j][1] * wG[1] + P.G[j][2] * wG[2];
#pragma unroll
    for (int a = 0; a < NV; a++)
#pragma unroll
      for (int b = 0; b < NV; b++) {
        double v = acc[a][b][j];
        if (M::hasA(a, b)) v += P.c.A[a][b] * pp;
        if (anyB[a][b]) v += bgw[a][b] * P.N[j];
        if (M::hasD(a, b)) v += P.c.D[a][b] * dd;
        acc[a][b][j] = v;
      }
  }
}

// ---- one row (local node `irow`) of Ke and Fe over all quadrature points ------------------
template <class M, int NEN, int EXP_MODE>
RDC_HD void rd_row(const typename M::K& k, const double (&X)[NEN][3],
                                       const double (&U)[NEN][M::NV],
                                       const double (&AX)[NEN][M::NAUX > 0 ? M::NAUX : 1], int irow,
                                       double (&acc)[M::NV][M::NV][NEN], double (&fe)[M::NV],
                                       const double* ED = nullptr /* M::NELEM per-element inputs */) {
  rd_row_zero<M, NEN>(acc, fe);
#pragma unroll 1
  for (int q = 0; q < Ref<NEN>::NQP; q++) rd_row_point<M, NEN, EXP_MODE>(k, X, U, AX, q, irow, acc, fe, ED);
}

}  // namespace rdc
#endif
