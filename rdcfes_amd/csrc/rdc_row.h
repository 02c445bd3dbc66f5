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
  double N[NEN], dd[NEN], W, Ni, gi[M::NG];  // dd[j] = grad phi_j . grad phi_i,  gi[k] = grad f_k . grad phi_i
  typename M::C c;
};

template <class M, int NEN, int EXP_MODE>
RDC_HD void rd_point_setup(const typename M::K& k, const double (&X)[NEN][3], const double (&U)[NEN][M::NV],
                           const double (&AX)[NEN][M::NAUX > 0 ? M::NAUX : 1], int q, int irow, const double* ED,
                           RowPoint<M, NEN>& P) {
  constexpr int NV = M::NV, NG = M::NG, NA = (M::NAUX > 0 ? M::NAUX : 1);
  // old solution, aux fields and gradient fields at the point (src/pihna.C:429-442)
  double uq[NV], aq[NA], GF[NG][3], Gi[3];
  if constexpr (NEN == 8 && M::HEX_REF_GRADS) {
    // HEX8: reference gradients first, ONE application of the inverse Jacobian per vector that is needed in physical
    // space -- the row node's gradient and the NG field gradients; the other seven grad phi_j enter only through
    // dd[j] = grad phi_j . grad phi_i = dN_j . (Ji grad phi_i).
    double Ji[3][3];
    fe_jacobian8(X, q, Ji, P.W);
#pragma unroll
    for (int n = 0; n < 8; n++) P.N[n] = kHex8Tab.N[q][n];
    // the row node's reference shape data from its corner signs (irow differs per lane; no 8-way selects)
    double xi[3], wq;
    Ref<8>::qpoint(q, xi, wq);
    const double sx = Ref<8>::sx(irow), sy = Ref<8>::sy(irow), sz = Ref<8>::sz(irow);
    const double fa = 1.0 + sx * xi[0], fb = 1.0 + sy * xi[1], fc = 1.0 + sz * xi[2];
    P.Ni = 0.125 * fa * fb * fc;
    const double dNi[3] = {0.125 * sx * fb * fc, 0.125 * fa * sy * fc, 0.125 * fa * fb * sz};
#pragma unroll
    for (int r = 0; r < 3; r++) Gi[r] = dNi[0] * Ji[0][r] + dNi[1] * Ji[1][r] + dNi[2] * Ji[2][r];
    double bi[3];
#pragma unroll
    for (int c = 0; c < 3; c++) bi[c] = Ji[c][0] * Gi[0] + Ji[c][1] * Gi[1] + Ji[c][2] * Gi[2];
#pragma unroll
    for (int j = 0; j < 8; j++)
      P.dd[j] = kHex8Tab.dN[q][j][0] * bi[0] + kHex8Tab.dN[q][j][1] * bi[1] + kHex8Tab.dN[q][j][2] * bi[2];
#pragma unroll
    for (int g = 0; g < NG; g++) {
      const int src = M::grad_src(g);
      if (src >= NV) { GF[g][0] = 0.0; GF[g][1] = 0.0; GF[g][2] = 0.0; continue; }  // filled by grad_post()
      double gr[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < 8; l++)
          s += kHex8Tab.dN[q][l][c] * (src >= 0 ? U[l][(src >= 0 && src < NV) ? src : 0] : AX[l][src < 0 ? (-1 - src) % NA : 0]);
        gr[c] = s;
      }
#pragma unroll
      for (int r = 0; r < 3; r++) GF[g][r] = gr[0] * Ji[0][r] + gr[1] * Ji[1][r] + gr[2] * Ji[2][r];
    }
  } else {
    double G[NEN][3];
    fe_point<NEN>(X, q, P.N, G, P.W);
    P.Ni = 0.0; Gi[0] = 0.0; Gi[1] = 0.0; Gi[2] = 0.0;
#pragma unroll
    for (int n = 0; n < NEN; n++)
      if (n == irow) { P.Ni = P.N[n]; Gi[0] = G[n][0]; Gi[1] = G[n][1]; Gi[2] = G[n][2]; }
#pragma unroll
    for (int j = 0; j < NEN; j++) P.dd[j] = G[j][0] * Gi[0] + G[j][1] * Gi[1] + G[j][2] * Gi[2];
#pragma unroll
    for (int g = 0; g < NG; g++) {
      const int src = M::grad_src(g);
      if (src >= NV) { GF[g][0] = 0.0; GF[g][1] = 0.0; GF[g][2] = 0.0; continue; }  // filled by grad_post()
#pragma unroll
      for (int d = 0; d < 3; d++) {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < NEN; l++) s += G[l][d] * (src >= 0 ? U[l][(src >= 0 && src < NV) ? src : 0] : AX[l][src < 0 ? (-1 - src) % NA : 0]);
        GF[g][d] = s;
      }
    }
  }
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
    if (M::grad_src(g) < 0) {  // RIPF: unit radiotherapy gradient (src/ripf.C:481-484)
      const double l2 = sqrt(GF[g][0] * GF[g][0] + GF[g][1] * GF[g][1] + GF[g][2] * GF[g][2]);
      if (l2 != 0.0) { const double il = rcp(l2); GF[g][0] *= il; GF[g][1] *= il; GF[g][2] *= il; }
      else { GF[g][0] = 0.0; GF[g][1] = 0.0; GF[g][2] = 0.0; }
    }
  }
  if (M::NELEM > 0) M::grad_post(k, GF, ED);
  typename M::Pt pt;
  M::template point<EXP_MODE>(k, uq, aq, pt);
  M::coef(k, pt, P.c);
#pragma unroll
  for (int g = 0; g < NG; g++) P.gi[g] = GF[g][0] * Gi[0] + GF[g][1] * Gi[1] + GF[g][2] * Gi[2];
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
    const double dd = P.W * P.dd[j];
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

// ---- accumulation of one quadrature point, all equation rows (P: rd_point_setup, or the point record of rdc_hex8_cl.h) ----
template <class M, int NEN>
RDC_HD void rd_point_accum(const RowPoint<M, NEN>& P, double (&acc)[M::NV][M::NV][NEN], double (&fe)[M::NV]) {
  constexpr int NV = M::NV, NG = M::NG;
#pragma unroll
  for (int a = 0; a < NV; a++) {
    double r = P.c.R[a] * P.Ni;
#pragma unroll
    for (int g = 0; g < NG; g++)
      if (M::hasRG(a, g)) r += P.c.RG[a][g] * P.gi[g];
    fe[a] += P.W * r;
  }
  const double wNi = P.W * P.Ni;
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
    const double dd = P.W * P.dd[j];
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

// ---- contribution of quadrature point q to the row of local node `irow` ------------------------------
template <class M, int NEN, int EXP_MODE>
RDC_HD void rd_row_point(const typename M::K& k, const double (&X)[NEN][3], const double (&U)[NEN][M::NV],
                         const double (&AX)[NEN][M::NAUX > 0 ? M::NAUX : 1], int q, int irow,
                         double (&acc)[M::NV][M::NV][NEN], double (&fe)[M::NV], const double* ED) {
  RowPoint<M, NEN> P;
  rd_point_setup<M, NEN, EXP_MODE>(k, X, U, AX, q, irow, ED, P);
  rd_point_accum<M, NEN>(P, acc, fe);
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
