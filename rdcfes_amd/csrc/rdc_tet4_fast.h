// rdc_tet4_fast.h — factored evaluation of one row-node of the TET4 element matrices.
//
// On a linear tetrahedron grad(phi) and every interpolated gradient are constant, and the
// QGauss(THIRD) rule (centroid c + four "hot node" points h_k, phi_j(h_k) = 1/6 + delta_jk/3,
// phi_j(c) = 1/4) is invariant under vertex permutations.  With the row node
// rotated to local index 0 (even permutation j -> j XOR i, so det(J) keeps its sign) the row
//
//   Ke_ab(0,j) = sum_q JxW_q [ A_ab(q) phi_0(q) phi_j(q) + phi_j(q) beta_ab(q) + D_ab(q) dd_j ]
//
// with beta_ab(q) = sum_k B_ab^k(q) (grad f_k . grad phi_0), dd_j = grad phi_j . grad phi_0
// collapses to
//
//   m_ab(q)    = JxW_q [ A_ab(q) phi_0(q) + beta_ab(q) ]
//   Ke_ab(0,j) = [ m_ab(c)/4 + sum_k m_ab(h_k)/6 ]  +  m_ab(h_j)/3  +  dd_j * sum_q JxW_q D_ab(q)
//
// i.e. 5 point-coefficients per block instead of the 5 x 4 x 4 term loop of src/pihna.C:427-750.
// The function emits one equation row `a` at a time through a sink, so the 100-value row never
// has to live in registers at once.
#ifndef RDC_TET4_FAST_H
#define RDC_TET4_FAST_H
#include "rdc_fe.h"

namespace rdc {

// Element-level data shared by the five equation rows of one (row node, element) pair.
template <class M>
struct Tet4Pre {
  double dd[4];            // grad phi_j . grad phi_0
  double gk[M::NG];        // grad f_k . grad phi_0
  double Wc, Wh;           // JxW at the centroid / at a hot point
  typename M::Pt pt[5];    // point nonlinearities at c, h_0..h_3
};

template <class M, int EXP_MODE>
RDC_HD void tet4_prepare(const typename M::K& k, const double (&X)[4][3], const double (&U)[4][M::NV],
                         const double (&AX)[4][M::NAUX > 0 ? M::NAUX : 1], Tet4Pre<M>& P,
                         const double* ED = nullptr /* M::NELEM per-element inputs */) {
  constexpr int NV = M::NV, NG = M::NG, NA = (M::NAUX > 0 ? M::NAUX : 1);
  // ---- geometry: grad phi_1..3 = cofactors / det, grad phi_0 = -(sum) ----------------------
  double e1[3], e2[3], e3[3];
#pragma unroll
  for (int d = 0; d < 3; d++) { e1[d] = X[1][d] - X[0][d]; e2[d] = X[2][d] - X[0][d]; e3[d] = X[3][d] - X[0][d]; }
  double G[4][3];
  G[1][0] = e2[1] * e3[2] - e2[2] * e3[1]; G[1][1] = e2[2] * e3[0] - e2[0] * e3[2]; G[1][2] = e2[0] * e3[1] - e2[1] * e3[0];
  G[2][0] = e3[1] * e1[2] - e3[2] * e1[1]; G[2][1] = e3[2] * e1[0] - e3[0] * e1[2]; G[2][2] = e3[0] * e1[1] - e3[1] * e1[0];
  G[3][0] = e1[1] * e2[2] - e1[2] * e2[1]; G[3][1] = e1[2] * e2[0] - e1[0] * e2[2]; G[3][2] = e1[0] * e2[1] - e1[1] * e2[0];
  const double det = e1[0] * G[1][0] + e1[1] * G[1][1] + e1[2] * G[1][2];
  const double inv = rcp(det);
#pragma unroll
  for (int d = 0; d < 3; d++) {
    G[1][d] *= inv; G[2][d] *= inv; G[3][d] *= inv;
    G[0][d] = -(G[1][d] + G[2][d] + G[3][d]);
  }
  // libMesh requires det > 0; fabs() only protects the rotated view of a valid element
  const double adet = fabs(det);
#pragma unroll
  for (int j = 0; j < 4; j++) P.dd[j] = G[j][0] * G[0][0] + G[j][1] * G[0][1] + G[j][2] * G[0][2];
  // ---- constant gradient fields, projected on grad phi_0 ------------------------------------
  double GF[NG][3];
#pragma unroll
  for (int g = 0; g < NG; g++) {
    const int src = M::grad_src(g);
    if (src >= NV) { GF[g][0] = 0.0; GF[g][1] = 0.0; GF[g][2] = 0.0; continue; }  // filled by grad_post()
#pragma unroll
    for (int d = 0; d < 3; d++) {
      double s = 0.0;
#pragma unroll
      for (int l = 0; l < 4; l++) s += G[l][d] * (src >= 0 ? U[l][(src >= 0 && src < NV) ? src : 0] : AX[l][src < 0 ? (-1 - src) % NA : 0]);
      GF[g][d] = s;
    }
    if (src < 0) {  // unit radiotherapy gradient, src/ripf.C:481-484
      const double l2 = sqrt(GF[g][0] * GF[g][0] + GF[g][1] * GF[g][1] + GF[g][2] * GF[g][2]);
      if (l2 != 0.0) { const double il = rcp(l2); GF[g][0] *= il; GF[g][1] *= il; GF[g][2] *= il; }
      else { GF[g][0] = 0.0; GF[g][1] = 0.0; GF[g][2] = 0.0; }
    }
  }
  if (M::NELEM > 0) M::grad_post(k, GF, ED);
#pragma unroll
  for (int g = 0; g < NG; g++) P.gk[g] = GF[g][0] * G[0][0] + GF[g][1] * G[0][1] + GF[g][2] * G[0][2];
  // ---- point nonlinearities at c, h_0..h_3 (point index q: 0 = c, 1 + k = h_k) ---------------
  double S[NV], SA[NA];
#pragma unroll
  for (int v = 0; v < NV; v++) S[v] = (U[0][v] + U[1][v]) + (U[2][v] + U[3][v]);
#pragma unroll
  for (int v = 0; v < NA; v++) SA[v] = (AX[0][v] + AX[1][v]) + (AX[2][v] + AX[3][v]);
#pragma unroll
  for (int q = 0; q < 5; q++) {
    double uq[NV], aq[NA];
#pragma unroll
    for (int v = 0; v < NV; v++) uq[v] = (q == 0) ? 0.25 * S[v] : (S[v] * (1.0 / 6.0) + U[q == 0 ? 0 : q - 1][v] * (1.0 / 3.0));
#pragma unroll
    for (int v = 0; v < NA; v++) aq[v] = (q == 0) ? 0.25 * SA[v] : (SA[v] * (1.0 / 6.0) + AX[q == 0 ? 0 : q - 1][v] * (1.0 / 3.0));
    M::template point<EXP_MODE>(k, uq, aq, P.pt[q]);
  }
  P.Wc = adet * (-2.0 / 15.0);
  P.Wh = adet * 0.075;
}

// Sink interface:  void ke(int a, int b, int j, double v);   void fe(int a, double v);
// j is the ROTATED local column index (original local index = j ^ irow).
// One equation row `a` of the pair (a is a compile-time constant after unrolling/inlining).
template <class M, class Sink>
RDC_HD void tet4_row(const typename M::K& k, const Tet4Pre<M>& P, int a, Sink& sink) {
  constexpr int NV = M::NV, NG = M::NG;
  // JxW_q and JxW_q * phi_0(q)
  const double W[5] = {P.Wc, P.Wh, P.Wh, P.Wh, P.Wh};
  const double Om[5] = {P.Wc * 0.25, P.Wh * 0.5, P.Wh * (1.0 / 6.0), P.Wh * (1.0 / 6.0), P.Wh * (1.0 / 6.0)};
  double T[NV], Dh[NV], mh[NV][4], fe = 0.0, rgh[NG];
#pragma unroll
  for (int b = 0; b < NV; b++) { T[b] = 0.0; Dh[b] = 0.0; }
#pragma unroll
  for (int g = 0; g < NG; g++) rgh[g] = 0.0;
#pragma unroll
  for (int q = 0; q < 5; q++) {
    typename M::C c;
    M::coef(k, P.pt[q], c);  // only row `a` is consumed; the rest is dead code
    fe += Om[q] * c.R[a];
#pragma unroll
    for (int g = 0; g < NG; g++)
      if (M::hasRG(a, g)) rgh[g] += W[q] * c.RG[a][g];
#pragma unroll
    for (int b = 0; b < NV; b++) {
      double m = 0.0;
      bool any = false;
      if (M::hasA(a, b)) { m = Om[q] * c.A[a][b]; any = true; }
      double beta = 0.0;
      bool anyb = false;
#pragma unroll
      for (int g = 0; g < NG; g++)
        if (M::hasB(a, b, g)) { beta += c.B[a][b][g] * P.gk[g]; anyb = true; }
      if (anyb) { m += W[q] * beta; any = true; }
      if (any) {
        if (q == 0) T[b] += 0.25 * m;
        else { T[b] += m * (1.0 / 6.0); mh[b][q == 0 ? 0 : q - 1] = m; }
      } else if (q > 0) {
        mh[b][q - 1] = 0.0;
      }
      if (M::hasD(a, b)) Dh[b] += W[q] * c.D[a][b];
    }
  }
#pragma unroll
  for (int g = 0; g < NG; g++)
    if (M::hasRG(a, g)) fe += rgh[g] * P.gk[g];
  sink.fe(a, fe);
#pragma unroll
  for (int b = 0; b < NV; b++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      double v = T[b] + mh[b][j] * (1.0 / 3.0);
      if (M::hasD(a, b)) v += P.dd[j] * Dh[b];
      sink.ke(a, b, j, v);
    }
}

// How the rows of a pair are formed from the prepared element data.  Default: coefficient form, one equation row at
// a time (tet4_row).  A model may specialise this (rdc_tet4_pihna_moments.h).
template <class M>
struct Tet4Rows {
  template <class Sink>
  RDC_HD static void run(const typename M::K& k, const Tet4Pre<M>& P, Sink& sink) {
#pragma unroll
    for (int a = 0; a < M::NV; a++) tet4_row<M>(k, P, a, sink);
  }
};

// all rows of the pair
template <class M, int EXP_MODE, class Sink>
RDC_HD void tet4_row0(const typename M::K& k, const double (&X)[4][3], const double (&U)[4][M::NV],
                      const double (&AX)[4][M::NAUX > 0 ? M::NAUX : 1], Sink& sink, const double* ED = nullptr) {
  Tet4Pre<M> P;
  tet4_prepare<M, EXP_MODE>(k, X, U, AX, P, ED);
  Tet4Rows<M>::run(k, P, sink);
}

}  // namespace rdc
#endif
