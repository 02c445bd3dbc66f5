// rdc_tet4_evc.h — one TET4 element VISIT in coefficient form, for any model without per-element inputs (RIPF
// src/ripf.C:382-553, coupled HCC src/coupled_hcc.C:433-645 and their parameter-pattern instantiations).
//
// The pair kernels (rdc_tet4_fast.h) redo the per-element part -- geometry, the gradient fields, the five point states
// with their exp / pow / sqrt -- in each of the element's four (row node, element) pairs.  The element-visit kernel
// (rdc_tet4_evc.hip, lists rdc_prep_ev.cpp) evaluates it once per visit and emits the rows of every cluster node of the
// element.  With the vertex-symmetric 5-point rule (phi_j(c) = 1/4, phi_j(h_k) = 1/6 + delta_jk/3) and constant gradients
//
//   Ke_ab(i,j) = sum_q JxW_q [ A_ab(q) phi_i phi_j + phi_j sum_g B_abg(q) (grad f_g . grad phi_i) + D_ab(q) grad phi_j . grad phi_i ]
//              = [S + t_i + t_j + 2 delta_ij t_i]  +  sum_g gk_i[g] (FS_g + f_g[j])  +  dd_ij Ds
//   S = Wc A(c)/16 + Wh sum_k A(h_k)/36,  t_k = Wh A(h_k)/18,  FS_g = Wc B_g(c)/4 + Wh sum_k B_g(h_k)/6,  f_g[j] = Wh B_g(h_j)/3,
//   Ds = sum_q JxW_q D(q),  gk_i[g] = grad f_g . grad phi_i,  dd_ij = grad phi_j . grad phi_i
//
// -- the same sums as tet4_row() with the row node left at its own local index instead of rotated to 0, so the point
// data is shared by all rows of the visit (per block: 5 + 5 NG + 1 numbers per visit, then 3-6 operations per entry).
#ifndef RDC_TET4_EVC_H
#define RDC_TET4_EVC_H
#include "rdc_fe.h"

namespace rdc {

// X, U, AX: the element's vertices with the `r` cluster-owned ones first (any vertex order is legal, as in rdc_tet4_ev.h).
// Sink: ke(a, b, i, j, v) = entry (equation a, unknown b) of block (vertex i, vertex j), i < r;  fe(a, i, v) = rhs entry.
template <class M, int EXP_MODE, class Sink>
RDC_HD void tet4_visit(const typename M::K& k, const double (&X)[4][3], const double (&U)[4][M::NV],
                       const double (&AX)[4][M::NAUX > 0 ? M::NAUX : 1], const int r, Sink& sink) {
  constexpr int NV = M::NV, NG = M::NG, NA = (M::NAUX > 0 ? M::NAUX : 1);
  static_assert(M::NELEM == 0 && M::AUX_LOCAL_NODE < 0, "models with per-element inputs stay on the pair kernels");
  // ---- geometry: grad phi_1..3 = cofactors / det, grad phi_0 = -(sum) -----------------------------------------------------
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
  const double adet = fabs(det);   // the owned-first vertex order may be an odd permutation
  const double Wc = adet * (-2.0 / 15.0), Wh = adet * 0.075;
  // ---- constant gradient fields ------------------------------------------------------------------------------------------
  double GF[NG][3];
#pragma unroll
  for (int g = 0; g < NG; g++) {
    const int src = M::grad_src(g);
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
  // ---- what depends on the row vertex -- dd_i[j] = grad phi_j . grad phi_i, gk_i[g] = grad f_g . grad phi_i -- is formed where
  // a row is emitted (3 FMAs each) rather than kept for the whole visit: 16 + 4 NG doubles fewer live registers
  // (Ripf: 232 -> 192 registers; 1.48 -> 1.42 ms on K(94))
  // ---- point states at c, h_0..h_3 (point index q: 0 = c, 1 + k = h_k) ---------------------------------------------------
  typename M::Pt pt[5];
  {
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
      M::template point<EXP_MODE>(k, uq, aq, pt[q]);
    }
  }
  // weights folded with the rule's constants once per visit (phi_j(c) = 1/4, phi_j(h_k) = 1/6 + delta_jk/3); the hot-point parts of
  // S and FS are half the sums of t and f (Wh/36 = (Wh/18)/2, Wh/6 = (Wh/3)/2): one multiplication per point and coefficient
  const double Wc16 = Wc * (1.0 / 16.0), Wh18 = Wh * (1.0 / 18.0), Wc4 = Wc * 0.25, Wh3 = Wh * (1.0 / 3.0);
  const double W[5] = {Wc, Wh, Wh, Wh, Wh};
  // ---- one equation row at a time: point coefficients -> (S, t), (FS, f), Ds per block -> entries of every owned row -----------
#pragma unroll
  for (int a = 0; a < NV; a++) {
    double Sb[NV], tb[NV][4], FS[NV][NG], fb[NV][NG][4], Ds[NV];
    double Rc = 0.0, Rh[4], RGs[NG];
#pragma unroll
    for (int b = 0; b < NV; b++) {
      Sb[b] = 0.0; Ds[b] = 0.0;
#pragma unroll
      for (int g = 0; g < NG; g++) FS[b][g] = 0.0;
    }
#pragma unroll
    for (int g = 0; g < NG; g++) RGs[g] = 0.0;
#pragma unroll
    for (int q = 0; q < 5; q++) {
      typename M::C c;
      M::coef(k, pt[q], c);   // only row `a` is consumed; the rest is dead code
      if (q == 0) Rc = W[0] * c.R[a]; else Rh[q == 0 ? 0 : q - 1] = W[q] * c.R[a];
#pragma unroll
      for (int g = 0; g < NG; g++)
        if (M::hasRG(a, g)) RGs[g] += W[q] * c.RG[a][g];
#pragma unroll
      for (int b = 0; b < NV; b++) {
        if (M::hasA(a, b)) {
          if (q == 0) Sb[b] = Wc16 * c.A[a][b];
          else tb[b][q == 0 ? 0 : q - 1] = Wh18 * c.A[a][b];
        }
#pragma unroll
        for (int g = 0; g < NG; g++)
          if (M::hasB(a, b, g)) {
            if (q == 0) FS[b][g] = Wc4 * c.B[a][b][g];
            else fb[b][g][q == 0 ? 0 : q - 1] = Wh3 * c.B[a][b][g];
          }
        if (M::hasD(a, b)) Ds[b] += W[q] * c.D[a][b];
      }
    }
#pragma unroll
    for (int b = 0; b < NV; b++) {
      if (M::hasA(a, b)) Sb[b] += 0.5 * ((tb[b][0] + tb[b][1]) + (tb[b][2] + tb[b][3]));
#pragma unroll
      for (int g = 0; g < NG; g++)
        if (M::hasB(a, b, g)) FS[b][g] += 0.5 * ((fb[b][g][0] + fb[b][g][1]) + (fb[b][g][2] + fb[b][g][3]));
    }
    const double Rs = Rc * 0.25 + ((Rh[0] + Rh[1]) + (Rh[2] + Rh[3])) * (1.0 / 6.0);
    // F_j = FS + f_j is the same for every row of the visit: formed once, in place
#pragma unroll
    for (int b = 0; b < NV; b++)
#pragma unroll
      for (int g = 0; g < NG; g++)
        if (M::hasB(a, b, g)) {
#pragma unroll
          for (int j = 0; j < 4; j++) fb[b][g][j] += FS[b][g];
        }
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (i < r) {
        double gk[NG], dd[4];
#pragma unroll
        for (int g = 0; g < NG; g++) gk[g] = GF[g][0] * G[i][0] + GF[g][1] * G[i][1] + GF[g][2] * G[i][2];
#pragma unroll
        for (int j = 0; j < 4; j++) dd[j] = G[j][0] * G[i][0] + G[j][1] * G[i][1] + G[j][2] * G[i][2];
        double fe = Rs + Rh[i] * (1.0 / 3.0);
#pragma unroll
        for (int g = 0; g < NG; g++)
          if (M::hasRG(a, g)) fe += RGs[g] * gk[g];
        sink.fe(a, i, fe);
#pragma unroll
        for (int b = 0; b < NV; b++) {
          bool any = M::hasA(a, b) || M::hasD(a, b);
#pragma unroll
          for (int g = 0; g < NG; g++) any = any || M::hasB(a, b, g);
          if (!any) continue;   // structurally zero block: never touched (the kernel writes its zeros)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            double v = 0.0;
            if (M::hasA(a, b)) v = (j == i) ? Sb[b] + 4.0 * tb[b][i] : Sb[b] + tb[b][i] + tb[b][j];
#pragma unroll
            for (int g = 0; g < NG; g++)
              if (M::hasB(a, b, g)) v += gk[g] * fb[b][g][j];
            if (M::hasD(a, b)) v += dd[j] * Ds[b];
            sink.ke(a, b, i, j, v);
          }
        }
      }
  }
}

// block (a, b) of the model is structurally non-zero, and its index among the non-zero blocks in (a, b) order
template <class M> RDC_HD constexpr bool evc_block(int a, int b) {
  bool nz = M::hasA(a, b) || M::hasD(a, b);
  for (int g = 0; g < M::NG; g++) nz = nz || M::hasB(a, b, g);
  return nz;
}
template <class M> RDC_HD constexpr int evc_index(int a, int b) {
  int n = 0;
  for (int x = 0; x < a * M::NV + b; x++) n += evc_block<M>(x / M::NV, x % M::NV) ? 1 : 0;
  return n;
}
template <class M> RDC_HD constexpr int evc_blocks() { return evc_index<M>(M::NV, 0); }

}  // namespace rdc
#endif
