// rdc_tet4_pihna_moments.h — PIHNA (cell transport off, the shipped run/PIHNA/input.dat) on TET4 in MOMENT FORM.
//
// Every mass-type coefficient of the model is a fixed linear combination of a few point functions
//   A_ab(q) = sum_m K_abm * beta_m(q),     beta_m in {1, n, c, h, v, a, 1-Ve, P c, P h, Q c, Q h, Tau, dT c, ...}
// (P = Ve/(c+h+v), Q = (1-Ve)/(c+h+v); K_abm = products of the rates and of dt/2), and the factored TET4 row of
// rdc_tet4_fast.h is linear in A_ab(q).  So instead of evaluating all 21 coefficients at all 5 quadrature points and
// contracting each of them (~140 + 60 FP64 operations per point: half of the kernel's arithmetic), the row weights are
// applied ONCE per point function,
//   E(beta)[j] = sum_q JxW_q phi_0(q) phi_j(q) beta(q)    (9 operations per beta, j = 0..3)
//   r(beta)    = sum_q JxW_q phi_0(q) beta(q)
// and the entries are assembled from these moments with the constant factors: Ke_ab(0,j) = sum_m K_abm E(beta_m)[j].
// Same sums as src/pihna.C:427-750 in a different association (differences ~1e-16 relative); 1,374 -> ~900 FP64
// instructions per (node, element) pair.  Term-by-term correspondence with Pihna::coef() is noted per row.
#ifndef RDC_TET4_PIHNA_MOMENTS_H
#define RDC_TET4_PIHNA_MOMENTS_H
#include "rdc_integrands.h"
#include "rdc_tet4_fast.h"

namespace rdc {

// same model, same masks; only the evaluation order differs (selected by the host for the shipped parameter pattern)
struct PihnaNoCellTransportMoments : PihnaNoCellTransport {};

namespace mom {
struct V4 { double v[4]; };
struct Wts { double a0, b1, b2, d1, d2, o0, o1, o2; };
// E(beta)[j]: Tm + mh_j / 3 of tet4_row for m(q) = JxW_q phi_0(q) beta(q)
RDC_HD V4 E(const Wts& w, double b0, double b1, double b2, double b3, double b4) {
  const double base = w.a0 * b0 + w.b1 * b1 + w.b2 * ((b2 + b3) + b4);
  V4 r;
  r.v[0] = base + w.d1 * b1; r.v[1] = base + w.d2 * b2; r.v[2] = base + w.d2 * b3; r.v[3] = base + w.d2 * b4;
  return r;
}
RDC_HD double R(const Wts& w, double b0, double b1, double b2, double b3, double b4) {
  return w.o0 * b0 + w.o1 * b1 + w.o2 * ((b2 + b3) + b4);
}
}  // namespace mom

template <>
struct Tet4Rows<PihnaNoCellTransportMoments> {
  using M = PihnaNoCellTransportMoments;
  template <class Sink>
  RDC_HD static void run(const PihnaK& k, const Tet4Pre<M>& P, Sink& sink) {
    using mom::V4;
    const Pihna::Pt* s = P.pt;
    mom::Wts w;
    // JxW_q phi_0(q) {1/4 at c, 1/6 at h_k} and the extra phi_j(h_j) third: Wc/4/4, Wh/2/6, Wh/6/6, Wh/2/3, Wh/6/3
    w.a0 = P.Wc * (1.0 / 16.0); w.b1 = P.Wh * (1.0 / 12.0); w.b2 = P.Wh * (1.0 / 36.0);
    w.d1 = P.Wh * (1.0 / 6.0); w.d2 = P.Wh * (1.0 / 18.0);
    w.o0 = P.Wc * 0.25; w.o1 = P.Wh * 0.5; w.o2 = P.Wh * (1.0 / 6.0);
#define RDC_E(x) mom::E(w, x[0], x[1], x[2], x[3], x[4])
#define RDC_R(x) mom::R(w, x[0], x[1], x[2], x[3], x[4])
#define RDC_PT(name, expr) double name[5]; _Pragma("unroll") for (int q = 0; q < 5; q++) name[q] = (expr);
    V4 E1;
    {
      const double base = w.a0 + w.b1 + 3.0 * w.b2;
      E1.v[0] = base + w.d1; E1.v[1] = E1.v[2] = E1.v[3] = base + w.d2;
    }
    // The rows are emitted as soon as their moments exist, so that the LDS atomics of one equation row run under
    // the arithmetic of the next (all 89 of them bunched at the end leave the VALU idle while the LDS pipeline drains).
    // ---- vascular-fraction functions (Pihna::coef: oneVe, nVe_dc = Ve*rV =: P, Ve_dv = oneVe*rV =: Q) ----
    RDC_PT(oneVe, 1.0 - s[q].Ve)
    RDC_PT(Pq, s[q].Ve * s[q].rV)
    RDC_PT(Qq, oneVe[q] * s[q].rV)
    RDC_PT(Ph, Pq[q] * s[q].h)
    RDC_PT(Qh, Qq[q] * s[q].h)
    RDC_PT(oh, oneVe[q] * s[q].h)
    const V4 EoV = RDC_E(oneVe), EPh = RDC_E(Ph), EQh = RDC_E(Qh);
    const double r_oh = RDC_R(oh);
    // ---- unknowns and their products ----
    RDC_PT(pn, s[q].n)
    RDC_PT(pc, s[q].c)
    RDC_PT(ph, s[q].h)
    RDC_PT(pv, s[q].v)
    RDC_PT(cn, s[q].c * s[q].n)
    RDC_PT(hn, s[q].h * s[q].n)
    RDC_PT(vn, s[q].v * s[q].n)
    const V4 En = RDC_E(pn), Ec = RDC_E(pc), Eh = RDC_E(ph), Ev = RDC_E(pv);
    const double r_n = RDC_R(pn), r_c = RDC_R(pc), r_h = RDC_R(ph), r_v = RDC_R(pv);
    const double r_cn = RDC_R(cn), r_hn = RDC_R(hn), r_vn = RDC_R(vn);
    const double r_Veh = r_h - r_oh;  // Ve h = h - (1 - Ve) h
    // ---- n equation (coef: R[0], A[0][0..3]) ---------------------------------------------------------------------
    sink.fe(0, r_n + k.Tn_c * r_cn + k.Tn_h * r_hn + k.Tn_v * r_vn + k.Th2n * r_oh);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      sink.ke(0, 0, j, E1.v[j] - k.Tn_c * Ec.v[j] - k.Tn_h * Eh.v[j] - k.Tn_v * Ev.v[j]);
      sink.ke(0, 1, j, -(k.Tn_c * En.v[j] + k.Th2n * EPh.v[j]));
      sink.ke(0, 2, j, -(k.Tn_h * En.v[j] + k.Th2n * EPh.v[j] + k.Th2n * EoV.v[j]));
      sink.ke(0, 3, j, k.Th2n * EQh.v[j] - k.Tn_v * En.v[j]);
      sink.ke(0, 4, j, 0.0);
    }
    // ---- a equation (coef: R[4], A[4][1..4]) ---------------------------------------------------------------------
    {
      RDC_PT(pa, s[q].a)
      RDC_PT(va, s[q].v * s[q].a)
      const V4 Ea = RDC_E(pa);
      const double r_a = RDC_R(pa), r_va = RDC_R(va);
      sink.fe(4, r_a + k.Tsec_c * r_c + k.Tsec_h * r_h - k.Tupt * r_va - k.Tdec * r_a);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        sink.ke(4, 0, j, 0.0);
        sink.ke(4, 1, j, -(k.Tsec_c * E1.v[j]));
        sink.ke(4, 2, j, -(k.Tsec_h * E1.v[j]));
        sink.ke(4, 3, j, k.Tupt * Ea.v[j]);
        sink.ke(4, 4, j, E1.v[j] + k.Tupt * Ev.v[j] + k.Tdec * E1.v[j]);
      }
    }
    // ---- crowding functions and the remaining vascular-fraction products ----
    RDC_PT(Pc, Pq[q] * s[q].c)
    RDC_PT(Qc, Qq[q] * s[q].c)
    RDC_PT(oc, oneVe[q] * s[q].c)
    RDC_PT(tau, s[q].Tau)
    RDC_PT(dTc, s[q].dT * s[q].c)
    RDC_PT(Tc, s[q].Tau * s[q].c)
    const V4 EPc = RDC_E(Pc), EQc = RDC_E(Qc), ETau = RDC_E(tau), EdTc = RDC_E(dTc);
    const double r_oc = RDC_R(oc), r_Tc = RDC_R(Tc);
    // ---- c equation (coef: R[1], A[1][0..3]; pc = prod_c dT c) ----------------------------------------------------
    sink.fe(1, r_c + k.Tprod_c * r_Tc - k.Tc2h * r_oc + k.Th2c * r_Veh - k.Tn_c * r_cn);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const double pcj = k.Tprod_c * EdTc.v[j];
      const double x = k.Tc2h * EPc.v[j] + k.Th2c * EPh.v[j];
      sink.ke(1, 0, j, k.Tn_c * Ec.v[j] - pcj);
      sink.ke(1, 1, j, E1.v[j] - k.Tprod_c * ETau.v[j] - pcj + k.Tc2h * EoV.v[j] + x + k.Tn_c * En.v[j]);
      sink.ke(1, 2, j, x - pcj - k.Th2c * (E1.v[j] - EoV.v[j]));
      sink.ke(1, 3, j, -(pcj + k.Tc2h * EQc.v[j] + k.Th2c * EQh.v[j]));
      sink.ke(1, 4, j, 0.0);
    }
    // ---- h equation (coef: R[2], A[2][0..3]) ---------------------------------------------------------------------
    sink.fe(2, r_h + k.Tc2h * r_oc - k.Th2c * r_Veh - k.Tn_h * r_hn - k.Th2n * r_oh);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const double x = k.Tc2h * EPc.v[j] + k.Th2c * EPh.v[j] - k.Th2n * EPh.v[j];
      sink.ke(2, 0, j, k.Tn_h * Eh.v[j]);
      sink.ke(2, 1, j, -(k.Tc2h * EoV.v[j] + x));
      sink.ke(2, 2, j, E1.v[j] - x + k.Th2c * (E1.v[j] - EoV.v[j]) + k.Tn_h * En.v[j] + k.Th2n * EoV.v[j]);
      sink.ke(2, 3, j, k.Tc2h * EQc.v[j] + k.Th2c * EQh.v[j] - k.Th2n * EQh.v[j]);
      sink.ke(2, 4, j, 0.0);
    }
    // ---- v equation (coef: Ua, Ua_da, pv, R[3], RG[3][2], A[3][0..4], B[3][b][2], D[3][3]; gradient field 2 = v) ----
    {
      double g3[5], g4[5], tu[5], dvT[5], dvdT[5];
#pragma unroll
      for (int q = 0; q < 5; q++) {
        const double raK = rcp(s[q].a + k.Ka);
        const double Ua = s[q].a * raK, Ua_da = raK - Ua * raK;
        const double uav = Ua * s[q].v;
        g3[q] = s[q].dT * uav;
        tu[q] = s[q].Tau * uav;
        g4[q] = s[q].Tau * Ua_da * s[q].v;
        const bool on = s[q].v > k.Lambda;  // thresholded diffusion, :504-509
        dvT[q] = on ? s[q].Tau : 0.0;
        dvdT[q] = on ? s[q].dT : 0.0;
      }
      const V4 Eg3 = RDC_E(g3), Eg4 = RDC_E(g4);
      // JxW-weighted (no phi_0): the B term enters m(q) as W_q beta(q), the D and RG terms as sum_q W_q (.)
      const double sT = P.Wc * dvT[0] + P.Wh * ((dvT[1] + dvT[2]) + (dvT[3] + dvT[4]));
      const double bbase = P.Wc * 0.25 * dvdT[0] + P.Wh * (1.0 / 6.0) * ((dvdT[1] + dvdT[2]) + (dvdT[3] + dvdT[4]));
      const double bg = k.Tdif_v * P.gk[2];
      const double dcoef = k.Tdif_v * sT;
      sink.fe(3, r_v + k.Tprod_v * RDC_R(tu) - k.Tn_v * r_vn - dcoef * P.gk[2]);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const double bj = bg * (bbase + P.Wh * (1.0 / 3.0) * dvdT[j + 1]);
        const double pvj = bj - k.Tprod_v * Eg3.v[j];
        sink.ke(3, 0, j, pvj + k.Tn_v * Ev.v[j]);
        sink.ke(3, 1, j, pvj);
        sink.ke(3, 2, j, pvj);
        sink.ke(3, 3, j, E1.v[j] + pvj + k.Tn_v * En.v[j] + P.dd[j] * dcoef);
        sink.ke(3, 4, j, -(k.Tprod_v * Eg4.v[j]));
      }
    }
#undef RDC_E
#undef RDC_R
#undef RDC_PT
  }
};

}  // namespace rdc
#endif
