// rdc_tet4_ev.h — PIHNA (cell transport off: the shipped run/PIHNA/input.dat pattern) on TET4, evaluated ONCE PER
// ELEMENT VISIT and accumulated as MOMENTS.
//
// k_tet4_rg5 runs one thread per (row node, element) pair: the per-element part (geometry, the point nonlinearities,
// the point functions) is repeated by the four pairs of an element, and every pair combines ~16 moment vectors into its
// 21 x 4 matrix entries.  Here a workgroup owns a compact CLUSTER of nodes; one thread visits one element that touches
// the cluster, evaluates the per-element part once and emits the rows of all of the element's nodes that the cluster
// owns (the host orders the element's vertices owned-first; on a K(n) mesh a visit serves 1.8-2.2 rows).  What is
// accumulated in LDS are not the 25 entries of a node block but the 16 MOMENTS they are linear combinations of,
//
//   Ke_ab(I,J) = sum_m K_abm G_m(I,J),      G_m(I,J) = sum_e sum_q JxW_q phi_i(q) phi_j(q) beta_m(q)
//
// (K_abm = products of the rates and dt/2), so that the combination with the rates is done once per node block by the
// expansion phase of the kernel instead of once per contribution.  On TET4 with the vertex-symmetric 5-point rule
// (centroid c + "hot" points h_k, phi_j(h_k) = 1/6 + delta_jk/3)
//
//   E_m(i,j) = S_m + t_m,i + t_m,j + 2 delta_ij t_m,i,     S_m = Wc/16 beta_m(c) + Wh/36 sum_k beta_m(h_k),  t_m,k = Wh/18 beta_m(h_k)
//
// The same sums as src/pihna.C:427-750 in another association (differences ~1e-16 relative); term-by-term
// correspondence with Pihna::coef() as in rdc_tet4_pihna_moments.h, from which the moment list is taken.
#ifndef RDC_TET4_EV_H
#define RDC_TET4_EV_H
#include "rdc_integrands.h"

namespace rdc {
namespace ev {

constexpr int NM = 16;    // moments per node block (shipped parameter pattern: cell transport off)
constexpr int NMG = 22;   // ... with every term on (the six transport moments of the c and h rows behind the 16)
constexpr int NBP = 256;  // node blocks per workgroup (padded): moment m of block b lives at M[m * NBP + b]
constexpr int MAXN = 16;  // owned nodes per workgroup
enum { M_E1, M_EoV, M_EPh, M_EQh, M_En, M_Ec, M_Eh, M_Ev, M_Ea, M_Ex, M_Ey, M_ETau, M_EdTc, M_Epv, M_Eg4, M_Edd,
       M_Xc, M_Yc, M_Zc, M_Xh, M_Yh, M_Zh };
constexpr bool symmetric_moment(int m) { return m != M_Epv; }   // E_m(i, j) == E_m(j, i)
// general parameters: the taxis parts gk_i F_j are not symmetric; M_Edd then also carries the taxis part of block (v, v)
constexpr bool symmetric_moment_gen(int m) { return m != M_Epv && m != M_Edd && m != M_Xc && m != M_Yc && m != M_Xh && m != M_Yh; }

// One element visit.  X, U: the element's vertices with the `r` cluster-owned ones first (any vertex order is legal:
// grad phi = cofactor / det and every product used here is invariant under the orientation).
// Sink: mom(m, i, j, v) adds v to moment m of block (node i, node j), i < r;  rhs(a, i, v) adds to rhs entry a of node i.
// MIRROR: every moment but M_Epv is symmetric in (i, j), and a column j < i of row i < r is itself an owned row: such a
// contribution is only added to block (j -> i) by row j, and the expansion adds the moments of the MIRROR block
// (column node -> row node, HostPrepEv::bpart) to a block's own.  Row position i then issues 15 (4 - i) + 4 + 5 atomics
// instead of 69 -- the under-filled high row positions of a wave are the cheap ones.
// GEN: every term of Pihna::coef() (diffuse/c, taxis/c, diffuse/h, taxis/h, taxis/v != 0): with the gradient projections
// gk_i[f] = grad f . grad phi_i, dd_ij = grad phi_i . grad phi_j and the first moments F_j(beta) = sum_q JxW_q beta(q) phi_j(q)
//   X_c(i,j) = gk_i[c] F_j(T dif_c dT) + gk_i[v] F_j(T tax_c dT c)          -> blocks (c, n..v)   (coef: B[1][b][0], B[1][b][2])
//   Y_c(i,j) = gk_i[v] F_j(T tax_c Tau) + dd_ij sum_q JxW_q T dif_c Tau     -> block (c, c)       (B[1][1][2] extra, D[1][1])
//   Z_c(i,j) = dd_ij sum_q JxW_q T tax_c Tau c                              -> block (c, v)       (D[1][3])
// (transport coefficients thresholded per point, src/pihna.C:504-509), the same for h, and for taxis/v: B[3][b][3] joins M_Epv,
// the extra of B[3][3][3] joins M_Edd (both feed block (v, v) only), D[3][4] joins M_Eg4 (block (v, a)).
//
// bg ("background"): the caller knows that n = c = h = a = 0 and v > 0 at the four vertices of every visit it passes together (the
// device: of every active lane of the wave -- a scalar branch).  Then Ve = 1 and 1 - Ve = 0 (src/pihna.C:474-499), and the nine
// moments E(1-Ve), E(P h), E(Q h), E(n), E(c), E(h), x, y, E(dTau c), the transport moments of the c and h rows and the right-hand
// sides of the n, c, h, a equations are sums of exact zeros: they are not evaluated and nothing is added for them.  The shipped
// initial field (run/PIHNA/Brain_Model_Initial_Nodal_Field.dat) is this state at 24,880 of its 24,903 nodes.
// (a run-time flag: as two instantiations inlined side by side the kernel spills 30 registers instead of 3.)
template <int EXP_MODE, class Sink, bool MIRROR = true, bool GEN = false>
RDC_HD void pihna_visit(const PihnaK& k, const double (&X)[4][3], const double (&U)[4][5], const int r, Sink& sink, const bool bg = false) {
  // ---- geometry: unscaled cofactors g_j = det * grad phi_j ------------------------------------------------------
  double e1[3], e2[3], e3[3];
#pragma unroll
  for (int d = 0; d < 3; d++) { e1[d] = X[1][d] - X[0][d]; e2[d] = X[2][d] - X[0][d]; e3[d] = X[3][d] - X[0][d]; }
  double g[4][3];
  g[1][0] = e2[1] * e3[2] - e2[2] * e3[1]; g[1][1] = e2[2] * e3[0] - e2[0] * e3[2]; g[1][2] = e2[0] * e3[1] - e2[1] * e3[0];
  g[2][0] = e3[1] * e1[2] - e3[2] * e1[1]; g[2][1] = e3[2] * e1[0] - e3[0] * e1[2]; g[2][2] = e3[0] * e1[1] - e3[1] * e1[0];
  g[3][0] = e1[1] * e2[2] - e1[2] * e2[1]; g[3][1] = e1[2] * e2[0] - e1[0] * e2[2]; g[3][2] = e1[0] * e2[1] - e1[1] * e2[0];
  const double det = e1[0] * g[1][0] + e1[1] * g[1][1] + e1[2] * g[1][2];
  const double inv = rcp(det), inv2 = inv * inv;
  const double adet = fabs(det);
#pragma unroll
  for (int d = 0; d < 3; d++) g[0][d] = -(g[1][d] + g[2][d] + g[3][d]);
  const double Wc = adet * (-2.0 / 15.0), Wh = adet * 0.075;
  const double a0 = Wc * (1.0 / 16.0), b36 = Wh * (1.0 / 36.0), w18 = Wh * (1.0 / 18.0);
  // ---- point values of the unknowns and the point nonlinearities (q = 0: centroid, 1 + k: hot point k) ------------
  Pihna::Pt s[5];
  {
    double S[5];
#pragma unroll
    for (int v = 0; v < 5; v++) S[v] = (U[0][v] + U[1][v]) + (U[2][v] + U[3][v]);
#pragma unroll
    for (int q = 0; q < 5; q++) {
      double uq[5];
#pragma unroll
      for (int v = 0; v < 5; v++) uq[v] = (q == 0) ? 0.25 * S[v] : (S[v] * (1.0 / 6.0) + U[q == 0 ? 0 : q - 1][v] * (1.0 / 3.0));
      Pihna::point<EXP_MODE>(k, uq, nullptr, s[q]);
    }
  }
  // symmetric (mass-type) moment of a point function: rows of the owned vertices
#define RDC_EV_SYM(m, beta)                                                                              \
  {                                                                                                      \
    const double Sm_ = a0 * beta[0] + b36 * ((beta[1] + beta[2]) + (beta[3] + beta[4]));                 \
    const double t_[4] = {w18 * beta[1], w18 * beta[2], w18 * beta[3], w18 * beta[4]};                   \
    _Pragma("unroll") for (int i = 0; i < 4; i++) if (i < r) {                                           \
      const double rb_ = Sm_ + t_[i];                                                                    \
      _Pragma("unroll") for (int j = 0; j < 4; j++) if (!MIRROR || j >= i) sink.mom(m, i, j, j == i ? rb_ + 3.0 * t_[i] : rb_ + t_[j]); \
    }                                                                                                    \
  }
#define RDC_EV_PT(name, expr) double name[5]; _Pragma("unroll") for (int q = 0; q < 5; q++) name[q] = (expr);
  // right-hand sides from the pointwise G_a(q) (Pihna::coef R[a], the factor T folded into k.T*):
  //   fe_a(i) = sum_q JxW_q phi_i(q) G_a(q) = Wc/4 G_a(c) + Wh/6 sum_k G_a(h_k) + Wh/3 G_a(h_i)
  // one equation at a time, emitted at once (25 live doubles otherwise)
  const double o0 = Wc * 0.25, o6 = Wh * (1.0 / 6.0), o3 = Wh * (1.0 / 3.0);
#define RDC_EV_RHS(a, G)                                                             \
    {                                                                                \
      const double base_ = o0 * G[0] + o6 * ((G[1] + G[2]) + (G[3] + G[4]));         \
      _Pragma("unroll") for (int i = 0; i < 4; i++) if (i < r) sink.rhs(a, i, base_ + o3 * G[i + 1]); \
    }
  // ---- v equation first: it is the only one that needs the gradients (coef: Ua, pv, R[3], RG[3][2], A[3][*], B[3][b][2], D[3][3]) ----
  // (staged so that few point arrays are alive at a time: the kernel has to fit 168 registers)
  {
    double uav[5];
    {
      double g4[5];
#pragma unroll
      for (int q = 0; q < 5; q++) {
        const double raK = rcp(s[q].a + k.Ka);
        const double Ua = s[q].a * raK, Ua_da = raK - Ua * raK;
        uav[q] = Ua * s[q].v;
        g4[q] = -k.Tprod_v * (s[q].Tau * Ua_da * s[q].v);     // (3,4) = -Tprod_v * E(Tau Ua_da v)
      }
      RDC_EV_SYM(M_Eg4, g4)
    }
    {
      RDC_EV_PT(G3, s[q].v + k.Tprod_v * (s[q].Tau * uav[q]) - k.Tn_v * (s[q].v * s[q].n))
      RDC_EV_RHS(3, G3)
    }
    // -Tprod_v E(dT Ua v): enters pv with its factor
    double Sm_, t_[4];
    {
      RDC_EV_PT(g3, -k.Tprod_v * (s[q].dT * uav[q]))
      Sm_ = a0 * g3[0] + b36 * ((g3[1] + g3[2]) + (g3[3] + g3[4]));
#pragma unroll
      for (int j = 0; j < 4; j++) t_[j] = w18 * g3[j + 1];
    }
    // thresholded diffusion of v, src/pihna.C:504-509: Tau and dTau where v > Lambda
    double sT, FS, f_[4];
    {
      RDC_EV_PT(dvT, s[q].v > k.Lambda ? s[q].Tau : 0.0)
      sT = k.Tdif_v * (Wc * dvT[0] + Wh * ((dvT[1] + dvT[2]) + (dvT[3] + dvT[4])));   // dcoef of the pair kernels
    }
    {
      RDC_EV_PT(dvdT, s[q].v > k.Lambda ? s[q].dT : 0.0)
      FS = Wc * 0.25 * dvdT[0] + Wh * (1.0 / 6.0) * ((dvdT[1] + dvdT[2]) + (dvdT[3] + dvdT[4]));
#pragma unroll
      for (int j = 0; j < 4; j++) f_[j] = Wh * (1.0 / 3.0) * dvdT[j + 1];
    }
    // gradient of v (unscaled) and its projections: gk_i = grad v . grad phi_i = (gv . g_i) / det^2
    double gv[3];
#pragma unroll
    for (int d = 0; d < 3; d++) gv[d] = (U[0][3] * g[0][d] + U[1][3] * g[1][d]) + (U[2][3] * g[2][d] + U[3][3] * g[3][d]);
    const double sTi = sT * inv2, Td2 = k.Tdif_v * inv2;
    // taxis/v (GEN): gradient of a; F(tax_v dT v) joins pv, F(tax_v Tau) joins block (v, v), sum tax_v Tau v joins (v, a) and the rhs
    double ga[3] = {0.0, 0.0, 0.0}, FSa = 0.0, fa_[4] = {0.0, 0.0, 0.0, 0.0}, FSy = 0.0, fy_[4] = {0.0, 0.0, 0.0, 0.0}, sZv = 0.0;
    if (GEN) {
#pragma unroll
      for (int d = 0; d < 3; d++) ga[d] = (U[0][4] * g[0][d] + U[1][4] * g[1][d]) + (U[2][4] * g[2][d] + U[3][4] * g[3][d]);
      const double Tt = k.Ttax_v * inv2;
      {
        RDC_EV_PT(b, s[q].v > k.Lambda ? s[q].dT * s[q].v : 0.0)
        FSa = Tt * (Wc * 0.25 * b[0] + Wh * (1.0 / 6.0) * ((b[1] + b[2]) + (b[3] + b[4])));
#pragma unroll
        for (int j = 0; j < 4; j++) fa_[j] = Tt * (Wh * (1.0 / 3.0)) * b[j + 1];
      }
      {
        RDC_EV_PT(b, s[q].v > k.Lambda ? s[q].Tau : 0.0)
        FSy = Tt * (Wc * 0.25 * b[0] + Wh * (1.0 / 6.0) * ((b[1] + b[2]) + (b[3] + b[4])));
#pragma unroll
        for (int j = 0; j < 4; j++) fy_[j] = Tt * (Wh * (1.0 / 3.0)) * b[j + 1];
      }
      {
        RDC_EV_PT(b, s[q].v > k.Lambda ? s[q].Tau * s[q].v : 0.0)
        sZv = Tt * (Wc * b[0] + Wh * ((b[1] + b[2]) + (b[3] + b[4])));
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) if (i < r) {
      const double gki = gv[0] * g[i][0] + gv[1] * g[i][1] + gv[2] * g[i][2];   // unscaled
      const double gkT = Td2 * gki;                                               // Tdif_v * grad v . grad phi_i
      const double gai = GEN ? ga[0] * g[i][0] + ga[1] * g[i][1] + ga[2] * g[i][2] : 0.0;   // unscaled (the factor 1 / det^2 is in FSa, fa_, FSy, fy_, sZv)
      double rb_ = Sm_ + t_[i] + gkT * FS;
      if (GEN) rb_ += gai * FSa;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        // pv(i,j) = Tdif_v gk_i F(dT_on)(j) - Tprod_v E(dT Ua v)(i,j)        (coef: B[3][b][2] and -T pv)  [+ Ttax_v gk_i[a] F(dT v)(j): B[3][b][3]]
        double pv_ = (j == i ? rb_ + 3.0 * t_[i] : rb_ + t_[j]) + gkT * f_[j];
        if (GEN) pv_ += gai * fa_[j];
        sink.mom(M_Epv, i, j, pv_);
        // dd(i,j) * Tdif_v * sum_q JxW_q Tau_on(q)                            (coef: D[3][3])  [+ Ttax_v gk_i[a] F(Tau)(j): the extra of B[3][3][3]]
        const double dd_ = g[i][0] * g[j][0] + g[i][1] * g[j][1] + g[i][2] * g[j][2];
        if (GEN) {
          sink.mom(M_Edd, i, j, sTi * dd_ + gai * (FSy + fy_[j]));
          if (!MIRROR || j >= i) sink.mom(M_Eg4, i, j, sZv * dd_);              // D[3][4]: block (v, a), on top of M_Eg4's own part below
        } else if (!MIRROR || j >= i) sink.mom(M_Edd, i, j, sTi * dd_);
      }
      sink.rhs(3, i, GEN ? -(sTi * gki + sZv * gai) : -(sTi * gki));              // RG[3][2] (, RG[3][3]): -dcoef * gk_i
    }
  }
  // ---- cell transport (GEN): rows c (species 1) and h (species 2), gradient fields (own, v) ---------------------------------
  if (GEN && !bg) {
#pragma unroll
    for (int sp = 1; sp <= 2; sp++) {
      const double Tdif = sp == 1 ? k.Tdif_c : k.Tdif_h, Ttax = sp == 1 ? k.Ttax_c : k.Ttax_h;
      if (Tdif == 0.0 && Ttax == 0.0) continue;   // uniform
      double go[3], gv[3];
#pragma unroll
      for (int d = 0; d < 3; d++) {
        go[d] = (U[0][sp] * g[0][d] + U[1][sp] * g[1][d]) + (U[2][sp] * g[2][d] + U[3][sp] * g[3][d]);
        gv[d] = (U[0][3] * g[0][d] + U[1][3] * g[1][d]) + (U[2][3] * g[2][d] + U[3][3] * g[3][d]);
      }
      // first moments (FS + f_j) of the thresholded point functions, zero-order sums; all carry 1 / det^2 for the unscaled projections
      double FS1, f1[4], FS2, f2[4], FS3, f3[4], sD, sZ;
      const double Td = Tdif * inv2, Tt = Ttax * inv2;
#define RDC_EV_F(FSx, fx, scale, expr)                                                                     \
      {                                                                                                     \
        RDC_EV_PT(b, ((sp == 1 ? s[q].c : s[q].h) > k.Lambda) ? (expr) : 0.0)                             \
        FSx = (scale) * (Wc * 0.25 * b[0] + Wh * (1.0 / 6.0) * ((b[1] + b[2]) + (b[3] + b[4])));         \
        _Pragma("unroll") for (int j = 0; j < 4; j++) fx[j] = (scale) * (Wh * (1.0 / 3.0)) * b[j + 1];   \
      }
      RDC_EV_F(FS1, f1, Td, s[q].dT)
      RDC_EV_F(FS2, f2, Tt, s[q].dT * (sp == 1 ? s[q].c : s[q].h))
      RDC_EV_F(FS3, f3, Tt, s[q].Tau)
#undef RDC_EV_F
      {
        RDC_EV_PT(b, ((sp == 1 ? s[q].c : s[q].h) > k.Lambda) ? s[q].Tau : 0.0)
        sD = Td * (Wc * b[0] + Wh * ((b[1] + b[2]) + (b[3] + b[4])));
      }
      {
        RDC_EV_PT(b, ((sp == 1 ? s[q].c : s[q].h) > k.Lambda) ? s[q].Tau * (sp == 1 ? s[q].c : s[q].h) : 0.0)
        sZ = Tt * (Wc * b[0] + Wh * ((b[1] + b[2]) + (b[3] + b[4])));
      }
      const int mX = sp == 1 ? M_Xc : M_Xh, mY = sp == 1 ? M_Yc : M_Yh, mZ = sp == 1 ? M_Zc : M_Zh;
#pragma unroll
      for (int i = 0; i < 4; i++) if (i < r) {
        const double gko = go[0] * g[i][0] + go[1] * g[i][1] + go[2] * g[i][2];   // unscaled projections
        const double gkv = gv[0] * g[i][0] + gv[1] * g[i][1] + gv[2] * g[i][2];
        const double xb = gko * FS1 + gkv * FS2, yb = gkv * FS3;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const double dd_ = g[i][0] * g[j][0] + g[i][1] * g[j][1] + g[i][2] * g[j][2];
          sink.mom(mX, i, j, xb + gko * f1[j] + gkv * f2[j]);
          sink.mom(mY, i, j, yb + gkv * f3[j] + sD * dd_);
          if (!MIRROR || j >= i) sink.mom(mZ, i, j, sZ * dd_);
        }
        sink.rhs(sp, i, -(sD * gko + sZ * gkv));                                  // RG[sp][own], RG[sp][2]
      }
    }
  }
  // ---- vascular-fraction functions (coef: oneVe, nVe_dc = Ve rV =: P, Ve_dv = oneVe rV =: Q) ----------------------
  if (!bg) {
    RDC_EV_PT(oneVe, 1.0 - s[q].Ve)
    RDC_EV_PT(Pq, s[q].Ve * s[q].rV)
    RDC_EV_PT(Qq, oneVe[q] * s[q].rV)
    RDC_EV_SYM(M_EoV, oneVe)
    {
      RDC_EV_PT(Ph, Pq[q] * s[q].h)
      RDC_EV_SYM(M_EPh, Ph)
      RDC_EV_PT(xx, k.Tc2h * (Pq[q] * s[q].c) + k.Th2c * Ph[q])        // x = Tc2h E(P c) + Th2c E(P h)
      RDC_EV_SYM(M_Ex, xx)
    }
    {
      RDC_EV_PT(Qh, Qq[q] * s[q].h)
      RDC_EV_SYM(M_EQh, Qh)
      RDC_EV_PT(yy, k.Tc2h * (Qq[q] * s[q].c) + k.Th2c * Qh[q])        // y = Tc2h E(Q c) + Th2c E(Q h)
      RDC_EV_SYM(M_Ey, yy)
    }
    {
      RDC_EV_PT(G, s[q].n + k.Tn_c * (s[q].c * s[q].n) + k.Tn_h * (s[q].h * s[q].n) + k.Tn_v * (s[q].v * s[q].n) + k.Th2n * (oneVe[q] * s[q].h))
      RDC_EV_RHS(0, G)
    }
    {
      RDC_EV_PT(G, s[q].c + k.Tprod_c * (s[q].Tau * s[q].c) - k.Tc2h * (oneVe[q] * s[q].c) + k.Th2c * (s[q].h - oneVe[q] * s[q].h) - k.Tn_c * (s[q].c * s[q].n))
      RDC_EV_RHS(1, G)
    }
    {
      RDC_EV_PT(G, s[q].h + k.Tc2h * (oneVe[q] * s[q].c) - k.Th2c * (s[q].h - oneVe[q] * s[q].h) - k.Tn_h * (s[q].h * s[q].n) - k.Th2n * (oneVe[q] * s[q].h))
      RDC_EV_RHS(2, G)
    }
  }
  if (!bg) {
    RDC_EV_PT(G, s[q].a + k.Tsec_c * s[q].c + k.Tsec_h * s[q].h - k.Tupt * (s[q].v * s[q].a) - k.Tdec * s[q].a)
    RDC_EV_RHS(4, G)
  }
#undef RDC_EV_RHS
  // ---- the unknowns themselves and the crowding functions ----------------------------------------------------------
  {
    const double one[5] = {1.0, 1.0, 1.0, 1.0, 1.0};
    RDC_EV_SYM(M_E1, one)
  }
  if (!bg) {
    { RDC_EV_PT(b, s[q].n) RDC_EV_SYM(M_En, b) }
    { RDC_EV_PT(b, s[q].c) RDC_EV_SYM(M_Ec, b) }
    { RDC_EV_PT(b, s[q].h) RDC_EV_SYM(M_Eh, b) }
    { RDC_EV_PT(b, s[q].dT * s[q].c) RDC_EV_SYM(M_EdTc, b) }
  }
  { RDC_EV_PT(b, s[q].v) RDC_EV_SYM(M_Ev, b) }
  // E(a) only enters through the uptake rate (pihna_expand: o[23] = Tupt E(a)): with uptake/a/from/v = 0, the shipped value, its ten
  // atomics per visit are not issued (k is uniform: a scalar branch); the slice keeps its zeros and o[23] = 0 * 0
  if (k.Tupt != 0.0) { RDC_EV_PT(b, s[q].a) RDC_EV_SYM(M_Ea, b) }
  { RDC_EV_PT(b, s[q].Tau) RDC_EV_SYM(M_ETau, b) }
#undef RDC_EV_SYM
#undef RDC_EV_PT
}

// a visit in the background state (pihna_visit, bg)
RDC_HD bool pihna_background(const double (&U)[4][5]) {
  bool b = true;
#pragma unroll
  for (int j = 0; j < 4; j++) b = b && U[j][0] == 0.0 && U[j][1] == 0.0 && U[j][2] == 0.0 && U[j][4] == 0.0 && U[j][3] > 0.0;
  return b;
}

// The 25 entries of a node block from its 16 moments (e[m]); o[a * 5 + b].  Same formulas as the rows of
// rdc_tet4_pihna_moments.h (coef: A[a][b] with the factor -T, the mass term on the diagonal blocks).
template <int NMT>
RDC_HD void pihna_expand(const PihnaK& k, const double (&e)[NMT], double (&o)[25]) {
  const double E1 = e[M_E1], EoV = e[M_EoV], EPh = e[M_EPh], EQh = e[M_EQh], En = e[M_En], Ec = e[M_Ec], Eh = e[M_Eh], Ev = e[M_Ev];
  const double x = e[M_Ex], y = e[M_Ey], pcj = k.Tprod_c * e[M_EdTc], pv = e[M_Epv];
  // n equation
  o[0] = E1 - k.Tn_c * Ec - k.Tn_h * Eh - k.Tn_v * Ev;
  o[1] = -(k.Tn_c * En + k.Th2n * EPh);
  o[2] = -(k.Tn_h * En + k.Th2n * EPh + k.Th2n * EoV);
  o[3] = k.Th2n * EQh - k.Tn_v * En;
  o[4] = 0.0;
  // c equation
  o[5] = k.Tn_c * Ec - pcj;
  o[6] = E1 - k.Tprod_c * e[M_ETau] - pcj + k.Tc2h * EoV + x + k.Tn_c * En;
  o[7] = x - pcj - k.Th2c * (E1 - EoV);
  o[8] = -(pcj + y);
  o[9] = 0.0;
  // h equation
  const double x2 = x - k.Th2n * EPh;
  o[10] = k.Tn_h * Eh;
  o[11] = -(k.Tc2h * EoV + x2);
  o[12] = E1 - x2 + k.Th2c * (E1 - EoV) + k.Tn_h * En + k.Th2n * EoV;
  o[13] = y - k.Th2n * EQh;
  o[14] = 0.0;
  // v equation
  o[15] = pv + k.Tn_v * Ev;
  o[16] = pv;
  o[17] = pv;
  o[18] = E1 + pv + k.Tn_v * En + e[M_Edd];
  o[19] = e[M_Eg4];
  // a equation
  o[20] = 0.0;
  o[21] = -(k.Tsec_c * E1);
  o[22] = -(k.Tsec_h * E1);
  o[23] = k.Tupt * e[M_Ea];
  o[24] = E1 + k.Tupt * Ev + k.Tdec * E1;
  if (NMT == NMG) {   // cell transport: rdc_tet4_ev.h, GEN
    const double Xc = e[M_Xc % NMT], Xh = e[M_Xh % NMT];
    o[5] += Xc; o[6] += Xc + e[M_Yc % NMT]; o[7] += Xc; o[8] += Xc + e[M_Zc % NMT];
    o[10] += Xh; o[11] += Xh; o[12] += Xh + e[M_Yh % NMT]; o[13] += Xh + e[M_Zh % NMT];
  }
}

}  // namespace ev
}  // namespace rdc
#endif
