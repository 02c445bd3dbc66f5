/*
 * rdc_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * CPU (FP64, scalar) restatement of the element-assembly hot path of rdcFEs, used as the
 * checker for the HIP path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this; the product (rdcfes_amd/, include/) never does.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or expected outputs
 * (SURVEY.md §4), and it cannot be built here (needs libMesh d3bda6c + PETSc 746207a, neither
 * present; no stand-in headers are written for them).  This restatement is pinned only by the
 * known-answer tests we author in tests/test_oracle_*.py (partition of unity, quadrature
 * exactness, uniform-state closed forms, finite-difference Jacobian checks, F=I limits).
 * One exception, off the assembly path proper: the eigenvalue step of SolidSystem::post_process is pinned against the
 * reference's own src/eig3.C, which compiles from its own source (oracle/ref_eig3_wrap.cpp, oracle/_ref,
 * tests/test_ref_eig3.py).
 *
 * Each function cites the reference lines it follows (paths relative to the upstream tree).
 * Third-party arithmetic that is NOT in the upstream tree (libMesh d3bda6c): FIRST LAGRANGE
 * shape functions on TET4/HEX8, QGauss(THIRD) rules, FEMap (J, JxW) — restated from the
 * published definitions, SURVEY.md App. B.
 *
 * Loop order is kept reference-faithful on purpose (elem -> qp -> i -> j, dense Ke/Fe, then a
 * per-entry sorted-row insertion like MatSetValues) so that the same code doubles as the
 * "port" CPU baseline.
 */
#include "rdc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * FE tables: libMesh FIRST LAGRANGE on TET4 / HEX8, QGauss(3, THIRD)   [SURVEY App. B.2-B.4]
 * ------------------------------------------------------------------------------------------ */

/* TET4 5-point rule with one negative weight (exact for cubics). */
static const double TET_QP[5][3] = {
  {0.25, 0.25, 0.25},
  {0.5, 1.0 / 6.0, 1.0 / 6.0},
  {1.0 / 6.0, 0.5, 1.0 / 6.0},
  {1.0 / 6.0, 1.0 / 6.0, 0.5},
  {1.0 / 6.0, 1.0 / 6.0, 1.0 / 6.0}};
static const double TET_QW[5] = {-2.0 / 15.0, 0.075, 0.075, 0.075, 0.075};

/* HEX8 reference corners (libMesh / Gmsh order). */
static const double HEX_C[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1},
                                   {-1, -1, 1},  {1, -1, 1},  {1, 1, 1},  {-1, 1, 1}};

int oracle_nqp(int elem_type) { return elem_type == 4 ? 5 : (elem_type == 8 ? 8 : -1); }

static void ref_shape(int elem_type, const double xi[3], double* N, double (*dN)[3]) {
  if (elem_type == 4) {
    N[0] = 1.0 - xi[0] - xi[1] - xi[2];
    N[1] = xi[0];
    N[2] = xi[1];
    N[3] = xi[2];
    dN[0][0] = dN[0][1] = dN[0][2] = -1.0;
    dN[1][0] = 1; dN[1][1] = 0; dN[1][2] = 0;
    dN[2][0] = 0; dN[2][1] = 1; dN[2][2] = 0;
    dN[3][0] = 0; dN[3][1] = 0; dN[3][2] = 1;
  } else {
    for (int n = 0; n < 8; n++) {
      const double a = 1.0 + HEX_C[n][0] * xi[0], b = 1.0 + HEX_C[n][1] * xi[1],
                   c = 1.0 + HEX_C[n][2] * xi[2];
      N[n] = 0.125 * a * b * c;
      dN[n][0] = 0.125 * HEX_C[n][0] * b * c;
      dN[n][1] = 0.125 * a * HEX_C[n][1] * c;
      dN[n][2] = 0.125 * a * b * HEX_C[n][2];
    }
  }
}

static void ref_qpoint(int elem_type, int q, double xi[3], double* w) {
  if (elem_type == 4) {
    xi[0] = TET_QP[q][0]; xi[1] = TET_QP[q][1]; xi[2] = TET_QP[q][2];
    *w = TET_QW[q];
  } else {
    /* tensor 2-point Gauss, x fastest */
    const double g = 0.57735026918962576451; /* 1/sqrt(3) */
    xi[0] = (q & 1) ? g : -g;
    xi[1] = (q & 2) ? g : -g;
    xi[2] = (q & 4) ? g : -g;
    *w = 1.0;
  }
}

static double det3(const double m[3][3]) {
  return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) -
         m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
         m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

static void inv3(const double m[3][3], double r[3][3]) {
  const double d = det3(m), s = 1.0 / d;
  r[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) * s;
  r[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * s;
  r[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * s;
  r[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * s;
  r[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * s;
  r[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * s;
  r[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) * s;
  r[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * s;
  r[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * s;
}

/* fe->reinit(elem): phi[q][n], dphi[q][n][3] (physical gradients), JxW[q].
 * (libMesh FEMap: J = dx/dxi, dphi = dN/dxi . J^-1, JxW = det(J) * w.)   src/pihna.C:420 */
int oracle_fe_reinit(int elem_type, const double* X /*[nen][3]*/, double* phi, double* dphi,
                     double* JxW) {
  const int nen = elem_type, nqp = oracle_nqp(elem_type);
  if (nqp < 0) return 1;
  for (int q = 0; q < nqp; q++) {
    double xi[3], w, N[8], dN[8][3], J[3][3] = {{0}}, Ji[3][3];
    ref_qpoint(elem_type, q, xi, &w);
    ref_shape(elem_type, xi, N, dN);
    for (int n = 0; n < nen; n++)
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) J[r][c] += X[3 * n + r] * dN[n][c]; /* dx_r / dxi_c */
    inv3(J, Ji);                                                         /* dxi_c / dx_r = Ji[c][r] */
    for (int n = 0; n < nen; n++) {
      phi[q * nen + n] = N[n];
      for (int r = 0; r < 3; r++)
        dphi[(q * nen + n) * 3 + r] = dN[n][0] * Ji[0][r] + dN[n][1] * Ji[1][r] + dN[n][2] * Ji[2][r];
    }
    JxW[q] = det3(J) * w;
  }
  return 0;
}

#define PHI(n) phi[q * nen + (n)]
#define DPHI(n, d) dphi[(q * nen + (n)) * 3 + (d)]
static inline double dot_dphi(const double g[3], const double* dphi, int q, int nen, int n) {
  return g[0] * DPHI(n, 0) + g[1] * DPHI(n, 1) + g[2] * DPHI(n, 2);
}
static inline double dphi_dphi(const double* dphi, int q, int nen, int a, int b) {
  return DPHI(a, 0) * DPHI(b, 0) + DPHI(a, 1) * DPHI(b, 1) + DPHI(a, 2) * DPHI(b, 2);
}

/* ------------------------------------------------------------------------------------------
 * PIHNA integrand                                                    src/pihna.C:427-750
 * u: [nen][5] old solution (n,c,h,v,a);  Ke: [5nen][5nen] var-major (src/pihna.C:408-410)
 * ------------------------------------------------------------------------------------------ */
void oracle_pihna_element(int nen, int nqp, const double* phi, const double* dphi,
                          const double* JxW, const double* u, const rdc_pihna_params* P,
                          double* Ke, double* Fe) {
  const int nd = 5 * nen;
  memset(Ke, 0, sizeof(double) * nd * nd);
  memset(Fe, 0, sizeof(double) * nd);
  /* src/pihna.C:358-381 */
  const double DT_2 = P->time_step / 2.0;
  const double Lambda_k = P->cells_min_capacity, Kappa_k = P->cells_max_capacity,
               Kappa_a = P->cytokines_max_capacity, ek = P->cells_max_capacity_exponent;
  const double nec_c = P->necrosis_c / Kappa_k, nec_h = P->necrosis_h / Kappa_k,
               nec_v = P->necrosis_v / Kappa_k;
  const double prod_c = P->produce_c, c2h = P->switch_c2h, h2c = P->switch_h2c,
               h2n = P->switch_h2n, prod_v = P->produce_v;
  const double sec_c = P->secrete_a_c, sec_h = P->secrete_a_h, upt = P->uptake_a_v,
               dec = P->decay_a;
#define KE(a, b) Ke[((a) * nen + i) * nd + (b) * nen + j]
#define FE(a) Fe[(a) * nen + i]
  for (int q = 0; q < nqp; q++) {
    /* src/pihna.C:429-442 */
    double n_ = 0, c_ = 0, h_ = 0, v_ = 0, a_ = 0;
    double Gc[3] = {0, 0, 0}, Gh[3] = {0, 0, 0}, Gv[3] = {0, 0, 0}, Ga[3] = {0, 0, 0};
    for (int l = 0; l < nen; l++) {
      n_ += PHI(l) * u[5 * l + 0];
      c_ += PHI(l) * u[5 * l + 1];
      h_ += PHI(l) * u[5 * l + 2];
      v_ += PHI(l) * u[5 * l + 3];
      a_ += PHI(l) * u[5 * l + 4];
      for (int d = 0; d < 3; d++) {
        Gc[d] += DPHI(l, d) * u[5 * l + 1];
        Gh[d] += DPHI(l, d) * u[5 * l + 2];
        Gv[d] += DPHI(l, d) * u[5 * l + 3];
        Ga[d] += DPHI(l, d) * u[5 * l + 4];
      }
    }
    /* crowding function, src/pihna.C:444-472; the four derivatives are identical */
    double Tau, dT;
    {
      const double Te = (n_ + c_ + h_ + v_) / Kappa_k;
      if (Te <= 0.0) { Tau = 1.0; dT = 0.0; }
      else if (Te >= 1.0) { Tau = 0.0; dT = 0.0; }
      else { Tau = pow(1.0 - Te, ek); dT = (-ek / Kappa_k) * pow(1.0 - Te, ek - 1.0); }
    }
    /* vascular fraction, src/pihna.C:474-499 (NaN when c+h+v == 0 falls into the else branch) */
    double Ve, Ve_dc, Ve_dh, Ve_dv;
    {
      const double Ve_ = v_ / (c_ + h_ + v_);
      if (Ve_ <= 0.0) { Ve = 0.0; Ve_dc = Ve_dh = Ve_dv = 0.0; }
      else if (Ve_ >= 1.0) { Ve = 1.0; Ve_dc = Ve_dh = Ve_dv = 0.0; }
      else {
        Ve = Ve_;
        Ve_dc = Ve_dh = -Ve_ / (c_ + h_ + v_);
        Ve_dv = (1.0 - Ve_) / (c_ + h_ + v_);
      }
    }
    /* src/pihna.C:501-509 */
    const double Ua = a_ / (a_ + Kappa_a), Ua_da = 1.0 / (a_ + Kappa_a) - Ua / (a_ + Kappa_a);
    const double dif_c = (c_ > Lambda_k ? P->diffuse_c : 0.0), tax_c = (c_ > Lambda_k ? P->taxis_c : 0.0),
                 dif_h = (h_ > Lambda_k ? P->diffuse_h : 0.0), tax_h = (h_ > Lambda_k ? P->taxis_h : 0.0),
                 dif_v = (v_ > Lambda_k ? P->diffuse_v : 0.0), tax_v = (v_ > Lambda_k ? P->taxis_v : 0.0);
    const double W = JxW[q];
    for (int i = 0; i < nen; i++) {
      const double pi = PHI(i);
      const double gc = dot_dphi(Gc, dphi, q, nen, i), gh = dot_dphi(Gh, dphi, q, nen, i),
                   gv = dot_dphi(Gv, dphi, q, nen, i), ga = dot_dphi(Ga, dphi, q, nen, i);
      /* right-hand side, src/pihna.C:514-566 */
      FE(0) += W * (n_ * pi + DT_2 * (nec_c * c_ * n_ * pi + nec_h * h_ * n_ * pi + nec_v * v_ * n_ * pi +
                                      h2n * (1.0 - Ve) * h_ * pi));
      FE(1) += W * (c_ * pi + DT_2 * (prod_c * Tau * c_ * pi - c2h * (1.0 - Ve) * c_ * pi + h2c * Ve * h_ * pi -
                                      nec_c * c_ * n_ * pi - dif_c * Tau * gc - tax_c * Tau * c_ * gv));
      FE(2) += W * (h_ * pi + DT_2 * (c2h * (1.0 - Ve) * c_ * pi - h2c * Ve * h_ * pi - nec_h * h_ * n_ * pi -
                                      dif_h * Tau * gh - tax_h * Tau * h_ * gv - h2n * (1.0 - Ve) * h_ * pi));
      FE(3) += W * (v_ * pi + DT_2 * (prod_v * Tau * Ua * v_ * pi - nec_v * v_ * n_ * pi - dif_v * Tau * gv -
                                      tax_v * Tau * v_ * ga));
      FE(4) += W * (a_ * pi + DT_2 * (sec_c * c_ * pi + sec_h * h_ * pi - upt * v_ * a_ * pi - dec * a_ * pi));
      for (int j = 0; j < nen; j++) {
        const double pj = PHI(j), pp = pj * pi, dd = dphi_dphi(dphi, q, nen, j, i);
        /* n-row, src/pihna.C:571-597 */
        KE(0, 0) += W * (pp - DT_2 * (nec_c * c_ * pp + nec_h * h_ * pp + nec_v * v_ * pp));
        KE(0, 1) += W * (-DT_2 * (nec_c * pj * n_ * pi + h2n * (-Ve_dc) * pj * h_ * pi));
        KE(0, 2) += W * (-DT_2 * (nec_h * pj * n_ * pi + h2n * (-Ve_dh) * pj * h_ * pi + h2n * (1.0 - Ve) * pp));
        KE(0, 3) += W * (-DT_2 * (nec_v * pj * n_ * pi + h2n * (-Ve_dv) * pj * h_ * pi));
        /* c-row, src/pihna.C:599-641 */
        KE(1, 0) += W * (-DT_2 * (prod_c * dT * pj * c_ * pi - nec_c * c_ * pp - dif_c * dT * pj * gc -
                                  tax_c * dT * pj * c_ * gv));
        KE(1, 1) += W * (pp - DT_2 * (prod_c * Tau * pp + prod_c * dT * pj * c_ * pi - c2h * (1.0 - Ve) * pp -
                                      c2h * (-Ve_dc) * pj * c_ * pi + h2c * Ve_dc * pj * h_ * pi -
                                      nec_c * pj * n_ * pi - dif_c * dT * pj * gc - dif_c * Tau * dd -
                                      tax_c * dT * pj * c_ * gv - tax_c * Tau * pj * gv));
        KE(1, 2) += W * (-DT_2 * (prod_c * dT * pj * c_ * pi - c2h * (-Ve_dh) * pj * c_ * pi +
                                  h2c * Ve_dh * pj * h_ * pi + h2c * Ve * pp - dif_c * dT * pj * gc -
                                  tax_c * dT * pj * c_ * gv));
        KE(1, 3) += W * (-DT_2 * (prod_c * dT * pj * c_ * pi - c2h * (-Ve_dv) * pj * c_ * pi +
                                  h2c * Ve_dv * pj * h_ * pi - dif_c * dT * pj * gc -
                                  tax_c * dT * pj * c_ * gv - tax_c * Tau * c_ * dd));
        /* h-row, src/pihna.C:643-684 */
        KE(2, 0) += W * (-DT_2 * (-nec_h * h_ * pp - dif_h * dT * pj * gh - tax_h * dT * pj * h_ * gv));
        KE(2, 1) += W * (-DT_2 * (c2h * (1.0 - Ve) * pp + c2h * (-Ve_dc) * pj * c_ * pi - h2c * Ve_dc * pj * h_ * pi -
                                  dif_h * dT * pj * gh - tax_h * dT * pj * h_ * gv -
                                  h2n * (-Ve_dc) * pj * h_ * pi));
        KE(2, 2) += W * (pp - DT_2 * (c2h * (-Ve_dh) * pj * c_ * pi - h2c * Ve_dh * pj * h_ * pi - h2c * Ve * pp -
                                      nec_h * pj * n_ * pi - dif_h * dT * pj * gh - dif_h * Tau * dd -
                                      tax_h * dT * pj * h_ * gv - tax_h * Tau * pj * gv -
                                      h2n * (-Ve_dh) * pj * h_ * pi - h2n * (1.0 - Ve) * pp));
        KE(2, 3) += W * (-DT_2 * (c2h * (-Ve_dv) * pj * c_ * pi - h2c * Ve_dv * pj * h_ * pi - dif_h * dT * pj * gh -
                                  tax_h * dT * pj * h_ * gv - tax_h * Tau * h_ * dd -
                                  h2n * (-Ve_dv) * pj * h_ * pi));
        /* v-row, src/pihna.C:686-724 */
        KE(3, 0) += W * (-DT_2 * (prod_v * dT * pj * Ua * v_ * pi - nec_v * v_ * pp - dif_v * dT * pj * gv -
                                  tax_v * dT * pj * v_ * ga));
        KE(3, 1) += W * (-DT_2 * (prod_v * dT * pj * Ua * v_ * pi - dif_v * dT * pj * gv - tax_v * dT * pj * v_ * ga));
        KE(3, 2) += W * (-DT_2 * (prod_v * dT * pj * Ua * v_ * pi - dif_v * dT * pj * gv - tax_v * dT * pj * v_ * ga));
        KE(3, 3) += W * (pp - DT_2 * (prod_v * dT * pj * Ua * v_ * pi - nec_v * pj * n_ * pi - dif_v * dT * pj * gv -
                                      dif_v * Tau * dd - tax_v * dT * pj * v_ * ga - tax_v * Tau * pj * ga));
        KE(3, 4) += W * (-DT_2 * (prod_v * Tau * Ua_da * pj * v_ * pi - tax_v * Tau * v_ * dd));
        /* a-row, src/pihna.C:726-747 */
        KE(4, 1) += W * (-DT_2 * (sec_c * pp));
        KE(4, 2) += W * (-DT_2 * (sec_h * pp));
        KE(4, 3) += W * (-DT_2 * (-upt * pj * a_ * pi));
        KE(4, 4) += W * (pp - DT_2 * (-upt * v_ * pp - dec * pp));
      }
    }
  }
#undef KE
#undef FE
}

/* ------------------------------------------------------------------------------------------
 * RIPF integrand                                                      src/ripf.C:449-665
 * u: [nen][3] (HU,cc,fb); aux: [nen][3] = {cc_dtime, fb_dtime, RT_total}
 * (TD vars 1,2 src/ripf.C:470-471; RT var 2 src/ripf.C:477-478).
 * ------------------------------------------------------------------------------------------ */
static inline double pow2(double v) { return v * v; }                          /* src/utils.h:69 */
static inline double apply_lbound(double L, double X) { return X < L ? L : X; } /* src/utils.h:86 */

void oracle_ripf_element(int nen, int nqp, const double* phi, const double* dphi,
                         const double* JxW, const double* u, const double* aux,
                         const rdc_ripf_params* P, double* Ke, double* Fe) {
  const int nd = 3 * nen;
  memset(Ke, 0, sizeof(double) * nd * nd);
  memset(Fe, 0, sizeof(double) * nd);
  const double DT_2 = P->time_step / 2.0;
  /* src/ripf.C:398-403: a zero "RT/r" falls back to the integer runtime parameter */
  const double lambda_RT_r = P->lambda_RT_r ? P->lambda_RT_r : (double)P->RT_dose_total_max;
  const double omicro_RT_r = P->omicro_RT_r ? P->omicro_RT_r : (double)P->RT_dose_total_max;
#define KE(a, b) Ke[((a) * nen + i) * nd + (b) * nen + j]
#define FE(a) Fe[(a) * nen + i]
  for (int q = 0; q < nqp; q++) {
    double HU = 0, cc = 0, fb = 0, cc_dt = 0, fb_dt = 0, RT = 0;
    double GHU[3] = {0, 0, 0}, Gfb[3] = {0, 0, 0}, GRT[3] = {0, 0, 0};
    for (int l = 0; l < nen; l++) { /* src/ripf.C:451-479 */
      HU += PHI(l) * u[3 * l + 0];
      cc += PHI(l) * u[3 * l + 1];
      fb += PHI(l) * u[3 * l + 2];
      cc_dt += PHI(l) * aux[3 * l + 0];
      fb_dt += PHI(l) * aux[3 * l + 1];
      RT += PHI(l) * aux[3 * l + 2];
      for (int d = 0; d < 3; d++) {
        GHU[d] += DPHI(l, d) * u[3 * l + 0];
        Gfb[d] += DPHI(l, d) * u[3 * l + 2];
        GRT[d] += DPHI(l, d) * aux[3 * l + 2];
      }
    }
    { /* unit radiotherapy gradient, src/ripf.C:481-484 */
      const double l2 = sqrt(GRT[0] * GRT[0] + GRT[1] * GRT[1] + GRT[2] * GRT[2]);
      if (l2) { GRT[0] /= l2; GRT[1] /= l2; GRT[2] /= l2; }
      else { GRT[0] = GRT[1] = GRT[2] = 0.0; }
    }
    /* src/ripf.C:486-496 */
    const double kappa_RT = P->kappa * exp(-P->kappa_RT_c * RT);
    const double delta_RT = P->delta * (1.0 - exp(-P->delta_RT_a * RT - P->delta_RT_b * pow2(RT)));
    const double lambda_RT = P->lambda * (RT / lambda_RT_r);
    const double omicro_RT = P->omicro * apply_lbound(0.0, 4.0 * ((RT / omicro_RT_r) - pow2(RT / omicro_RT_r)));
    double eps_cc = 0.0, eps_fb = 0.0;
    if (cc_dt > P->phi_tol) eps_cc = P->phi_cc_B; else if (cc_dt < -P->phi_tol) eps_cc = P->phi_cc_D;
    if (fb_dt > P->phi_tol) eps_fb = P->phi_fb_B; else if (fb_dt < -P->phi_tol) eps_fb = P->phi_fb_D;
    /* src/ripf.C:498-514 */
    const double VF_total = P->VolFr_stroma + P->VolFr_parenchyma + (cc + fb);
    double Tau = 0.0, dTau = 0.0; /* Tau__dcc == Tau__dfb */
    if (VF_total < 1.0) {
      Tau = pow(1.0 - VF_total, P->VolFr_exponent);
      dTau = -P->VolFr_exponent * pow(1.0 - VF_total, P->VolFr_exponent - 1.0);
      if (Tau < P->VolFr_min_vacant) { Tau = 0.0; dTau = 0.0; }
    }
    /* src/ripf.C:516-523 */
    double Koppa = 0.0, Koppa_dcc = 0.0;
    if (cc < 0.0) { }
    else if (cc < 1.0) { Koppa = 4.0 * (cc - cc * cc); Koppa_dcc = 4.0 - 8.0 * cc; }
    /* src/ripf.C:525-561 (the d/dcc and Omecro d/dHU derivatives are identically zero) */
    double Lom = 0.0, Lom_dHU = 0.0, Lom_dcc = 0.0, Lom_dfb = 0.0;
    double Ome = 0.0, Ome_dHU = 0.0, Ome_dcc = 0.0, Ome_dfb = 0.0;
    if (fb < 0.0) { }
    else if (fb < 1.0) {
      if (HU > P->lambda_HU_r && HU < 0.0) {
        Lom = (1.0 - pow2(fb)) * (HU / P->lambda_HU_r);
        Lom_dHU = (1.0 - pow2(fb)) / P->lambda_HU_r;
        Lom_dfb = -(2.0 * fb) * (HU / P->lambda_HU_r);
      } else if (HU < P->lambda_HU_r) {
        Lom = (1.0 - pow2(fb));
        Lom_dfb = -(2.0 * fb);
      }
      if (fb <= P->omicro_fb_b) {
        Ome = 4.0 * (P->omicro_fb_b - pow2(P->omicro_fb_b));
      } else {
        Ome = 4.0 * (fb - pow2(fb));
        Ome_dfb = 4.0 - 8.0 * fb;
      }
    }
    const double W = JxW[q];
    for (int i = 0; i < nen; i++) {
      const double pi = PHI(i);
      const double gfb = dot_dphi(Gfb, dphi, q, nen, i), gHU = dot_dphi(GHU, dphi, q, nen, i),
                   gRT = dot_dphi(GRT, dphi, q, nen, i);
      /* src/ripf.C:566-594; (GRAD_HU * fb * dphi) == fb * (GRAD_HU . dphi) */
      FE(0) += W * (HU * pi + DT_2 * (eps_cc * cc * pi + eps_fb * fb * pi + P->phi_cc * cc_dt * pi + P->phi_fb * fb_dt * pi));
      FE(1) += W * (cc * pi + DT_2 * (kappa_RT * Tau * Koppa * pi - delta_RT * cc * pi));
      FE(2) += W * (fb * pi + DT_2 * (lambda_RT * Tau * Lom * pi + omicro_RT * Tau * Ome * pi - P->omega * fb * pi -
                                      P->diffusion * Tau * gfb - P->haptotaxis * Tau * (fb * gHU) -
                                      P->radiotaxis * Tau * (fb * gRT)));
      for (int j = 0; j < nen; j++) {
        const double pj = PHI(j), pp = pj * pi, dd = dphi_dphi(dphi, q, nen, j, i);
        /* src/ripf.C:599-662 */
        KE(0, 0) += W * (pp);
        KE(0, 1) += W * (-DT_2 * (eps_cc * pp));
        KE(0, 2) += W * (-DT_2 * (eps_fb * pp));
        KE(1, 1) += W * (pp - DT_2 * (kappa_RT * dTau * Koppa * pp + kappa_RT * Tau * Koppa_dcc * pp - delta_RT * pp));
        KE(1, 2) += W * (-DT_2 * (kappa_RT * dTau * Koppa * pp));
        KE(2, 0) += W * (-DT_2 * (lambda_RT * Tau * Lom_dHU * pp + omicro_RT * Tau * Ome_dHU * pp -
                                  P->haptotaxis * Tau * (fb * dd)));
        KE(2, 1) += W * (-DT_2 * (lambda_RT * dTau * Lom * pp + lambda_RT * Tau * Lom_dcc * pp +
                                  omicro_RT * dTau * Ome * pp + omicro_RT * Tau * Ome_dcc * pp -
                                  P->diffusion * dTau * pj * gfb - P->haptotaxis * dTau * pj * (fb * gHU) -
                                  P->radiotaxis * dTau * pj * (fb * gRT)));
        KE(2, 2) += W * (pp - DT_2 * (lambda_RT * dTau * Lom * pp + lambda_RT * Tau * Lom_dfb * pp +
                                      omicro_RT * dTau * Ome * pp + omicro_RT * Tau * Ome_dfb * pp -
                                      P->omega * pp - P->diffusion * dTau * pj * gfb - P->diffusion * Tau * dd -
                                      P->haptotaxis * dTau * pj * (fb * gHU) - P->haptotaxis * Tau * (pj * gHU) -
                                      P->radiotaxis * dTau * pj * (fb * gRT) - P->radiotaxis * Tau * (pj * gRT)));
      }
    }
  }
#undef KE
#undef FE
}

/* ------------------------------------------------------------------------------------------
 * HCC integrand                                                  src/coupled_hcc.C:496-640
 * u: [nen][3] (l,c,n).  Reference quirks reproduced verbatim:
 *  - "capacity" pp term also in blocks [0][1], [0][2], [1][0]          (:577-598)
 *  - the d/dn block of the c-equation is added into [1][1] a second time, [1][2] never (:611-619)
 *  - GRAD_sigma is identically zero                                     (:508)
 * ------------------------------------------------------------------------------------------ */
void oracle_hcc_element(int nen, int nqp, const double* phi, const double* dphi,
                        const double* JxW, const double* u, const rdc_hcc_params* P, double* Ke,
                        double* Fe) {
  const int nd = 3 * nen;
  memset(Ke, 0, sizeof(double) * nd * nd);
  memset(Fe, 0, sizeof(double) * nd);
  const double DT_2 = P->time_step / 2.0;
  const double Lambda_k = P->cells_min_capacity, Kappa_k = P->cells_max_capacity,
               ek = P->cells_max_capacity_exponent;
  const double prod_l = P->produce_l, prod_c = P->produce_c;
  const double nec_l = P->necrosis_l / Kappa_k, nec_c = P->necrosis_c / Kappa_k;
#define KE(a, b) Ke[((a) * nen + i) * nd + (b) * nen + j]
#define FE(a) Fe[(a) * nen + i]
  for (int q = 0; q < nqp; q++) {
    double l_ = 0, c_ = 0, n_ = 0, Gc[3] = {0, 0, 0};
    for (int k = 0; k < nen; k++) { /* :498-506 */
      l_ += PHI(k) * u[3 * k + 0];
      c_ += PHI(k) * u[3 * k + 1];
      n_ += PHI(k) * u[3 * k + 2];
      for (int d = 0; d < 3; d++) Gc[d] += DPHI(k, d) * u[3 * k + 1];
    }
    const double Gs[3] = {0.0, 0.0, 0.0}; /* :508 */
    double Tau, dT;                       /* :510-532 */
    {
      const double Te = (l_ + c_ + n_) / Kappa_k;
      if (Te <= 0.0) { Tau = 1.0; dT = 0.0; }
      else if (Te >= 1.0) { Tau = 0.0; dT = 0.0; }
      else { Tau = pow(1.0 - Te, ek); dT = (-ek / Kappa_k) * pow(1.0 - Te, ek - 1.0); }
    }
    const double dif_c = (c_ > Lambda_k ? P->diffuse_c : 0.0), mec_c = (c_ > Lambda_k ? P->mechano_c : 0.0);
    const double W = JxW[q];
    for (int i = 0; i < nen; i++) {
      const double pi = PHI(i), gc = dot_dphi(Gc, dphi, q, nen, i), gs = dot_dphi(Gs, dphi, q, nen, i);
      /* :540-564 */
      FE(0) += W * (l_ * pi + DT_2 * (prod_l * Tau * l_ * pi - nec_l * l_ * n_ * pi));
      FE(1) += W * (c_ * pi + DT_2 * (prod_c * Tau * c_ * pi - nec_c * c_ * n_ * pi - dif_c * Tau * gc -
                                      mec_c * Tau * c_ * gs));
      FE(2) += W * (n_ * pi + DT_2 * (nec_l * l_ * n_ * pi + nec_c * c_ * n_ * pi));
      for (int j = 0; j < nen; j++) {
        const double pj = PHI(j), pp = pj * pi, dd = dphi_dphi(dphi, q, nen, j, i);
        /* :569-637 */
        KE(0, 0) += W * (pp - DT_2 * (prod_l * Tau * pp + prod_l * dT * pj * l_ * pi - nec_l * pj * n_ * pi));
        KE(0, 1) += W * (pp - DT_2 * (prod_l * dT * pj * l_ * pi));
        KE(0, 2) += W * (pp - DT_2 * (prod_l * dT * pj * l_ * pi - nec_l * l_ * pp));
        KE(1, 0) += W * (pp - DT_2 * (prod_c * dT * pj * c_ * pi - dif_c * dT * pj * gc - mec_c * dT * pj * c_ * gs));
        KE(1, 1) += W * (pp - DT_2 * (prod_c * Tau * pp + prod_c * dT * pj * c_ * pi - nec_c * pj * n_ * pi -
                                      dif_c * dT * pj * gc - dif_c * Tau * dd - mec_c * dT * pj * c_ * gs -
                                      mec_c * Tau * pj * gs));
        KE(1, 1) += W * (pp - DT_2 * (prod_c * dT * pj * c_ * pi - nec_c * c_ * pp - dif_c * dT * pj * gc -
                                      mec_c * dT * pj * c_ * gs));
        KE(2, 0) += W * (-DT_2 * (nec_l * pj * n_ * pi));
        KE(2, 1) += W * (-DT_2 * (nec_c * pj * n_ * pi));
        KE(2, 2) += W * (pp - DT_2 * (nec_l * l_ * pp + nec_c * c_ * pp));
      }
    }
  }
#undef KE
#undef FE
}

/* ------------------------------------------------------------------------------------------
 * Hyperelastic constitutive law      src/hyperelastic.h:25-87, src/hyperlastic_inline.h:3-189
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  double F[3][3], Fe[3][3], Fp[3][3], A[3];
  double sigma[3][3];
  double C[6][6];
} hyper_state;

static void mat3_mul(const double a[3][3], const double b[3][3], double r[3][3]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}

/* src/hyperlastic_inline.h:18-189 */
static void hyper_calculate_stress(hyper_state* S, double Young, double Poisson, double FibreStiffness,
                                   int tangent) {
  const double mu = 0.5 * Young / (1.0 + Poisson);
  const double lambda = Young * Poisson / ((1.0 + Poisson) * (1.0 - 2.0 * Poisson));
  const double koppa = FibreStiffness / 2.0;
  double FeT[3][3], Ce[3][3], CeINV[3][3], FpINV[3][3], CeCe[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) FeT[i][j] = S->Fe[j][i];
  mat3_mul(FeT, S->Fe, Ce);
  inv3(Ce, CeINV);
  inv3(S->Fp, FpINV);
  static const double delta[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  const double I1 = Ce[0][0] + Ce[1][1] + Ce[2][2];
  mat3_mul(Ce, Ce, CeCe);
  const double I2 = 0.5 * (pow2(I1) - (CeCe[0][0] + CeCe[1][1] + CeCe[2][2]));
  (void)I2;
  const double Je = det3(S->Fe);
  const double J_recip = 1.0 / det3(S->F);
  const double dWdI1 = (mu / 2.0), dWdI2 = 0.0;
  const double dWdJe = (-mu / Je) + (lambda / 2.0 * Je - lambda / 2.0 / Je);
  const double dWdI4 = (-koppa);
  const double d2WdI1dI1 = 0.0, d2WdI2dI2 = 0.0, d2WdI4dI4 = 0.0;
  const double d2WdJedJe = (mu / Je / Je) + (lambda / 2.0 + lambda / 2.0 / Je / Je);
  double dI1[3][3], dI2[3][3], dJe[3][3], dI4[3][3];
  double d2I2[3][3][3][3], d2Je[3][3][3][3]; /* on the stack: oracle_assemble_mt calls this from several threads */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      dI1[i][j] = delta[i][j];
      dI2[i][j] = delta[i][j] * I1 - Ce[i][j];
      dJe[i][j] = 0.5 * Je * CeINV[i][j];
      dI4[i][j] = S->A[i] * S->A[j];
      for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) {
          d2I2[i][j][k][l] = delta[i][j] * delta[k][l] - 0.5 * delta[i][k] * delta[j][l] - 0.5 * delta[i][l] * delta[j][k];
          d2Je[i][j][k][l] = 0.25 * Je * CeINV[i][j] * CeINV[k][l] - 0.25 * Je * CeINV[i][k] * CeINV[j][l] -
                             0.25 * Je * CeINV[i][l] * CeINV[j][k];
        }
    }
  double S2pk[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      S2pk[i][j] = 2.0 * dWdI1 * dI1[i][j] + 2.0 * dWdI2 * dI2[i][j] + 2.0 * dWdJe * dJe[i][j] + 2.0 * dWdI4 * dI4[i][j];
  /* push-forward with the TOTAL F and 1/det F (App. D.6) */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
      for (int I = 0; I < 3; I++)
        for (int J = 0; J < 3; J++) s += S->F[i][I] * S->F[j][J] * S2pk[I][J];
      S->sigma[i][j] = s * J_recip;
    }
  if (!tangent) return;
  double dSdCe[3][3][3][3], dCedC[3][3][3][3], dSdC[3][3][3][3], tsm[3][3][3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) {
          dSdCe[i][j][k][l] = 4.0 * dWdI2 * d2I2[i][j][k][l] + 4.0 * dWdJe * d2Je[i][j][k][l] +
                              4.0 * d2WdI1dI1 * dI1[i][j] * dI1[k][l] + 4.0 * d2WdI2dI2 * dI2[i][j] * dI2[k][l] +
                              4.0 * d2WdJedJe * dJe[i][j] * dJe[k][l] + 4.0 * d2WdI4dI4 * dI4[i][j] * dI4[k][l];
          dCedC[i][j][k][l] = 0.5 * FpINV[k][i] * FpINV[j][l] + 0.5 * FpINV[l][i] * FpINV[k][j];
        }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) {
          double s = 0.0;
          for (int m = 0; m < 3; m++)
            for (int n = 0; n < 3; n++) s += dSdCe[i][j][m][n] * dCedC[m][n][k][l];
          dSdC[i][j][k][l] = s;
        }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) {
          double s = 0.0;
          for (int I = 0; I < 3; I++)
            for (int J = 0; J < 3; J++)
              for (int K = 0; K < 3; K++)
                for (int L = 0; L < 3; L++)
                  s += S->F[i][I] * S->F[j][J] * S->F[k][K] * S->F[l][L] * dSdC[I][J][K][L];
          tsm[i][j][k][l] = s * J_recip;
        }
  /* Voigt order (00,11,22,01,12,02), src/hyperelastic.h:15-20, src/hyperlastic_inline.h:153-188 */
  static const int V[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {1, 2}, {0, 2}};
  for (int p = 0; p < 6; p++)
    for (int q = 0; q < 6; q++) S->C[p][q] = tsm[V[p][0]][V[p][1]][V[q][0]][V[q][1]];
}

/* src/hyperelastic.h:25-49.  gradX[d][c] = d X_d / d x_c */
static void hyper_initialize(hyper_state* S, const double gradX[3][3], const double lam[3],
                             const double f[3], double Young, double Poisson, double K, int tangent) {
  inv3(gradX, S->F);
  memset(S->Fp, 0, sizeof(S->Fp));
  for (int l = 0; l < 3; l++) S->Fp[l][l] = lam[l];
  double FpINV[3][3];
  inv3(S->Fp, FpINV);
  mat3_mul(S->F, FpINV, S->Fe);
  if (K > 0.0) {
    const double nrm = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (int d = 0; d < 3; d++) S->A[d] = f[d] / nrm;
  } else {
    S->A[0] = S->A[1] = S->A[2] = 0.0;
  }
  hyper_calculate_stress(S, Young, Poisson, K, tangent);
}

/* src/hyperlastic_inline.h:3-15 */
static void hyper_B(const double g[3], double B[3][6]) {
  memset(B, 0, sizeof(double) * 18);
  B[0][0] = g[0]; B[1][1] = g[1]; B[2][2] = g[2];
  B[0][3] = g[1]; B[1][3] = g[0];
  B[1][4] = g[2]; B[2][4] = g[1];
  B[0][5] = g[2]; B[2][5] = g[0];
}

/* test hook: sigma and Voigt tangent for given F-inverse (gradX), growth stretches, fibre */
void oracle_hyperelastic_point(const double* gradX9, const double* lambda3, const double* fibre3,
                               double Young, double Poisson, double K, double* sigma9, double* C36) {
  hyper_state S;
  double g[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) g[i][j] = gradX9[3 * i + j];
  hyper_initialize(&S, g, lambda3, fibre3, Young, Poisson, K, 1);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) sigma9[3 * i + j] = S.sigma[i][j];
  for (int p = 0; p < 6; p++) for (int q = 0; q < 6; q++) C36[6 * p + q] = S.C[p][q];
}

/* SolidSystem::element_time_derivative                          src/solid_system.C:146-271
 * phi/dphi/JxW evaluated on the CURRENT element; Xu: undeformed coords [nen][3];
 * Re: [3 nen] var-major; Je: [3nen][3nen] var-major blocks (ii,jj)(i,j). */
void oracle_solid_element(int nen, int nqp, const double* dphi, const double* JxW, const double* Xu,
                          const double* fibre3, const rdc_solid_material* M, double pseudo_time,
                          int request_jacobian, int use_symmetry, double* Je, double* Re) {
  const int nd = 3 * nen;
  memset(Je, 0, sizeof(double) * nd * nd);
  memset(Re, 0, sizeof(double) * nd);
  hyper_state S;
  for (int q = 0; q < nqp; q++) {
    double gradX[3][3] = {{0}}; /* :221-229 */
    for (int d = 0; d < 3; d++)
      for (int l = 0; l < nen; l++)
        for (int c = 0; c < 3; c++) gradX[d][c] += DPHI(l, c) * Xu[3 * l + d];
    double lam[3]; /* :232-234 */
    for (int d = 0; d < 3; d++) lam[d] = 1.0 + pseudo_time * M->rate[d];
    hyper_initialize(&S, gradX, lam, fibre3, M->Young, M->Poisson, M->FibreStiffness, request_jacobian);
    const double SV[6] = {S.sigma[0][0], S.sigma[1][1], S.sigma[2][2], S.sigma[0][1], S.sigma[1][2], S.sigma[0][2]};
    for (int i = 0; i < nen; i++) {
      double Bi[3][6], gi[3] = {DPHI(i, 0), DPHI(i, 1), DPHI(i, 2)};
      hyper_B(gi, Bi);
      for (int ii = 0; ii < 3; ii++) { /* get_residual, src/hyperelastic.h:52-66 */
        double r = 0.0;
        for (int p = 0; p < 6; p++) r += Bi[ii][p] * SV[p];
        Re[ii * nen + i] += r * JxW[q];
      }
      if (!request_jacobian) continue;
      for (int j = (use_symmetry ? i : 0); j < nen; j++) { /* :252-265 */
        double Bj[3][6], gj[3] = {DPHI(j, 0), DPHI(j, 1), DPHI(j, 2)}, D[3][3] = {{0}};
        hyper_B(gj, Bj);
        /* get_linearized_stiffness, src/hyperelastic.h:68-87 */
        double G = 0.0;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) G += gi[r] * S.sigma[r][c] * gj[c];
        for (int n = 0; n < 3; n++) D[n][n] += G;
        double BC[3][6];
        for (int r = 0; r < 3; r++)
          for (int p = 0; p < 6; p++) {
            double s = 0.0;
            for (int t = 0; t < 6; t++) s += Bi[r][t] * S.C[t][p];
            BC[r][p] = s;
          }
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) {
            double s = 0.0;
            for (int p = 0; p < 6; p++) s += BC[r][p] * Bj[c][p];
            D[r][c] += s;
          }
        for (int ii = 0; ii < 3; ii++)
          for (int jj = 0; jj < 3; jj++) {
            Je[(ii * nen + i) * nd + jj * nen + j] += D[ii][jj] * JxW[q];
            if (use_symmetry && i != j) Je[(ii * nen + j) * nd + jj * nen + i] += D[jj][ii] * JxW[q];
          }
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * ADPM                                                              src/adpm.C:324-652
 * piecewise rate functions                                          src/utils.h:100-187
 * ------------------------------------------------------------------------------------------ */
static double Pi_(double C, const double* p_) {
  const double cM = p_[0];
  if (0.0 >= cM) return 0.0;
  const double c0 = p_[1], c1 = p_[2];
  if (C < c0) return 0.0;
  else if (C < c1) return cM;
  else return 0.0;
}
static double SD_(double C, const double* p_) {
  const double cM = p_[0];
  if (0.0 >= cM) return 0.0;
  const double c0 = p_[1], c1 = p_[2];
  if (C < c0) return cM;
  else if (C < c1) return cM * (c1 - C) / (c1 - c0);
  else return 0.0;
}
static double deriv_SD_(double C, const double* p_) {
  const double cM = p_[0];
  if (0.0 >= cM) return 0.0;
  const double c0 = p_[1], c1 = p_[2];
  if (C < c0) return 0.0;
  else if (C < c1) return -cM / (c1 - c0);
  else return 0.0;
}
static double Tr_(double C, const double* p_) {
  const double cM = p_[0];
  if (0.0 >= cM) return 0.0;
  const double c0 = p_[1], c1 = p_[2], c2 = p_[3], c3 = p_[4];
  if (C < c0) return 0.0;
  else if (C < c1) return cM * (C - c0) / (c1 - c0);
  else if (C < c2) return cM;
  else if (C < c3) return cM * (c3 - C) / (c3 - c2);
  else return 0.0;
}
static double deriv_Tr_(double C, const double* p_) {
  const double cM = p_[0];
  if (0.0 >= cM) return 0.0;
  const double c0 = p_[1], c1 = p_[2], c2 = p_[3], c3 = p_[4];
  if (C < c0) return 0.0;
  else if (C < c1) return cM / (c1 - c0);
  else if (C < c2) return 0.0;
  else if (C < c3) return -cM / (c3 - c2);
  else return 0.0;
}

/* u: [nen][3] = (PrP, A_b, Tau); tracts3: the element's tract vector ("Tracts" system, :448-453).
 * The boundary-penalty block of upstream is dead code (`if ( 0 )`, :601) and is not restated. */
#define KE(a, b) Ke[((a) * nen + i) * nd + (b) * nen + j]
#define FE(a) Fe[(a) * nen + i]
void oracle_adpm_element(int nen, int nqp, const double* phi, const double* dphi, const double* JxW,
                         const double* u, const double* tracts3, const rdc_adpm_params* P, double* Ke,
                         double* Fe) {
  const int nd = 3 * nen;
  memset(Ke, 0, sizeof(double) * nd * nd);
  memset(Fe, 0, sizeof(double) * nd);
  const double DT_2 = P->time_step / 2.0;                                        /* :365 */
  const double decay_PrP[3] = {P->decay_PrP[0] * pow(P->time, P->decay_PrP_time_exponent), P->decay_PrP[1],
                               P->decay_PrP[2]};                                 /* :369-372 */
  const double* diffuse_A_b = P->diffuse_A_b; const double* taxis1_A_b = P->taxis1_A_b;
  const double* taxis2_A_b = P->taxis2_A_b;   const double* produce_A_b = P->produce_A_b;
  const double* transform_A_b = P->transform_A_b; const double* decay_A_b = P->decay_A_b;
  const double* diffuse_Tau = P->diffuse_Tau; const double* taxis1_Tau = P->taxis1_Tau;
  const double* taxis2_Tau = P->taxis2_Tau;   const double* produce_Tau = P->produce_Tau;
  const double* transform_Tau = P->transform_Tau; const double* decay_Tau = P->decay_Tau;
  const double omega_A_b = cos(P->taxis_A_b_angle), omega_Tau = cos(P->taxis_Tau_angle); /* :412-413 */
  for (int q = 0; q < nqp; q++) {
    double PrP_old = 0, A_b_old = 0, Tau_old = 0, GA[3] = {0, 0, 0}, GT[3] = {0, 0, 0};
    for (int l = 0; l < nen; l++) { /* :462-471 */
      PrP_old += PHI(l) * u[3 * l + 0];
      A_b_old += PHI(l) * u[3 * l + 1];
      Tau_old += PHI(l) * u[3 * l + 2];
      for (int d = 0; d < 3; d++) { GA[d] += DPHI(l, d) * u[3 * l + 1]; GT[d] += DPHI(l, d) * u[3 * l + 2]; }
    }
    const double nA = sqrt(GA[0] * GA[0] + GA[1] * GA[1] + GA[2] * GA[2]);
    const double nT = sqrt(GT[0] * GT[0] + GT[1] * GT[1] + GT[2] * GT[2]);       /* :473 */
    double tract_A_b[3] = {0, 0, 0}, tract_Tau[3] = {0, 0, 0};
    if (nA) {                                                                    /* :477-484 */
      const double d = (GA[0] / nA) * tracts3[0] + (GA[1] / nA) * tracts3[1] + (GA[2] / nA) * tracts3[2];
      if (d > +omega_A_b) for (int x = 0; x < 3; x++) tract_A_b[x] = tracts3[x];
      else if (d < -omega_A_b) for (int x = 0; x < 3; x++) tract_A_b[x] = -tracts3[x];
    }
    if (nT) {                                                                    /* :485-493 */
      const double d = (GT[0] / nT) * tracts3[0] + (GT[1] / nT) * tracts3[1] + (GT[2] / nT) * tracts3[2];
      if (d > +omega_Tau) for (int x = 0; x < 3; x++) tract_Tau[x] = tracts3[x];
      else if (d < -omega_Tau) for (int x = 0; x < 3; x++) tract_Tau[x] = -tracts3[x];
    }
    const double W = JxW[q];
    for (int i = 0; i < nen; i++) {
      const double pi = PHI(i);
      const double gA = dot_dphi(GA, dphi, q, nen, i), gT = dot_dphi(GT, dphi, q, nen, i);
      const double tA = dot_dphi(tract_A_b, dphi, q, nen, i), tT = dot_dphi(tract_Tau, dphi, q, nen, i);
      FE(0) += W * (PrP_old * pi                                                  /* :497-504 */
                    + DT_2 * (-Tr_(A_b_old, transform_A_b) * PrP_old * pi - Tr_(Tau_old, transform_Tau) * PrP_old * pi -
                              Pi_(PrP_old, decay_PrP) * PrP_old * pi));
      FE(1) += W * (A_b_old * pi                                                  /* :506-518 */
                    + DT_2 * (SD_(A_b_old, produce_A_b) * A_b_old * pi + Tr_(A_b_old, transform_A_b) * PrP_old * pi -
                              Pi_(A_b_old, decay_A_b) * A_b_old * pi - Pi_(A_b_old, diffuse_A_b) * gA -
                              Pi_(A_b_old, taxis1_A_b) * A_b_old * tA + Pi_(Tau_old, taxis2_A_b) * A_b_old * tT));
      FE(2) += W * (Tau_old * pi                                                  /* :520-532 */
                    + DT_2 * (SD_(Tau_old, produce_Tau) * Tau_old * pi + Tr_(Tau_old, transform_Tau) * PrP_old * pi -
                              Pi_(Tau_old, decay_Tau) * Tau_old * pi - Pi_(Tau_old, diffuse_Tau) * gT -
                              Pi_(Tau_old, taxis1_Tau) * Tau_old * tT + Pi_(A_b_old, taxis2_Tau) * Tau_old * tA));
      for (int j = 0; j < nen; j++) {
        const double pj = PHI(j), pp = pj * pi, dd = dphi_dphi(dphi, q, nen, j, i);
        KE(0, 0) += W * (pp - DT_2 * (-Tr_(A_b_old, transform_A_b) * pp - Tr_(Tau_old, transform_Tau) * pp -
                                      Pi_(PrP_old, decay_PrP) * pp));             /* :537-545 */
        KE(0, 1) += W * (-DT_2 * (-deriv_Tr_(A_b_old, transform_A_b) * PrP_old * pp)); /* :546-550 */
        KE(0, 2) += W * (-DT_2 * (-deriv_Tr_(Tau_old, transform_Tau) * PrP_old * pp)); /* :551-555 */
        KE(1, 0) += W * (-DT_2 * (Tr_(A_b_old, transform_A_b) * pp));             /* :557-561 */
        KE(1, 1) += W * (pp - DT_2 * (SD_(A_b_old, produce_A_b) * pp + deriv_SD_(A_b_old, produce_A_b) * A_b_old * pp +
                                      deriv_Tr_(A_b_old, transform_A_b) * PrP_old * pp - Pi_(A_b_old, decay_A_b) * pp -
                                      Pi_(A_b_old, diffuse_A_b) * dd - Pi_(A_b_old, taxis1_A_b) * pj * tA +
                                      Pi_(Tau_old, taxis2_A_b) * pj * tT));       /* :562-575 */
        KE(2, 0) += W * (-DT_2 * (Tr_(Tau_old, transform_Tau) * pp));             /* :577-581 */
        KE(2, 2) += W * (pp - DT_2 * (SD_(Tau_old, produce_Tau) * pp + deriv_SD_(Tau_old, produce_Tau) * Tau_old * pp +
                                      deriv_Tr_(Tau_old, transform_Tau) * PrP_old * pp - Pi_(Tau_old, decay_Tau) * pp -
                                      Pi_(Tau_old, diffuse_Tau) * dd - Pi_(Tau_old, taxis1_Tau) * pj * tT +
                                      Pi_(A_b_old, taxis2_Tau) * pj * tA));       /* :582-595 */
      }
    }
  }
}
#undef KE
#undef FE

/* ------------------------------------------------------------------------------------------
 * PROTEAS                                                           src/proteas.C:338-705
 * u: [nen][5] = (hos, tum, nec, vsc, oed); aux0: [nen] nodal values of AUX variable 0 ("HU").
 * As upstream, RTD at a point is phi_AUX[1][qp] * AUX(dof_indices_AUX_var[0][1]) = phi_1 * aux0[1]   (:481);
 * HU, GRAD_HU, GRAD_RTD are computed upstream but never used in the integrands.
 * ------------------------------------------------------------------------------------------ */
static double heaviside_(double x) { return (x > 0 ? 1 : 0); } /* src/utils.h:84 */
#define KE(a, b) Ke[((a) * nen + i) * nd + (b) * nen + j]
#define FE(a) Fe[(a) * nen + i]
void oracle_proteas_element(int nen, int nqp, const double* phi, const double* dphi, const double* JxW,
                            const double* u, const double* aux0, const rdc_proteas_params* P, double* Ke,
                            double* Fe) {
  const int nd = 5 * nen;
  memset(Ke, 0, sizeof(double) * nd * nd);
  memset(Fe, 0, sizeof(double) * nd);
  const double DT_2 = P->time_step / 2.0, T_max = P->cells_total_capacity, RT_max = P->RT_max_dosage;
  const double rho_h = P->host_proliferation, u_h = P->host_vsc_threshold, delta_h = P->host_RT_death_rate,
               a_RT_h = P->host_RT_exp_a, b_RT_h = P->host_RT_exp_b, nu_h = P->host_necrosis_rate;
  const double D_c = P->tumour_diffusion, D_c_h = P->tumour_diffusion_host, rho_c = P->tumour_proliferation,
               u_c = P->tumour_vsc_threshold, delta_c = P->tumour_RT_death_rate, a_RT_c = P->tumour_RT_exp_a,
               b_RT_c = P->tumour_RT_exp_b, nu_c = P->tumour_necrosis_rate;
  const double psi_n = P->necrosis_clearance, k_n = P->necrosis_slope, u_n = P->necrosis_vsc_threshold;
  const double rho_v = P->vascular_proliferation, nu_v = P->vascular_necrosis_rate;
  const double D_e = P->oedema_diffusion, rho_e = P->oedema_proliferation, u_e = P->oedema_vsc_threshold,
               xi_e = P->oedema_RT_coeff, p_RT_e = P->oedema_RT_exp, psi_e = P->oedema_reabsorption_rate;
  for (int q = 0; q < nqp; q++) {
    double hos_old = 0, tum_old = 0, nec_old = 0, vsc_old = 0, oed_old = 0;
    double Gh[3] = {0, 0, 0}, Gt[3] = {0, 0, 0}, Go[3] = {0, 0, 0};
    for (int l = 0; l < nen; l++) { /* :459-469 */
      hos_old += PHI(l) * u[5 * l + 0];
      tum_old += PHI(l) * u[5 * l + 1];
      nec_old += PHI(l) * u[5 * l + 2];
      vsc_old += PHI(l) * u[5 * l + 3];
      oed_old += PHI(l) * u[5 * l + 4];
      for (int d = 0; d < 3; d++) {
        Gh[d] += DPHI(l, d) * u[5 * l + 0];
        Gt[d] += DPHI(l, d) * u[5 * l + 1];
        Go[d] += DPHI(l, d) * u[5 * l + 4];
      }
    }
    const double RTD = PHI(1) * aux0[1]; /* :481 */
    const double T = hos_old + tum_old + nec_old + vsc_old;
    double Kappa = 1.0 - T / T_max;
    Kappa = fmin(fmax(Kappa, 0.0), 1.0); /* :492 */
    const double dKappa = -1.0 / T_max;
    const double host_prol = rho_h * Kappa * heaviside_(vsc_old - u_h);
    const double dhost_prol = rho_h * dKappa * heaviside_(vsc_old - u_h);
    const double host_RT_death = delta_h * (1.0 - exp(-a_RT_h * RTD - b_RT_h * (RTD * RTD)));
    const double host_nec = nu_h * nec_old;
    const double tumour_prol = rho_c * Kappa * heaviside_(vsc_old - u_c);
    const double dtumour_prol = rho_c * dKappa * heaviside_(vsc_old - u_c);
    const double tumour_RT_death = delta_c * (1.0 - exp(-a_RT_c * RTD - b_RT_c * (RTD * RTD)));
    const double tumour_nec = nu_c * nec_old;
    const double nec_prol = nu_h * hos_old + nu_c * tum_old + nu_v * vsc_old;
    const double nec_clearance = psi_n * (1.0 - tanh(k_n * vsc_old - u_n));
    const double dnec_clearance_dv = psi_n * -k_n / (cosh(k_n * vsc_old - u_n) * cosh(k_n * vsc_old - u_n));
    const double vsc_prol = rho_v * Kappa * tum_old;
    const double dvsc_prol = rho_v * dKappa * tum_old;
    const double vsc_nec = nu_v * nec_old;
    const double oed_prol = rho_e * tum_old * (1.0 - tum_old);
    const double doed_prol_dc = rho_e * (1.0 - 2.0 * tum_old);
    const double oed_RT = xi_e * pow(RTD / RT_max, p_RT_e);
    const double oed_clearance = psi_e * (1.0 - heaviside_(vsc_old - u_e));
    const double W = JxW[q];
    for (int i = 0; i < nen; i++) {
      const double pi = PHI(i);
      const double gt = dot_dphi(Gt, dphi, q, nen, i), go = dot_dphi(Go, dphi, q, nen, i);
      const double Ght[3] = {Gh[0] * tum_old, Gh[1] * tum_old, Gh[2] * tum_old}; /* (GRAD_hos_old * tum_old) * dphi */
      const double ght = dot_dphi(Ght, dphi, q, nen, i);
      FE(0) += W * (hos_old * pi + DT_2 * (host_prol * hos_old * (1.0 - hos_old) * pi - host_RT_death * hos_old * pi -
                                           host_nec * hos_old * pi));                       /* :522-529 */
      FE(1) += W * (tum_old * pi + DT_2 * (-D_c * Kappa * gt - D_c_h * Kappa * ght + tumour_prol * tum_old * pi -
                                           tumour_RT_death * tum_old * pi - tumour_nec * tum_old * pi)); /* :531-541 */
      FE(2) += W * (nec_old * pi + DT_2 * (nec_prol * nec_old * pi - nec_clearance * nec_old * pi)); /* :543-550 */
      FE(3) += W * (vsc_old * pi + DT_2 * (vsc_prol * vsc_old * pi - vsc_nec * vsc_old * pi));       /* :552-559 */
      FE(4) += W * (oed_old * pi + DT_2 * (-D_e * go + oed_prol * oed_old * pi - oed_RT * oed_old * pi -
                                           oed_clearance * oed_old * pi));                  /* :561-570 */
      for (int j = 0; j < nen; j++) {
        const double pj = PHI(j), pp = pj * pi;
        const double dd = dphi_dphi(dphi, q, nen, j, i);
        const double ddt = (DPHI(j, 0) * tum_old) * DPHI(i, 0) + (DPHI(j, 1) * tum_old) * DPHI(i, 1) +
                           (DPHI(j, 2) * tum_old) * DPHI(i, 2);                             /* dphi[j]*tum_old*dphi[i] */
        const double hh = hos_old * (1.0 - hos_old);
        KE(0, 0) += W * (pp - DT_2 * (dhost_prol * hh * pp + host_prol * (1.0 - 2.0 * hos_old) * pp - host_RT_death * pp -
                                      host_nec * pp));                                      /* :577-585 */
        KE(0, 1) += W * (-DT_2 * (dhost_prol * hh * pp));                                   /* :586-590 */
        KE(0, 2) += W * (-DT_2 * (dhost_prol * hh * pp - nu_h * pj * hos_old * pi));        /* :591-596 */
        KE(0, 3) += W * (-DT_2 * (dhost_prol * hh * pp));                                   /* :597-601 */
        KE(1, 0) += W * (-DT_2 * (-D_c * dKappa * pj * gt - D_c_h * dKappa * pj * ght - D_c_h * Kappa * ddt +
                                  dtumour_prol * pj * tum_old * pi));                       /* :603-610 */
        KE(1, 1) += W * (pp - DT_2 * (-D_c * dKappa * pj * gt - D_c * Kappa * dd + dtumour_prol * pj * tum_old * pi +
                                      tumour_prol * pp - tumour_RT_death * pp - tumour_nec * pp)); /* :611-621 */
        KE(1, 2) += W * (-DT_2 * (-D_c * dKappa * pj * gt - D_c_h * dKappa * pj * ght + dtumour_prol * pj * tum_old * pi -
                                  nu_c * pj * tum_old * pi));                               /* :622-629 */
        KE(1, 3) += W * (-DT_2 * (-D_c * dKappa * pj * gt - D_c_h * dKappa * pj * ght + dtumour_prol * pj * tum_old * pi)); /* :630-636 */
        KE(2, 0) += W * (-DT_2 * (nu_h * pj * nec_old * pi));                               /* :638-642 */
        KE(2, 1) += W * (-DT_2 * (nu_c * pj * nec_old * pi));                               /* :643-647 */
        KE(2, 2) += W * (pp - DT_2 * (nec_prol * pp - nec_clearance * pp));                 /* :648-654 */
        KE(2, 3) += W * (-DT_2 * (nu_v * pj * nec_old * pi - dnec_clearance_dv * pj * nec_old * pi)); /* :655-660 */
        KE(3, 0) += W * (-DT_2 * (dvsc_prol * pj * vsc_old * pi));                          /* :662-666 */
        KE(3, 1) += W * (-DT_2 * (dvsc_prol * pj * vsc_old * pi));                          /* :667-671 */
        KE(3, 2) += W * (-DT_2 * (dvsc_prol * pj * vsc_old * pi - nu_v * pj * vsc_old * pi)); /* :672-677 */
        KE(3, 3) += W * (pp - DT_2 * (dvsc_prol * pj * vsc_old * pi + vsc_prol * pp - vsc_nec * pp)); /* :678-685 */
        KE(4, 1) += W * (-DT_2 * (doed_prol_dc * pj * oed_old * pi));                       /* :687-691 */
        KE(4, 4) += W * (pp - DT_2 * (-D_e * dd + oed_prol * pp - oed_RT * pp - oed_clearance * pp)); /* :692-700 */
      }
    }
  }
}
#undef KE
#undef FE

/* Eigenvalues of a symmetric 3x3 matrix by cyclic Jacobi rotations.  Upstream calls eigen_decomposition()
 * (src/eig3.C:261-271, Householder tridiagonalisation + QL); only the eigenVALUES are consumed
 * (src/solid_system.C:519-524), and both algorithms deliver them to a few ulp of |A|. */
static void sym3_eigenvalues(const double Ain[3][3], double ev[3]) {
  double a[3][3];
  memcpy(a, Ain, sizeof(a));
  for (int sweep = 0; sweep < 60; sweep++) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    if (off == 0.0) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        const int r = 3 - p - q;
        const double app = a[p][p], aqq = a[q][q], apq = a[p][q], arp = a[r][p], arq = a[r][q];
        a[p][p] = app - t * apq;
        a[q][q] = aqq + t * apq;
        a[p][q] = a[q][p] = 0.0;
        a[r][p] = a[p][r] = c * arp - sn * arq;
        a[r][q] = a[q][r] = sn * arp + c * arq;
      }
  }
  ev[0] = a[0][0]; ev[1] = a[1][1]; ev[2] = a[2][2];
}

/* the eigenvalue step and the two stress measures of post_process, exported so that the tests can pin them against the
 * reference's own src/eig3.C (oracle/_ref/libref_eig3.so): A9 = symmetric 3x3 row-major; ev[3] unordered */
void oracle_stress_measures(const double* A9, double* ev, double* pressure, double* von_mises) {
  double A[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] = A9[3 * i + j];
  sym3_eigenvalues(A, ev);
  *pressure = (ev[0] + ev[1] + ev[2]) / 3.0;                                                                                    /* src/solid_system.C:517 */
  *von_mises = sqrt(ev[0] * ev[0] + ev[1] * ev[1] + ev[2] * ev[2] - ev[0] * ev[1] - ev[0] * ev[2] - ev[1] * ev[2]);  /* :519-520 */
}

/* SolidSystem::post_process                                       src/solid_system.C:394-538
 * per element: plain average over the quadrature points of the Cauchy stress and of F*eta (:500-516,526),
 * hydrostatic pressure and von Mises stress from the principal stresses (:518-524).
 * pressure, von_mises: [n_elem]; fibre_current: [n_elem][3]. */
int oracle_solid_post_process(int elem_type, int64_t n_elem, const uint32_t* conn, const double* xyz,
                              const double* xyz_undeformed, const double* elem_fibre, const int32_t* elem_material,
                              const rdc_solid_material* materials, double pseudo_time, double* pressure,
                              double* von_mises, double* fibre_current) {
  const int nen = elem_type, nqp = oracle_nqp(elem_type);
  if (nqp < 0) return 1;
  double X[8 * 3], XU[8 * 3], phi[8 * 8], dphi[8 * 8 * 3], JxW[8];
  hyper_state S;
  for (int64_t e = 0; e < n_elem; e++) {
    const uint32_t* c = conn + e * nen;
    for (int i = 0; i < nen; i++)
      for (int d = 0; d < 3; d++) {
        X[3 * i + d] = xyz[3 * (int64_t)c[i] + d];
        XU[3 * i + d] = xyz_undeformed[3 * (int64_t)c[i] + d];
      }
    oracle_fe_reinit(elem_type, X, phi, dphi, JxW);
    const rdc_solid_material* M = &materials[elem_material[e]];
    const double* eta = elem_fibre + 3 * e;
    double sc[3][3] = {{0}}, fv[3] = {0, 0, 0};
    for (int q = 0; q < nqp; q++) {
      double gradX[3][3] = {{0}}; /* :490-498 */
      for (int d = 0; d < 3; d++)
        for (int l = 0; l < nen; l++)
          for (int k = 0; k < 3; k++) gradX[d][k] += DPHI(l, k) * XU[3 * l + d];
      double lam[3];
      for (int d = 0; d < 3; d++) lam[d] = 1.0 + pseudo_time * M->rate[d]; /* :501-503 */
      hyper_initialize(&S, gradX, lam, eta, M->Young, M->Poisson, M->FibreStiffness, 0);
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) sc[i][j] += S.sigma[i][j];                                  /* :509 */
        fv[i] += S.F[i][0] * eta[0] + S.F[i][1] * eta[1] + S.F[i][2] * eta[2];                   /* :511 */
      }
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) sc[i][j] /= nqp;                                               /* :515 */
    const double Sc[3][3] = {{sc[0][0], sc[0][1], sc[0][2]}, {sc[0][1], sc[1][1], sc[1][2]}, {sc[0][2], sc[1][2], sc[2][2]}};
    double ev[3];
    sym3_eigenvalues(Sc, ev);
    pressure[e] = (ev[0] + ev[1] + ev[2]) / 3.0;                                                 /* :522 */
    von_mises[e] = sqrt(ev[0] * ev[0] + ev[1] * ev[1] + ev[2] * ev[2] - ev[0] * ev[1] - ev[0] * ev[2] - ev[1] * ev[2]); /* :524 */
    for (int d = 0; d < 3; d++) fibre_current[3 * e + d] = fv[d] / nqp;                          /* :526 */
  }
  return 0;
}

/* side tables (libMesh Tet4::side_nodes_map / Hex8::side_nodes_map) */
static const int TET_SIDE[4][3] = {{0, 2, 1}, {0, 1, 3}, {1, 2, 3}, {2, 0, 3}};
static const int HEX_SIDE[6][4] = {{0, 3, 2, 1}, {0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {3, 0, 4, 7}, {4, 5, 6, 7}};

/* SolidSystem::side_time_derivative                              src/solid_system.C:273-371
 * x: current coords [nen][3], Xu: undeformed; disp: prescribed displacement (NaN = free);
 * ADDS into Je/Re.  Side rule: QGauss(2, THIRD) on TRI3 (4 pts) / QUAD4 (2x2) [App. B.2]. */
void oracle_solid_side(int nen, int side, const double* x, const double* Xu, const double* disp3,
                       double pseudo_time, double penalty, int request_jacobian, double* Je, double* Re) {
  const int nd = 3 * nen;
  const double ratio = pseudo_time * 1.000001; /* :291-292 */
  const int nsn = (nen == 4) ? 3 : 4;
  const int* sn = (nen == 4) ? TET_SIDE[side] : HEX_SIDE[side];
  const int nq = 4;
  double qxi[4][2], qw[4];
  if (nen == 4) {
    static const double P[4][2] = {{1.0 / 3.0, 1.0 / 3.0}, {0.2, 0.6}, {0.2, 0.2}, {0.6, 0.2}};
    static const double Wt[4] = {-27.0 / 96.0, 25.0 / 96.0, 25.0 / 96.0, 25.0 / 96.0};
    for (int q = 0; q < 4; q++) { qxi[q][0] = P[q][0]; qxi[q][1] = P[q][1]; qw[q] = Wt[q]; }
  } else {
    const double g = 0.57735026918962576451;
    for (int q = 0; q < 4; q++) { qxi[q][0] = (q & 1) ? g : -g; qxi[q][1] = (q & 2) ? g : -g; qw[q] = 1.0; }
  }
  for (int q = 0; q < nq; q++) {
    double N[4], dN[4][2];
    if (nsn == 3) {
      N[0] = 1.0 - qxi[q][0] - qxi[q][1]; N[1] = qxi[q][0]; N[2] = qxi[q][1];
      dN[0][0] = -1; dN[0][1] = -1; dN[1][0] = 1; dN[1][1] = 0; dN[2][0] = 0; dN[2][1] = 1;
    } else {
      static const double Cq[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
      for (int n = 0; n < 4; n++) {
        N[n] = 0.25 * (1 + Cq[n][0] * qxi[q][0]) * (1 + Cq[n][1] * qxi[q][1]);
        dN[n][0] = 0.25 * Cq[n][0] * (1 + Cq[n][1] * qxi[q][1]);
        dN[n][1] = 0.25 * (1 + Cq[n][0] * qxi[q][0]) * Cq[n][1];
      }
    }
    double t1[3] = {0, 0, 0}, t2[3] = {0, 0, 0}, cur[3] = {0, 0, 0}, org[3] = {0, 0, 0};
    for (int n = 0; n < nsn; n++)
      for (int d = 0; d < 3; d++) {
        t1[d] += x[3 * sn[n] + d] * dN[n][0];
        t2[d] += x[3 * sn[n] + d] * dN[n][1];
        cur[d] += x[3 * sn[n] + d] * N[n];   /* coords[qp], :338 */
        org[d] += Xu[3 * sn[n] + d] * N[n];  /* orig_point, :326-334 */
      }
    const double cx = t1[1] * t2[2] - t1[2] * t2[1], cy = t1[2] * t2[0] - t1[0] * t2[2], cz = t1[0] * t2[1] - t1[1] * t2[0];
    const double W = sqrt(cx * cx + cy * cy + cz * cz) * qw[q];
    double diff[3];
    for (int d = 0; d < 3; d++) diff[d] = cur[d] - org[d] - disp3[d] * ratio; /* :337-339 */
    for (int a = 0; a < nsn; a++) {
      const int i = sn[a];
      for (int di = 0; di < 3; di++) {
        if (isnan(diff[di])) continue;
        Re[di * nen + i] += W * N[a] * diff[di] * penalty; /* :346-351 */
      }
      if (!request_jacobian) continue;
      for (int b = 0; b < nsn; b++) {
        const int j = sn[b];
        for (int dj = 0; dj < 3; dj++) {
          if (isnan(diff[dj])) continue;
          Je[(dj * nen + i) * nd + dj * nen + j] += W * N[a] * N[b] * penalty; /* :358-362 */
        }
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * Sparsity pattern (what es.init() builds) and global scatter (add_matrix / add_vector)
 * ------------------------------------------------------------------------------------------ */
static int cmp_i64(const void* a, const void* b) {
  const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

/* Node-graph pattern for the owned rows.  Two-call protocol: with bcol == NULL only bptr and the
 * return value (number of node blocks) are produced. */
int64_t oracle_build_node_pattern(int nen, int64_t n_elem, int64_t n_node, int64_t n_owned,
                                  const uint32_t* conn, int64_t* bptr /*[n_owned+1]*/, int32_t* bcol) {
  int64_t* cnt = (int64_t*)calloc((size_t)n_node + 1, sizeof(int64_t));
  for (int64_t e = 0; e < n_elem; e++)
    for (int i = 0; i < nen; i++) cnt[conn[e * nen + i] + 1]++;
  for (int64_t n = 0; n < n_node; n++) cnt[n + 1] += cnt[n];
  int64_t* inc = (int64_t*)malloc(sizeof(int64_t) * (size_t)cnt[n_node]);
  int64_t* fill = (int64_t*)calloc((size_t)n_node, sizeof(int64_t));
  for (int64_t e = 0; e < n_elem; e++)
    for (int i = 0; i < nen; i++) {
      const int64_t n = conn[e * nen + i];
      inc[cnt[n] + fill[n]++] = e;
    }
  int64_t total = 0, cap = 64;
  int64_t* tmp = (int64_t*)malloc(sizeof(int64_t) * (size_t)cap);
  bptr[0] = 0;
  for (int64_t n = 0; n < n_owned; n++) {
    const int64_t ne = cnt[n + 1] - cnt[n];
    if (ne * nen + 1 > cap) { cap = 2 * (ne * nen + 1); tmp = (int64_t*)realloc(tmp, sizeof(int64_t) * (size_t)cap); }
    int64_t m = 0;
    tmp[m++] = n; /* diagonal always present */
    for (int64_t k = 0; k < ne; k++)
      for (int i = 0; i < nen; i++) tmp[m++] = conn[inc[cnt[n] + k] * nen + i];
    qsort(tmp, (size_t)m, sizeof(int64_t), cmp_i64);
    int64_t u = 0;
    for (int64_t k = 0; k < m; k++)
      if (k == 0 || tmp[k] != tmp[k - 1]) {
        if (bcol) bcol[total + u] = (int32_t)tmp[k];
        u++;
      }
    total += u;
    bptr[n + 1] = total;
  }
  free(tmp); free(fill); free(inc); free(cnt);
  return total;
}

/* scalar CSR (AIJ) from the node pattern: row = node*nvar+a, cols = nodecol*nvar+b ascending */
void oracle_expand_pattern(int nvar, int64_t n_owned, const int64_t* bptr, const int32_t* bcol,
                           int64_t* row_ptr, int32_t* col_idx) {
  int64_t p = 0;
  row_ptr[0] = 0;
  for (int64_t n = 0; n < n_owned; n++)
    for (int a = 0; a < nvar; a++) {
      for (int64_t k = bptr[n]; k < bptr[n + 1]; k++)
        for (int b = 0; b < nvar; b++) col_idx[p++] = bcol[k] * nvar + b;
      row_ptr[n * nvar + a + 1] = p;
    }
}

/* MatSetValues(ADD_VALUES)-like insertion: binary search of the column in the sorted row */
static inline void csr_add(const int64_t* row_ptr, const int32_t* col_idx, double* val, int64_t row,
                           int32_t col, double v) {
  int64_t lo = row_ptr[row], hi = row_ptr[row + 1] - 1;
  while (lo <= hi) {
    const int64_t mid = (lo + hi) >> 1;
    const int32_t c = col_idx[mid];
    if (c == col) { val[mid] += v; return; }
    if (c < col) lo = mid + 1; else hi = mid - 1;
  }
  abort(); /* pattern violation: must never happen */
}

/* Whole-mesh assembly = the reference callback.  model: 0 PIHNA, 1 RIPF, 2 HCC, 3 SOLID.
 * Elements [e_begin, e_end) only (lets the CPU baseline time a bounded sample). */
/* The element loop over [e_begin, e_end) restricted to the rows of nodes [node_lo, node_hi): elements without a
 * node in that range are skipped, the others are evaluated completely and only those rows are inserted.  With
 * [0, n_owned) this is the reference loop.  A row receives its contributions in ascending element order whatever
 * the range is, so splitting the rows over threads (oracle_assemble_mt) reproduces the serial result bit for bit. */
static int assemble_rows(int model, int elem_type, int64_t e_begin, int64_t e_end, int64_t node_lo, int64_t node_hi,
                         const uint32_t* conn, const double* xyz, int nvar, const double* u_old,
                         const double* aux_nodal, const double* xyz_undeformed, const double* elem_fibre,
                         const int32_t* elem_material, const rdc_solid_material* materials,
                         const void* params, int request_jacobian, const int64_t* row_ptr,
                         const int32_t* col_idx, double* val, double* rhs) {
  const int nen = elem_type, nqp = oracle_nqp(elem_type);
  if (nqp < 0) return 1;
  const int nd = nvar * nen;
  double X[8 * 3], U[8 * 5], A[8 * 3], XU[8 * 3], phi[8 * 8], dphi[8 * 8 * 3], JxW[8];
  double* Ke = (double*)malloc(sizeof(double) * nd * nd);
  double* Fe = (double*)malloc(sizeof(double) * nd);
  for (int64_t e = e_begin; e < e_end; e++) {
    const uint32_t* c = conn + e * nen;
    {
      int any = 0;
      for (int i = 0; i < nen; i++) any |= ((int64_t)c[i] >= node_lo && (int64_t)c[i] < node_hi);
      if (!any) continue;
    }
    for (int i = 0; i < nen; i++) {
      for (int d = 0; d < 3; d++) X[3 * i + d] = xyz[3 * (int64_t)c[i] + d];
      if (u_old) for (int a = 0; a < nvar; a++) U[nvar * i + a] = u_old[(int64_t)c[i] * nvar + a];
      if (aux_nodal) for (int a = 0; a < 3; a++) A[3 * i + a] = aux_nodal[(int64_t)c[i] * 3 + a];
      if (xyz_undeformed) for (int d = 0; d < 3; d++) XU[3 * i + d] = xyz_undeformed[3 * (int64_t)c[i] + d];
    }
    oracle_fe_reinit(elem_type, X, phi, dphi, JxW);
    switch (model) {
      case 0: oracle_pihna_element(nen, nqp, phi, dphi, JxW, U, (const rdc_pihna_params*)params, Ke, Fe); break;
      case 1: oracle_ripf_element(nen, nqp, phi, dphi, JxW, U, A, (const rdc_ripf_params*)params, Ke, Fe); break;
      case 2: oracle_hcc_element(nen, nqp, phi, dphi, JxW, U, (const rdc_hcc_params*)params, Ke, Fe); break;
      case 5: {
        double a0[8];
        for (int i = 0; i < nen; i++) a0[i] = A[3 * i];
        oracle_proteas_element(nen, nqp, phi, dphi, JxW, U, a0, (const rdc_proteas_params*)params, Ke, Fe);
      } break;
      case 4: oracle_adpm_element(nen, nqp, phi, dphi, JxW, U, elem_fibre + 3 * e, (const rdc_adpm_params*)params, Ke, Fe); break;
      case 3: {
        const rdc_solid_params* sp = (const rdc_solid_params*)params;
        oracle_solid_element(nen, nqp, dphi, JxW, XU, elem_fibre + 3 * e, &materials[elem_material[e]],
                             sp->pseudo_time, request_jacobian, sp->use_symmetry, Ke, Fe);
      } break;
      default: free(Ke); free(Fe); return 2;
    }
    /* add_matrix / add_vector, src/pihna.C:754-755 (constraints are the identity here) */
    for (int a = 0; a < nvar; a++)
      for (int i = 0; i < nen; i++) {
        if ((int64_t)c[i] < node_lo || (int64_t)c[i] >= node_hi) continue; /* row owned elsewhere */
        const int64_t row = (int64_t)c[i] * nvar + a;
        rhs[row] += Fe[a * nen + i];
        if (model == 3 && !request_jacobian) continue;
        for (int b = 0; b < nvar; b++)
          for (int j = 0; j < nen; j++)
            csr_add(row_ptr, col_idx, val, row, (int32_t)(c[j] * nvar + b), Ke[(a * nen + i) * nd + b * nen + j]);
      }
  }
  free(Ke); free(Fe);
  return 0;
}

int oracle_assemble(int model, int elem_type, int64_t e_begin, int64_t e_end, int64_t n_owned,
                    const uint32_t* conn, const double* xyz, int nvar, const double* u_old,
                    const double* aux_nodal, const double* xyz_undeformed, const double* elem_fibre,
                    const int32_t* elem_material, const rdc_solid_material* materials,
                    const void* params, int request_jacobian, const int64_t* row_ptr,
                    const int32_t* col_idx, double* val, double* rhs) {
  return assemble_rows(model, elem_type, e_begin, e_end, 0, n_owned, conn, xyz, nvar, u_old, aux_nodal, xyz_undeformed,
                       elem_fibre, elem_material, materials, params, request_jacobian, row_ptr, col_idx, val, rhs);
}

/* The same on n_threads host cores: thread t owns the rows of a contiguous node range (ranges balanced by CSR
 * values) and runs the element loop for them -- the stand-in for the reference's `mpiexec -n P` (run/PIHNA/Makefile:8),
 * where every rank loops over its own partition.  Elements straddling two ranges are evaluated by both threads
 * (the reference exchanges such rows through PETSc's stash instead).  Bitwise equal to oracle_assemble. */
int oracle_assemble_mt(int n_threads, int model, int elem_type, int64_t e_begin, int64_t e_end, int64_t n_owned,
                       const uint32_t* conn, const double* xyz, int nvar, const double* u_old,
                       const double* aux_nodal, const double* xyz_undeformed, const double* elem_fibre,
                       const int32_t* elem_material, const rdc_solid_material* materials,
                       const void* params, int request_jacobian, const int64_t* row_ptr,
                       const int32_t* col_idx, double* val, double* rhs) {
  if (n_threads < 1) n_threads = 1;
  int64_t* cut = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_threads + 1));
  /* rows the element range can touch: [lo, hi) */
  int64_t lo = n_owned, hi = 0;
  for (int64_t x = e_begin * elem_type; x < e_end * elem_type; x++) {
    const int64_t n = conn[x];
    if (n >= n_owned) continue;
    if (n < lo) lo = n;
    if (n + 1 > hi) hi = n + 1;
  }
  if (lo > hi) lo = hi;
  const int64_t v0 = row_ptr[lo * nvar], nnz = row_ptr[hi * nvar] - v0;
  cut[0] = 0;
  {
    int64_t n = lo;
    for (int t = 1; t < n_threads; t++) {
      const int64_t target = v0 + nnz / n_threads * t;
      while (n < hi && row_ptr[n * nvar] < target) n++;
      cut[t] = n;
    }
  }
  cut[n_threads] = n_owned;
  int rc_all = 0;
#pragma omp parallel for num_threads(n_threads) schedule(static, 1)
  for (int t = 0; t < n_threads; t++) {
    const int rc = assemble_rows(model, elem_type, e_begin, e_end, cut[t], cut[t + 1], conn, xyz, nvar, u_old, aux_nodal,
                                 xyz_undeformed, elem_fibre, elem_material, materials, params, request_jacobian, row_ptr,
                                 col_idx, val, rhs);
    if (rc) {
#pragma omp atomic write
      rc_all = rc;
    }
  }
  free(cut);
  return rc_all;
}

/* boundary sides of the solid system, added after the element loop (FEMSystem::assembly order) */
int oracle_assemble_solid_sides(int elem_type, int64_t n_sides, const int64_t* side_elem,
                                const int32_t* side_id, const double* side_disp, int64_t n_owned,
                                const uint32_t* conn, const double* xyz, const double* xyz_undeformed,
                                const rdc_solid_params* sp, int request_jacobian, const int64_t* row_ptr,
                                const int32_t* col_idx, double* val, double* rhs) {
  const int nen = elem_type, nd = 3 * nen;
  double X[24], XU[24];
  double* Ke = (double*)malloc(sizeof(double) * nd * nd);
  double* Fe = (double*)malloc(sizeof(double) * nd);
  for (int64_t s = 0; s < n_sides; s++) {
    const int64_t e = side_elem[s];
    const uint32_t* c = conn + e * nen;
    for (int i = 0; i < nen; i++)
      for (int d = 0; d < 3; d++) {
        X[3 * i + d] = xyz[3 * (int64_t)c[i] + d];
        XU[3 * i + d] = xyz_undeformed[3 * (int64_t)c[i] + d];
      }
    memset(Ke, 0, sizeof(double) * nd * nd);
    memset(Fe, 0, sizeof(double) * nd);
    oracle_solid_side(nen, side_id[s], X, XU, side_disp + 3 * s, sp->pseudo_time, sp->displacement_penalty,
                      request_jacobian, Ke, Fe);
    for (int a = 0; a < 3; a++)
      for (int i = 0; i < nen; i++) {
        if ((int64_t)c[i] >= n_owned) continue;
        const int64_t row = (int64_t)c[i] * 3 + a;
        rhs[row] += Fe[a * nen + i];
        if (!request_jacobian) continue;
        for (int b = 0; b < 3; b++)
          for (int j = 0; j < nen; j++) {
            const double v = Ke[(a * nen + i) * nd + b * nen + j];
            if (v != 0.0) csr_add(row_ptr, col_idx, val, row, (int32_t)(c[j] * 3 + b), v);
          }
      }
  }
  free(Ke); free(Fe);
  return 0;
}

/* RIPF check_solution                                             src/ripf.C:675-775
 * sol/prev/td/rt: [n][3]; aux: [n][3] = {cc rate, fb rate, RT total} as assemble_ripf reads them (:470-478).
 * Returns RT_total_max (:705,761). */
double oracle_ripf_check_solution(int64_t n, const rdc_ripf_check_params* p, double* sol, double* prev, double* td,
                                  double* rt, double* aux) {
  const double DT_R = 1.0 / p->time_step;                               /* :695 */
  const double RT_broad_frac = p->RT_broad_fractions, RT_focus_frac = p->RT_focus_fractions,
               RT_total_frac = RT_broad_frac + RT_focus_frac;           /* :699-701 */
  const int day = p->day;                                               /* :703 */
  double RT_total_max = -1.0;                                           /* :705 */
  for (int64_t i = 0; i < n; i++) {
    const double soln[3] = {sol[3 * i], sol[3 * i + 1], sol[3 * i + 2]};
    double HU_ = soln[0], cc_ = soln[1], fb_ = soln[2];
    if (HU_ < p->HU_min) HU_ = p->HU_min; else if (HU_ > p->HU_max) HU_ = p->HU_max; /* :722 */
    if (cc_ < 0.0) cc_ = 0.0;
    if (fb_ < 0.0) fb_ = 0.0;
    sol[3 * i] = HU_; sol[3 * i + 1] = cc_; sol[3 * i + 2] = fb_;       /* :731-733 */
    td[3 * i] = (HU_ - prev[3 * i]) * DT_R;                              /* :738-740 */
    td[3 * i + 1] = (cc_ - prev[3 * i + 1]) * DT_R;
    td[3 * i + 2] = (fb_ - prev[3 * i + 2]) * DT_R;
    const double RT_broad_ = rt[3 * i], RT_focus_ = rt[3 * i + 1];
    double RT_total_ = 0.0;
    if (day < RT_broad_frac) RT_total_ = RT_broad_ / RT_broad_frac * (day + 1);                                   /* :755 */
    else if (day < RT_total_frac) RT_total_ = RT_focus_ / RT_focus_frac * ((day + 1) - RT_broad_frac) + RT_broad_; /* :756 */
    else RT_total_ = RT_broad_ + RT_focus_;
    rt[3 * i + 2] = RT_total_;
    if (RT_total_ > RT_total_max) RT_total_max = RT_total_;              /* :761 */
    prev[3 * i] = soln[0]; prev[3 * i + 1] = soln[1]; prev[3 * i + 2] = soln[2]; /* prev_soln = soln (unclamped), :769 */
    aux[3 * i] = td[3 * i + 1]; aux[3 * i + 1] = td[3 * i + 2]; aux[3 * i + 2] = RT_total_;
  }
  return RT_total_max;
}

/* save_solution volume sums of PIHNA                                src/pihna.C:898-958
 * u: [n_node][5]; elements [0, n_elem); out[4] = active tumour, necrotic, vascularity, total cell volumes.
 * elem->volume() is restated as the sum of JxW of the element's rule. */
int oracle_pihna_volume_integrals(int elem_type, int64_t n_elem, const uint32_t* conn, const double* xyz, const double* u,
                                  const rdc_pihna_ranges* r, double* out) {
  const int nen = elem_type, nqp = oracle_nqp(elem_type);
  if (nqp < 0) return 1;
  double X[8 * 3], phi[8 * 8], dphi[8 * 8 * 3], JxW[8];
  out[0] = out[1] = out[2] = out[3] = 0.0;
  for (int64_t e = 0; e < n_elem; e++) {
    const uint32_t* c = conn + e * nen;
    for (int i = 0; i < nen; i++)
      for (int d = 0; d < 3; d++) X[3 * i + d] = xyz[3 * (int64_t)c[i] + d];
    oracle_fe_reinit(elem_type, X, phi, dphi, JxW);
    double Volume = 0.0;
    for (int q = 0; q < nqp; q++) Volume += JxW[q];
    int consider;
    consider = 1; /* :905-917 */
    for (int n = 0; n < nen; n++) {
      const double c_h_ = u[5 * (int64_t)c[n] + 1] + u[5 * (int64_t)c[n] + 2];
      if (!(c_h_ >= r->active_tumor_min && c_h_ <= r->active_tumor_max)) { consider = 0; break; }
    }
    if (consider) out[0] += Volume;
    consider = 1; /* :919-930 */
    for (int n = 0; n < nen; n++) {
      const double n_ = u[5 * (int64_t)c[n]];
      if (!(n_ >= r->necrotic_min && n_ <= r->necrotic_max)) { consider = 0; break; }
    }
    if (consider) out[1] += Volume;
    consider = 1; /* :932-943 */
    for (int n = 0; n < nen; n++) {
      const double v_ = u[5 * (int64_t)c[n] + 3];
      if (!(v_ >= r->vascularity_min && v_ <= r->vascularity_max)) { consider = 0; break; }
    }
    if (consider) out[2] += Volume;
    consider = 1; /* :945-958 */
    for (int n = 0; n < nen; n++) {
      const double* un = u + 5 * (int64_t)c[n];
      const double T_ = (un[0] + un[1] + un[2] + un[3]) / r->cells_max_capacity;
      if (!(T_ >= r->total_cell_min && T_ <= r->total_cell_max)) { consider = 0; break; }
    }
    if (consider) out[3] += Volume;
  }
  return 0;
}

/* RIPF save_solution, the element loop of src/ripf.C:812-858: out = {tumour_volume, fibrosis_volume} */
int oracle_ripf_volume_integrals(int elem_type, int64_t n_elem, const uint32_t* conn, const double* xyz, const double* u,
                                 const rdc_ripf_ranges* r, double* out) {
  const int nen = elem_type, nqp = oracle_nqp(elem_type);
  if (nqp < 0) return 1;
  double X[8 * 3], phi[8 * 8], dphi[8 * 8 * 3], JxW[8];
  out[0] = out[1] = 0.0;
  for (int64_t e = 0; e < n_elem; e++) {
    const uint32_t* c = conn + e * nen;
    for (int i = 0; i < nen; i++)
      for (int d = 0; d < 3; d++) X[3 * i + d] = xyz[3 * (int64_t)c[i] + d];
    oracle_fe_reinit(elem_type, X, phi, dphi, JxW);
    double Volume = 0.0; /* elem->volume() */
    for (int q = 0; q < nqp; q++) Volume += JxW[q];
    int do_include = 1; /* :832-842 */
    for (int l = 0; l < nen && do_include; l++) {
      const double HU = u[3 * (int64_t)c[l]], cc = u[3 * (int64_t)c[l] + 1];
      if (!(HU >= r->cc_HU_min && HU <= r->cc_HU_max && cc >= r->cc_min)) do_include = 0;
    }
    if (do_include) out[0] += Volume;
    do_include = 1; /* :844-854 */
    for (int l = 0; l < nen && do_include; l++) {
      const double HU = u[3 * (int64_t)c[l]], fb = u[3 * (int64_t)c[l] + 2];
      if (!(HU >= r->fb_HU_min && HU <= r->fb_HU_max && fb >= r->fb_min)) do_include = 0;
    }
    if (do_include) out[1] += Volume;
  }
  return 0;
}

/* ADPM save_solution, the element loop of src/adpm.C:747-813.  out[n_ids][4] = {A_b concentration, Tau
 * concentration, A_b volume, Tau volume} per parcellation id.  The concentrations are ASSIGNED per element upstream
 * (:780-783), so the value of the last element of the region survives; regions without elements keep 0 (a
 * default-constructed std::map entry). */
int oracle_adpm_parcellation_integrals(int elem_type, int64_t n_elem, const uint32_t* conn, const double* xyz,
                                       const double* u, const rdc_adpm_ranges* r, const int32_t* elem_subdomain,
                                       const int32_t* ids, int32_t n_ids, double* out) {
  const int nen = elem_type, nqp = oracle_nqp(elem_type);
  if (nqp < 0) return 1;
  double X[8 * 3], phi[8 * 8], dphi[8 * 8 * 3], JxW[8];
  for (int32_t i = 0; i < 4 * n_ids; i++) out[i] = 0.0;
  for (int64_t e = 0; e < n_elem; e++) {
    int32_t slot = -1;
    for (int32_t i = 0; i < n_ids; i++) if (ids[i] == elem_subdomain[e]) { slot = i; break; }
    if (slot < 0) continue;
    const uint32_t* c = conn + e * nen;
    for (int i = 0; i < nen; i++)
      for (int d = 0; d < 3; d++) X[3 * i + d] = xyz[3 * (int64_t)c[i] + d];
    oracle_fe_reinit(elem_type, X, phi, dphi, JxW);
    double Volume = 0.0;
    for (int q = 0; q < nqp; q++) Volume += JxW[q];
    double A_b__average = 0.0, Tau__average = 0.0; /* :766-778 */
    for (int q = 0; q < nqp; q++) {
      double A_b = 0.0, Tau = 0.0;
      for (int l = 0; l < nen; l++) {
        A_b += phi[q * nen + l] * u[3 * (int64_t)c[l] + 1];
        Tau += phi[q * nen + l] * u[3 * (int64_t)c[l] + 2];
      }
      A_b__average += JxW[q] * A_b;
      Tau__average += JxW[q] * Tau;
    }
    out[4 * slot] = A_b__average / Volume;     /* :780 */
    out[4 * slot + 1] = Tau__average / Volume; /* :783 */
    int consider = 1; /* :787-798 */
    for (int n = 0; n < nen; n++) {
      const double A_b = u[3 * (int64_t)c[n] + 1];
      if (!(A_b >= r->A_b_min && A_b <= r->A_b_max)) { consider = 0; break; }
    }
    if (consider) out[4 * slot + 2] += Volume;
    consider = 1; /* :801-812 */
    for (int n = 0; n < nen; n++) {
      const double Tau = u[3 * (int64_t)c[n] + 2];
      if (!(Tau >= r->Tau_min && Tau <= r->Tau_max)) { consider = 0; break; }
    }
    if (consider) out[4 * slot + 3] += Volume;
  }
  return 0;
}

/* check_solution negativity clamp, src/pihna.C:785-790 */
void oracle_clamp_nonnegative(double* u, int64_t n) {
  for (int64_t i = 0; i < n; i++) if (u[i] < 0.0) u[i] = 0.0;
}
