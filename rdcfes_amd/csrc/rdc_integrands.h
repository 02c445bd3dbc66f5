// rdc_integrands.h — weak-form integrands of the reaction-diffusion-convection models in
// "coefficient form".
//
// Every Ke/Fe contribution of the reference callbacks has the shape
//
//   Ke_ab(i,j) += JxW * [ A_ab * phi_j*phi_i  +  phi_j * sum_k B_ab^k * (grad f_k . grad phi_i)
//                         +  D_ab * (grad phi_j . grad phi_i) ]
//   Fe_a(i)    += JxW * [ R_a * phi_i  +  sum_k RG_a^k * (grad f_k . grad phi_i) ]
//
// where A, B, D, R, RG depend only on the old solution at the quadrature point.  This file
// evaluates those point coefficients; the kernels (rdc_kernels.h) contract them with the shape
// data.  On TET4 (constant gradients) the contraction collapses into a handful of quadrature
// moments, which is what the fast path exploits.
//
// Reference lines restated here:
//   PIHNA  src/pihna.C:358-381 (constants), :444-509 (point nonlinearities), :514-747 (terms)
//   RIPF   src/ripf.C:377-408, :486-561, :566-662
//   HCC    src/coupled_hcc.C:450-461, :510-535, :540-637   (quirks of App. D reproduced)
#ifndef RDC_INTEGRANDS_H
#define RDC_INTEGRANDS_H

#include <math.h>
#include "../../include/rdc_assembly.h"

#if defined(__HIPCC__)
#define RDC_HD __host__ __device__ __forceinline__
#else
#define RDC_HD inline
#endif

namespace rdc {

// ---- generic coefficient container -------------------------------------------------------
template <int NV, int NG>
struct Coef {
  double A[NV][NV];      // mass-type
  double B[NV][NV][NG];  // phi_j * (grad f_k . grad phi_i)
  double D[NV][NV];      // stiffness-type
  double R[NV];          // rhs mass-type
  double RG[NV][NG];     // rhs gradient-type
  RDC_HD void zero() {
    for (int a = 0; a < NV; a++) {
      R[a] = 0.0;
      for (int k = 0; k < NG; k++) RG[a][k] = 0.0;
      for (int b = 0; b < NV; b++) {
        A[a][b] = 0.0;
        D[a][b] = 0.0;
        for (int k = 0; k < NG; k++) B[a][b][k] = 0.0;
      }
    }
  }
};

// 1/x.  The reference divides (IEEE); on the device a correctly rounded FP64 division costs ~11
// instructions, so quotients sharing a denominator use one reciprocal: v_rcp_f64 (~1e-8 rel) plus
// two Newton steps (<= 1 ulp off the IEEE quotient, far inside the 1e-10 parity bound).
// The special cases are those of the IEEE quotient 1.0 / x: the Newton steps alone turn x == +-0 (v_rcp: +-inf,
// fma(-0, inf, 1) = NaN) and x == +-inf into NaN, so the result passes through v_div_fixup_f64 (one instruction), which
// returns +-inf, +-0 and NaN exactly where the division does -- PIHNA's Ve_ = v / (c + h + v) with c + h + v == 0 and
// v > 0 is +inf as upstream (src/pihna.C:477-498 then takes the Ve_ >= 1 branch), not NaN.
RDC_HD double rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  return __builtin_amdgcn_div_fixup(r, x, 1.0);
#else
  return 1.0 / x;
#endif
}

// pow() kept out of line on the device: inlined into the generic row-gather kernel of a 5-unknown model (256 VGPRs
// + AGPR spills) the device-library pow produced NaN for positive arguments (observed with Proteas on TET4,
// ROCm 7.2); as a call it is correct everywhere, and the general-exponent path is the rare one.
#if defined(__HIP_DEVICE_COMPILE__) && defined(RDC_POW_INLINE)   // diagnostic builds only (tools/pow_inline_probe.sh)
__device__ __forceinline__ static double rdc_pow(double x, double e) { return pow(x, e); }
__device__ __forceinline__ static double rdc_pow_pos(double x, double e) { return exp(e * log(x)); }
#elif defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline)) static double rdc_pow(double x, double e) { return pow(x, e); }
// x^e for x > 0 (pow_pair: a crowding base 1 - T/K in (0, 1)) as exp(e log x): the device-library pow() spends most of its
// ~190 instructions on being correctly rounded over the whole plane (double-double logarithm, special cases); exp and log are
// each < 1 ulp, so the result is within |e log x| * 1.2e-16 + 2e-16 relative (<= 1e-14 for the bases that survive the
// min-vacancy cut-offs) of it -- four orders inside the 1e-10 parity bound -- at less than half the instructions.
__device__ __attribute__((noinline)) static double rdc_pow_pos(double x, double e) { return exp(e * log(x)); }
#else
static inline double rdc_pow(double x, double e) { return pow(x, e); }
static inline double rdc_pow_pos(double x, double e) { return pow(x, e); }
#endif

// x^e for the crowding functions.  The reference calls pow(1-Te, ek) with a real exponent
// (src/pihna.C:466); for the small integer exponents of the shipped inputs (ek = 3) the host
// selects EXP_MODE = that integer and the power is formed by multiplication (<= 1 ulp from pow).
// EXP_MODE 0 = general real exponent: one pow() and one reciprocal instead of upstream's two pow() calls
// (x^(e-1) = x^e / x, exact to an ulp).  p = x^e, pm1 = x^(e-1).
template <int EXP_MODE>
RDC_HD void pow_pair(double x, double e, double& p, double& pm1) {
  if (EXP_MODE == 1) { pm1 = 1.0; p = x; }
  else if (EXP_MODE == 2) { pm1 = x; p = x * x; }
  else if (EXP_MODE == 3) { pm1 = x * x; p = pm1 * x; }
  else if (EXP_MODE == 4) { pm1 = x * x * x; p = pm1 * x; }
  else if (EXP_MODE == 25) { pm1 = x * sqrt(x); p = pm1 * x; }  // e = 2.5 (run/RIPF133/input.dat): one sqrt, no pow
  else {
    // EXP_MODE 0: the exponent is a run-time value -- but a UNIFORM one (a kernel argument, held in scalar registers), so testing it
    // for the cheap cases costs a scalar compare and branch per call: small integers by multiplication, 2.5 with one square
    // root (each <= 1 ulp from pow), anything else through exp(e log x).  The dedicated instantiations above (EXP_MODE = the
    // shipped exponent of the model) only save these branches.
    if (e == 3.0) { pm1 = x * x; p = pm1 * x; }
    else if (e == 2.0) { pm1 = x; p = x * x; }
    else if (e == 2.5) { pm1 = x * sqrt(x); p = pm1 * x; }
    else if (e == 4.0) { pm1 = x * x * x; p = pm1 * x; }
    else if (e == 1.0) { pm1 = 1.0; p = x; }
    else { p = rdc_pow_pos(x, e); pm1 = p * rcp(x); }  // x^(e-1) = x^e / x for x > 0 (the only domain it is called on)
  }
}

// =========================================================================================
// PIHNA: unknowns (n, c, h, v, a); gradient fields k = 0..3 -> (c, h, v, a)
// =========================================================================================
struct PihnaK {  // src/pihna.C:358-381
  double DT2, Lambda, Kappa, Ka, ek;
  double iKappa, mekK;  // 1/Kappa, -ek/Kappa
  double nec_c, nec_h, nec_v;
  double dif_c, tax_c, dif_h, tax_h, prod_c, c2h, h2c, h2n;
  double dif_v, tax_v, prod_v;
  double sec_c, sec_h, upt, dec;
  // time-step-weighted rates of the moment form (rdc_tet4_pihna_moments.h): DT2 * rate, formed once on the host
  double Tn_c, Tn_h, Tn_v, Th2n, Tprod_c, Tc2h, Th2c, Tprod_v, Tdif_v, Tsec_c, Tsec_h, Tupt, Tdec;
  double Tdif_c, Ttax_c, Tdif_h, Ttax_h, Ttax_v;   // cell transport (general-parameter moment kernel)
};

struct Pihna {
  static constexpr int HEX_CL_POINTS = 1;   // k_hex8_cl: quadrature points per workgroup barrier
  static constexpr bool HEX_STAGED = false;  // node-staged HEX8 row gather measured slower for this model (tools/hex_ab.py)
  static constexpr bool HEX_REF_GRADS = true;  // HEX8 generic evaluator: inverse Jacobian applied to reference gradients (rdc_row.h)
  static constexpr int AUX_LOCAL_NODE = -1;  // aux nodal values of every element node are read
  static constexpr int NELEM = 0;  // per-element input doubles
  RDC_HD static void grad_post(const PihnaK&, double (*)[3], const double*) {}
  static constexpr int NV = 5, NG = 4, NAUX = 0;
  static constexpr int FAST_EXP_MODE = 3;  // exponent with a dedicated kernel instantiation (shipped value)
  using K = PihnaK;
  using C = Coef<NV, NG>;
  // which nodal array feeds gradient field k: index into u (>=0)
  RDC_HD static constexpr int grad_src(int k) { return k + 1; }
  RDC_HD static constexpr int row_order(int x) { return x; }
  // structural sparsity of coef(): entries outside these masks are identically zero
  RDC_HD static constexpr bool hasA(int a, int b) {
    return !((a == 0 && b == 4) || (a == 4 && b == 0) || (a == 1 && b == 4) || (a == 2 && b == 4));
  }
  RDC_HD static constexpr bool hasB(int a, int b, int k) {
    return b < 4 && ((a == 1 && (k == 0 || k == 2)) || (a == 2 && (k == 1 || k == 2)) || (a == 3 && (k == 2 || k == 3)));
  }
  RDC_HD static constexpr bool hasD(int a, int b) {
    return (a == 1 && (b == 1 || b == 3)) || (a == 2 && (b == 2 || b == 3)) || (a == 3 && (b == 3 || b == 4));
  }
  RDC_HD static constexpr bool hasRG(int a, int k) {
    return (a == 1 && (k == 0 || k == 2)) || (a == 2 && (k == 1 || k == 2)) || (a == 3 && (k == 2 || k == 3));
  }

  static inline K derive(const rdc_pihna_params& p) {
    K k;
    k.DT2 = p.time_step / 2.0;
    k.Lambda = p.cells_min_capacity;
    k.Kappa = p.cells_max_capacity;
    k.Ka = p.cytokines_max_capacity;
    k.ek = p.cells_max_capacity_exponent;
    k.iKappa = 1.0 / k.Kappa;
    k.mekK = -k.ek / k.Kappa;          // (-ek/Kappa_k), :470
    k.nec_c = p.necrosis_c / k.Kappa;  // :364-366
    k.nec_h = p.necrosis_h / k.Kappa;
    k.nec_v = p.necrosis_v / k.Kappa;
    k.dif_c = p.diffuse_c; k.tax_c = p.taxis_c; k.dif_h = p.diffuse_h; k.tax_h = p.taxis_h;
    k.prod_c = p.produce_c; k.c2h = p.switch_c2h; k.h2c = p.switch_h2c; k.h2n = p.switch_h2n;
    k.dif_v = p.diffuse_v; k.tax_v = p.taxis_v; k.prod_v = p.produce_v;
    k.sec_c = p.secrete_a_c; k.sec_h = p.secrete_a_h; k.upt = p.uptake_a_v; k.dec = p.decay_a;
    k.Tn_c = k.DT2 * k.nec_c; k.Tn_h = k.DT2 * k.nec_h; k.Tn_v = k.DT2 * k.nec_v; k.Th2n = k.DT2 * k.h2n;
    k.Tprod_c = k.DT2 * k.prod_c; k.Tc2h = k.DT2 * k.c2h; k.Th2c = k.DT2 * k.h2c; k.Tprod_v = k.DT2 * k.prod_v;
    k.Tdif_v = k.DT2 * k.dif_v; k.Tsec_c = k.DT2 * k.sec_c; k.Tsec_h = k.DT2 * k.sec_h; k.Tupt = k.DT2 * k.upt;
    k.Tdec = k.DT2 * k.dec;
    k.Tdif_c = k.DT2 * k.dif_c; k.Ttax_c = k.DT2 * k.tax_c; k.Tdif_h = k.DT2 * k.dif_h; k.Ttax_h = k.DT2 * k.tax_h; k.Ttax_v = k.DT2 * k.tax_v;
    return k;
  }
  static inline double exponent(const K& k) { return k.ek; }

  // point nonlinearities shared by all terms
  struct Pt {
    double n, c, h, v, a;
    double Tau, dT;            // crowding and its (common) derivative          :444-472
    double Ve, rV;             // vascular fraction and 1/(c+h+v) (0 in the clamped branches):
                               // Ve__dc = Ve__dh = -Ve*rV, Ve__dv = (1-Ve)*rV    :474-499
    // the cytokine uptake Ua, Ua_da (:501-502) is used by the v equation only: it is derived inside coef(), so it
    // does not occupy 5 points x 2 doubles of registers while the other four equation rows are evaluated
  };

  template <int EXP_MODE>
  RDC_HD static void point(const K& k, const double* u, const double* /*aux*/, Pt& s) {
    s.n = u[0]; s.c = u[1]; s.h = u[2]; s.v = u[3]; s.a = u[4];
    const double Te = (s.n + s.c + s.h + s.v) * k.iKappa;
    if (Te <= 0.0) { s.Tau = 1.0; s.dT = 0.0; }
    else if (Te >= 1.0) { s.Tau = 0.0; s.dT = 0.0; }
    else {
      double p, pm1;
      pow_pair<EXP_MODE>(1.0 - Te, k.ek, p, pm1);
      s.Tau = p;
      s.dT = k.mekK * pm1;
    }
    const double chv = s.c + s.h + s.v;
    const double rchv = rcp(chv);
    const double Ve_ = s.v * rchv;  // NaN for chv == 0 drops into the last branch, as upstream
    if (Ve_ <= 0.0) { s.Ve = 0.0; s.rV = 0.0; }
    else if (Ve_ >= 1.0) { s.Ve = 1.0; s.rV = 0.0; }
    else { s.Ve = Ve_; s.rV = rchv; }
  }

  // coefficients of equation row `a` only would be enough for a row kernel, but the full set is
  // cheap relative to the contraction; the compiler drops what a caller does not consume.
  RDC_HD static void coef(const K& k, const Pt& s, C& o) {
    o.zero();
    const double T = k.DT2;
    const double oneVe = 1.0 - s.Ve;
    const double Ve_dc = -s.Ve * s.rV, Ve_dv = oneVe * s.rV;
    const double nVe_dc = -Ve_dc, nVe_dv = -Ve_dv;  // (-Ve__dc) etc. as written upstream
    // thresholded transport coefficients, :504-509 (not stored in Pt: registers are scarcer than compares)
    const double dif_c = (s.c > k.Lambda ? k.dif_c : 0.0), tax_c = (s.c > k.Lambda ? k.tax_c : 0.0);
    const double dif_h = (s.h > k.Lambda ? k.dif_h : 0.0), tax_h = (s.h > k.Lambda ? k.tax_h : 0.0);
    const double dif_v = (s.v > k.Lambda ? k.dif_v : 0.0), tax_v = (s.v > k.Lambda ? k.tax_v : 0.0);
    // ---- n equation, :514-522 and :571-597
    o.R[0] = s.n + T * (k.nec_c * s.c * s.n + k.nec_h * s.h * s.n + k.nec_v * s.v * s.n + k.h2n * oneVe * s.h);
    o.A[0][0] = 1.0 - T * (k.nec_c * s.c + k.nec_h * s.h + k.nec_v * s.v);
    o.A[0][1] = -T * (k.nec_c * s.n + k.h2n * nVe_dc * s.h);
    o.A[0][2] = -T * (k.nec_h * s.n + k.h2n * nVe_dc * s.h + k.h2n * oneVe);
    o.A[0][3] = -T * (k.nec_v * s.n + k.h2n * nVe_dv * s.h);
    // ---- c equation, :524-534 and :599-641   (gradient fields: 0 = c, 2 = v)
    const double pc = k.prod_c * s.dT * s.c;
    o.R[1] = s.c + T * (k.prod_c * s.Tau * s.c - k.c2h * oneVe * s.c + k.h2c * s.Ve * s.h - k.nec_c * s.c * s.n);
    o.RG[1][0] = -T * dif_c * s.Tau;
    o.RG[1][2] = -T * tax_c * s.Tau * s.c;
    o.A[1][0] = -T * (pc - k.nec_c * s.c);
    o.A[1][1] = 1.0 - T * (k.prod_c * s.Tau + pc - k.c2h * oneVe - k.c2h * nVe_dc * s.c + k.h2c * Ve_dc * s.h - k.nec_c * s.n);
    o.A[1][2] = -T * (pc - k.c2h * nVe_dc * s.c + k.h2c * Ve_dc * s.h + k.h2c * s.Ve);
    o.A[1][3] = -T * (pc - k.c2h * nVe_dv * s.c + k.h2c * Ve_dv * s.h);
    {
      const double bc = T * dif_c * s.dT, bv = T * tax_c * s.dT * s.c;
      for (int b = 0; b < 4; b++) { o.B[1][b][0] = bc; o.B[1][b][2] = bv; }
      o.B[1][1][2] += T * tax_c * s.Tau;
    }
    o.D[1][1] = T * dif_c * s.Tau;
    o.D[1][3] = T * tax_c * s.Tau * s.c;
    // ---- h equation, :536-546 and :643-684   (gradient fields: 1 = h, 2 = v)
    o.R[2] = s.h + T * (k.c2h * oneVe * s.c - k.h2c * s.Ve * s.h - k.nec_h * s.h * s.n - k.h2n * oneVe * s.h);
    o.RG[2][1] = -T * dif_h * s.Tau;
    o.RG[2][2] = -T * tax_h * s.Tau * s.h;
    o.A[2][0] = T * k.nec_h * s.h;
    o.A[2][1] = -T * (k.c2h * oneVe + k.c2h * nVe_dc * s.c - k.h2c * Ve_dc * s.h - k.h2n * nVe_dc * s.h);
    o.A[2][2] = 1.0 - T * (k.c2h * nVe_dc * s.c - k.h2c * Ve_dc * s.h - k.h2c * s.Ve - k.nec_h * s.n -
                           k.h2n * nVe_dc * s.h - k.h2n * oneVe);
    o.A[2][3] = -T * (k.c2h * nVe_dv * s.c - k.h2c * Ve_dv * s.h - k.h2n * nVe_dv * s.h);
    {
      const double bh = T * dif_h * s.dT, bv = T * tax_h * s.dT * s.h;
      for (int b = 0; b < 4; b++) { o.B[2][b][1] = bh; o.B[2][b][2] = bv; }
      o.B[2][2][2] += T * tax_h * s.Tau;
    }
    o.D[2][2] = T * dif_h * s.Tau;
    o.D[2][3] = T * tax_h * s.Tau * s.h;
    // ---- v equation, :548-556 and :686-724   (gradient fields: 2 = v, 3 = a)
    const double raK = rcp(s.a + k.Ka);                                          // :501-502
    const double Ua = s.a * raK, Ua_da = raK - Ua * raK;
    const double pv = k.prod_v * s.dT * Ua * s.v;
    o.R[3] = s.v + T * (k.prod_v * s.Tau * Ua * s.v - k.nec_v * s.v * s.n);
    o.RG[3][2] = -T * dif_v * s.Tau;
    o.RG[3][3] = -T * tax_v * s.Tau * s.v;
    o.A[3][0] = -T * (pv - k.nec_v * s.v);
    o.A[3][1] = -T * pv;
    o.A[3][2] = -T * pv;
    o.A[3][3] = 1.0 - T * (pv - k.nec_v * s.n);
    o.A[3][4] = -T * (k.prod_v * s.Tau * Ua_da * s.v);
    {
      const double bv = T * dif_v * s.dT, ba = T * tax_v * s.dT * s.v;
      for (int b = 0; b < 4; b++) { o.B[3][b][2] = bv; o.B[3][b][3] = ba; }
      o.B[3][3][3] += T * tax_v * s.Tau;
    }
    o.D[3][3] = T * dif_v * s.Tau;
    o.D[3][4] = T * tax_v * s.Tau * s.v;
    // ---- a equation, :558-566 and :726-747
    o.R[4] = s.a + T * (k.sec_c * s.c + k.sec_h * s.h - k.upt * s.v * s.a - k.dec * s.a);
    o.A[4][1] = -T * k.sec_c;
    o.A[4][2] = -T * k.sec_h;
    o.A[4][3] = T * k.upt * s.a;
    o.A[4][4] = 1.0 + T * (k.upt * s.v + k.dec);
  }
};

// PIHNA with the cell transport terms switched off: diffuse/c = taxis/c = diffuse/h = taxis/h = taxis/v = 0
// exactly (the shipped run/PIHNA/input.dat; only diffuse/v is a transport term there).
// Same point functions and coefficients; only the structural masks shrink, so the factored kernel
// drops the identically-zero products (0 * finite = 0 upstream) instead of evaluating them.  The
// host selects this variant from the parameter values; any non-zero entry uses Pihna.
struct PihnaNoCellTransport : Pihna {
  static inline bool applies(const rdc_pihna_params& p) {
    return p.diffuse_c == 0.0 && p.taxis_c == 0.0 && p.diffuse_h == 0.0 && p.taxis_h == 0.0 && p.taxis_v == 0.0;
  }
  RDC_HD static constexpr bool hasB(int a, int b, int k) { return a == 3 && b < 4 && k == 2; }  // diffuse/v * dTau
  RDC_HD static constexpr bool hasD(int a, int b) { return a == 3 && b == 3; }
  RDC_HD static constexpr bool hasRG(int a, int k) { return a == 3 && k == 2; }
};

// The same with a slim per-point state: only the five interpolated unknowns are kept per point (5 x 5 doubles
// instead of 5 x 11) and Tau, Ve, Ua are re-derived inside coef() for the equation row that needs them -- more
// FP64 work per pair (1,774 vs 1,374 VALU instructions), but the kernel then fits three waves per SIMD (168
// registers, 44 B of scratch).  Cube exponent only (the host selects it for the shipped exponent 3).
// EXPERIMENTAL, rdc_set_option("slim", 1) + ("occupancy", 3): measured 2.95 ms vs 2.93 ms for the default on
// K(119) -- the third wave buys 15% (3.46 -> 2.95 ms at equal work), the re-derivation costs it again
// (profiles/r01e_ab_slim_occupancy3.txt).  Kept as the starting point for a cheaper re-derivation.
struct PihnaNoCellTransportSlim : PihnaNoCellTransport {
  struct Pt { double n, c, h, v, a; };
  template <int EXP_MODE>
  RDC_HD static void point(const K&, const double* u, const double*, Pt& s) {
    s.n = u[0]; s.c = u[1]; s.h = u[2]; s.v = u[3]; s.a = u[4];
  }
  RDC_HD static void coef(const K& k, const Pt& p, C& o) {
    double u[5] = {p.n, p.c, p.h, p.v, p.a};
#if defined(__HIP_DEVICE_COMPILE__)
    // opaque copies: without them the compiler hoists the derivation out of the equation-row loop and keeps all
    // of it live again (256 registers + 252 B of scratch instead of 186 registers)
#pragma unroll
    for (int v = 0; v < 5; v++) asm volatile("" : "+v"(u[v]));
#endif
    Pihna::Pt full;
    Pihna::point<3>(k, u, nullptr, full);
    Pihna::coef(k, full, o);
  }
};

// =========================================================================================
// RIPF: unknowns (HU, cc, fb); aux nodal (cc_dtime, fb_dtime, RT_total);
// gradient fields k: 0 = fb, 1 = HU, 2 = RT_total (normalised to unit length, :481-484)
// =========================================================================================
struct RipfK {
  double DT2;
  double VF_fixed, VF_exp, VF_min;
  double phi_cc_B, phi_cc_D, phi_cc, phi_fb_B, phi_fb_D, phi_fb, phi_tol;
  double kappa, kappa_RT_c, delta, delta_RT_a, delta_RT_b;
  double lambda, lambda_RT_r, lambda_HU_r, omicro, omicro_RT_r, omicro_fb_b;
  double i_lambda_RT_r, i_lambda_HU_r, i_omicro_RT_r;  // reciprocals: the per-point divisions become multiplies
  double omega, diffusion, haptotaxis, radiotaxis;
};

struct Ripf {
  static constexpr int HEX_CL_POINTS = 1;   // k_hex8_cl: quadrature points per workgroup barrier
  static constexpr bool HEX_STAGED = false;  // node-staged HEX8 row gather measured slower for this model (tools/hex_ab.py)
  static constexpr bool HEX_REF_GRADS = false; // three gradient fields: forming grad phi of all nodes measured faster (H(80) 2.08 vs 2.36 ms)
  static constexpr int AUX_LOCAL_NODE = -1;  // aux nodal values of every element node are read
  static constexpr int NELEM = 0;  // per-element input doubles
  RDC_HD static void grad_post(const RipfK&, double (*)[3], const double*) {}
  static constexpr int NV = 3, NG = 3, NAUX = 3;
  static constexpr int FAST_EXP_MODE = 25;  // volume_fraction/exponent = 2.5 in run/RIPF133/input.dat
  using K = RipfK;
  using C = Coef<NV, NG>;
  // gradient field sources: >=0 index into u, <0 -> aux index (-1-k)
  RDC_HD static constexpr int grad_src(int k) { return k == 0 ? 2 : (k == 1 ? 0 : -3); }
  RDC_HD static constexpr int row_order(int x) { return x; }
  RDC_HD static constexpr bool hasA(int a, int b) { return !((a == 1 && b == 0)); }
  RDC_HD static constexpr bool hasB(int a, int b, int) { return a == 2 && b >= 1; }
  RDC_HD static constexpr bool hasD(int a, int b) { return a == 2 && (b == 0 || b == 2); }
  RDC_HD static constexpr bool hasRG(int a, int) { return a == 2; }

  static inline K derive(const rdc_ripf_params& p) {
    K k;
    k.DT2 = p.time_step / 2.0;
    k.VF_fixed = p.VolFr_stroma + p.VolFr_parenchyma;  // summed left-to-right as in :499
    k.VF_exp = p.VolFr_exponent;
    k.VF_min = p.VolFr_min_vacant;
    k.phi_cc_B = p.phi_cc_B; k.phi_cc_D = p.phi_cc_D; k.phi_cc = p.phi_cc;
    k.phi_fb_B = p.phi_fb_B; k.phi_fb_D = p.phi_fb_D; k.phi_fb = p.phi_fb; k.phi_tol = p.phi_tol;
    k.kappa = p.kappa; k.kappa_RT_c = p.kappa_RT_c;
    k.delta = p.delta; k.delta_RT_a = p.delta_RT_a; k.delta_RT_b = p.delta_RT_b;
    k.lambda = p.lambda;
    k.lambda_RT_r = p.lambda_RT_r ? p.lambda_RT_r : (double)p.RT_dose_total_max;  // :398-403
    k.lambda_HU_r = p.lambda_HU_r;
    k.omicro = p.omicro;
    k.omicro_RT_r = p.omicro_RT_r ? p.omicro_RT_r : (double)p.RT_dose_total_max;
    k.omicro_fb_b = p.omicro_fb_b;
    k.omega = p.omega; k.diffusion = p.diffusion; k.haptotaxis = p.haptotaxis; k.radiotaxis = p.radiotaxis;
    k.i_lambda_RT_r = 1.0 / k.lambda_RT_r; k.i_lambda_HU_r = 1.0 / k.lambda_HU_r; k.i_omicro_RT_r = 1.0 / k.omicro_RT_r;
    return k;
  }
  static inline double exponent(const K& k) { return k.VF_exp; }

  // Per point only the interpolated inputs and the transcendental results are kept (10 doubles instead of
  // 20: the factored TET4 kernels hold 5 points at once); the select-and-multiply quantities are re-derived in
  // coef(), where the compiler drops what the requested equation row does not use.
  struct Pt {
    double HU, cc, fb, cc_dt, fb_dt, RT;
    double kappa_RT, delta_RT, Tau, dTau;
  };

  template <int EXP_MODE>
  RDC_HD static void point(const K& k, const double* u, const double* aux, Pt& s) {
    s.HU = u[0]; s.cc = u[1]; s.fb = u[2];
    s.cc_dt = aux[0]; s.fb_dt = aux[1];
    const double RT = aux[2];
    s.RT = RT;
    s.kappa_RT = k.kappa * exp(-k.kappa_RT_c * RT);                                   // :486
    s.delta_RT = k.delta * (1.0 - exp(-k.delta_RT_a * RT - k.delta_RT_b * (RT * RT)));  // :487
    const double VF = k.VF_fixed + (s.cc + s.fb);                                     // :498-499
    s.Tau = 0.0; s.dTau = 0.0;
    if (VF < 1.0) {                                                                   // :503-514
      double p, pm1;
      pow_pair<EXP_MODE>(1.0 - VF, k.VF_exp, p, pm1);
      s.Tau = p;
      s.dTau = -k.VF_exp * pm1;
      if (s.Tau < k.VF_min) { s.Tau = 0.0; s.dTau = 0.0; }
    }
  }

  RDC_HD static void coef(const K& k, const Pt& p, C& o) {
    o.zero();
    const double T = k.DT2;
    struct {
      double HU, cc, fb, cc_dt, fb_dt, kappa_RT, delta_RT, lambda_RT, omicro_RT, eps_cc, eps_fb, Tau, dTau, Koppa, Koppa_dcc,
          Lom, Lom_dHU, Lom_dfb, Ome, Ome_dfb;
    } s;
    s.HU = p.HU; s.cc = p.cc; s.fb = p.fb; s.cc_dt = p.cc_dt; s.fb_dt = p.fb_dt;
    s.kappa_RT = p.kappa_RT; s.delta_RT = p.delta_RT; s.Tau = p.Tau; s.dTau = p.dTau;
    const double RT = p.RT;
    s.lambda_RT = k.lambda * (RT * k.i_lambda_RT_r);                                     // :488
    {
      const double r = RT * k.i_omicro_RT_r;                                           // :489
      const double x = 4.0 * (r - r * r);
      s.omicro_RT = k.omicro * (x < 0.0 ? 0.0 : x);
    }
    s.eps_cc = 0.0; s.eps_fb = 0.0;                                                   // :491-496
    if (s.cc_dt > k.phi_tol) s.eps_cc = k.phi_cc_B; else if (s.cc_dt < -k.phi_tol) s.eps_cc = k.phi_cc_D;
    if (s.fb_dt > k.phi_tol) s.eps_fb = k.phi_fb_B; else if (s.fb_dt < -k.phi_tol) s.eps_fb = k.phi_fb_D;
    s.Koppa = 0.0; s.Koppa_dcc = 0.0;                                                 // :516-523
    if (s.cc >= 0.0 && s.cc < 1.0) { s.Koppa = 4.0 * (s.cc - s.cc * s.cc); s.Koppa_dcc = 4.0 - 8.0 * s.cc; }
    s.Lom = s.Lom_dHU = s.Lom_dfb = 0.0; s.Ome = s.Ome_dfb = 0.0;                     // :525-561
    if (s.fb >= 0.0 && s.fb < 1.0) {
      const double f2 = 1.0 - s.fb * s.fb;
      if (s.HU > k.lambda_HU_r && s.HU < 0.0) {
        s.Lom = f2 * (s.HU * k.i_lambda_HU_r);
        s.Lom_dHU = f2 * k.i_lambda_HU_r;
        s.Lom_dfb = -(2.0 * s.fb) * (s.HU * k.i_lambda_HU_r);
      } else if (s.HU < k.lambda_HU_r) {
        s.Lom = f2;
        s.Lom_dfb = -(2.0 * s.fb);
      }
      if (s.fb <= k.omicro_fb_b) {
        s.Ome = 4.0 * (k.omicro_fb_b - k.omicro_fb_b * k.omicro_fb_b);
      } else {
        s.Ome = 4.0 * (s.fb - s.fb * s.fb);
        s.Ome_dfb = 4.0 - 8.0 * s.fb;
      }
    }
    // HU equation, :566-574, :599-613
    o.R[0] = s.HU + T * (s.eps_cc * s.cc + s.eps_fb * s.fb + k.phi_cc * s.cc_dt + k.phi_fb * s.fb_dt);
    o.A[0][0] = 1.0;
    o.A[0][1] = -T * s.eps_cc;
    o.A[0][2] = -T * s.eps_fb;
    // cc equation, :576-582, :615-627
    o.R[1] = s.cc + T * (s.kappa_RT * s.Tau * s.Koppa - s.delta_RT * s.cc);
    o.A[1][1] = 1.0 - T * (s.kappa_RT * s.dTau * s.Koppa + s.kappa_RT * s.Tau * s.Koppa_dcc - s.delta_RT);
    o.A[1][2] = -T * (s.kappa_RT * s.dTau * s.Koppa);
    // fb equation, :584-594, :629-662  (Lombda__dcc, Omecro__dHU, Omecro__dcc are identically 0)
    const double src = s.lambda_RT * s.dTau * s.Lom + s.omicro_RT * s.dTau * s.Ome;
    o.R[2] = s.fb + T * (s.lambda_RT * s.Tau * s.Lom + s.omicro_RT * s.Tau * s.Ome - k.omega * s.fb);
    o.RG[2][0] = -T * k.diffusion * s.Tau;
    o.RG[2][1] = -T * k.haptotaxis * s.Tau * s.fb;
    o.RG[2][2] = -T * k.radiotaxis * s.Tau * s.fb;
    o.A[2][0] = -T * (s.lambda_RT * s.Tau * s.Lom_dHU);
    o.D[2][0] = T * k.haptotaxis * s.Tau * s.fb;
    o.A[2][1] = -T * src;
    o.A[2][2] = 1.0 - T * (src + s.lambda_RT * s.Tau * s.Lom_dfb + s.omicro_RT * s.Tau * s.Ome_dfb - k.omega);
    {
      const double b0 = T * k.diffusion * s.dTau, b1 = T * k.haptotaxis * s.dTau * s.fb,
                   b2 = T * k.radiotaxis * s.dTau * s.fb;
      o.B[2][1][0] = b0; o.B[2][1][1] = b1; o.B[2][1][2] = b2;
      o.B[2][2][0] = b0;
      o.B[2][2][1] = b1 + T * k.haptotaxis * s.Tau;
      o.B[2][2][2] = b2 + T * k.radiotaxis * s.Tau;
    }
    o.D[2][2] = T * k.diffusion * s.Tau;
  }
};

// RIPF with the cancer-growth, HU-rate, second fibrosis source and radiotaxis terms switched off: kappa = omicro =
// radiotaxis = 0 and every HU/phi rate = 0 exactly -- the shipped run/RIPF133/input.dat, where only the radiotherapy
// kill of cc and the HU-driven fibrosis with diffusion + haptotaxis act.  Upstream multiplies by those zeros
// (0 * finite = 0), so dropping the products is exact for finite states: one exp instead of two per point, no unit
// radiotherapy gradient, 5 instead of 8 coefficient blocks, and a 7-double point state (Ripf: 10).  The host selects
// this variant from the parameter values; anything else uses Ripf.
struct RipfReduced : Ripf {
  static inline bool applies(const rdc_ripf_params& p) {
    return p.kappa == 0.0 && p.kappa_RT_c >= 0.0 && p.omicro == 0.0 && p.radiotaxis == 0.0 && p.phi_cc_B == 0.0 &&
           p.phi_cc_D == 0.0 && p.phi_cc == 0.0 && p.phi_fb_B == 0.0 && p.phi_fb_D == 0.0 && p.phi_fb == 0.0;
  }
  RDC_HD static constexpr bool hasA(int a, int b) { return (a == 0 && b == 0) || (a == 1 && b == 1) || a == 2; }
  RDC_HD static constexpr bool hasB(int a, int b, int k) { return a == 2 && b >= 1 && k < 2; }
  RDC_HD static constexpr bool hasD(int a, int b) { return a == 2 && (b == 0 || b == 2); }
  RDC_HD static constexpr bool hasRG(int a, int k) { return a == 2 && k < 2; }

  struct Pt { double HU, cc, fb, RT, delta_RT, Tau, dTau; };

  template <int EXP_MODE>
  RDC_HD static void point(const K& k, const double* u, const double* aux, Pt& s) {
    s.HU = u[0]; s.cc = u[1]; s.fb = u[2];
    const double RT = aux[2];
    s.RT = RT;
    s.delta_RT = k.delta * (1.0 - exp(-k.delta_RT_a * RT - k.delta_RT_b * (RT * RT)));  // :487
    const double VF = k.VF_fixed + (s.cc + s.fb);                                     // :498-499
    s.Tau = 0.0; s.dTau = 0.0;
    if (VF < 1.0) {                                                                   // :503-514
      double p, pm1;
      pow_pair<EXP_MODE>(1.0 - VF, k.VF_exp, p, pm1);
      s.Tau = p;
      s.dTau = -k.VF_exp * pm1;
      if (s.Tau < k.VF_min) { s.Tau = 0.0; s.dTau = 0.0; }
    }
  }

  RDC_HD static void coef(const K& k, const Pt& p, C& o) {
    o.zero();
    const double T = k.DT2;
    const double lambda_RT = k.lambda * (p.RT * k.i_lambda_RT_r);                      // :488
    double Lom = 0.0, Lom_dHU = 0.0, Lom_dfb = 0.0;                                    // :525-547
    if (p.fb >= 0.0 && p.fb < 1.0) {
      const double f2 = 1.0 - p.fb * p.fb;
      if (p.HU > k.lambda_HU_r && p.HU < 0.0) {
        Lom = f2 * (p.HU * k.i_lambda_HU_r);
        Lom_dHU = f2 * k.i_lambda_HU_r;
        Lom_dfb = -(2.0 * p.fb) * (p.HU * k.i_lambda_HU_r);
      } else if (p.HU < k.lambda_HU_r) {
        Lom = f2;
        Lom_dfb = -(2.0 * p.fb);
      }
    }
    // HU equation: every rate is zero
    o.R[0] = p.HU;
    o.A[0][0] = 1.0;
    // cc equation: radiotherapy kill only
    o.R[1] = p.cc - T * (p.delta_RT * p.cc);
    o.A[1][1] = 1.0 + T * p.delta_RT;
    // fb equation, :584-594, :629-662 without the omicro and radiotaxis terms
    const double src = lambda_RT * p.dTau * Lom;
    o.R[2] = p.fb + T * (lambda_RT * p.Tau * Lom - k.omega * p.fb);
    o.RG[2][0] = -T * k.diffusion * p.Tau;
    o.RG[2][1] = -T * k.haptotaxis * p.Tau * p.fb;
    o.A[2][0] = -T * (lambda_RT * p.Tau * Lom_dHU);
    o.D[2][0] = T * k.haptotaxis * p.Tau * p.fb;
    o.A[2][1] = -T * src;
    o.A[2][2] = 1.0 - T * (src + lambda_RT * p.Tau * Lom_dfb - k.omega);
    {
      const double b0 = T * k.diffusion * p.dTau, b1 = T * k.haptotaxis * p.dTau * p.fb;
      o.B[2][1][0] = b0; o.B[2][1][1] = b1;
      o.B[2][2][0] = b0;
      o.B[2][2][1] = b1 + T * k.haptotaxis * p.Tau;
    }
    o.D[2][2] = T * k.diffusion * p.Tau;
  }
};

// =========================================================================================
// HCC: unknowns (l, c, n); gradient field 0 = c.  GRAD_sigma == 0 (src/coupled_hcc.C:508), so
// every mechano term vanishes and is omitted.  Quirks kept: spurious capacity term in blocks
// [0][1],[0][2],[1][0]; the d/dn block of the c equation lands in [1][1] (App. D.1-2).
// =========================================================================================
struct HccK {
  double DT2, Lambda, Kappa, ek, prod_l, dif_c, prod_c, nec_l, nec_c;
};

struct Hcc {
  static constexpr int HEX_CL_POINTS = 2;   // k_hex8_cl: quadrature points per workgroup barrier (two: 2.7 vs 3.0 ms on H(126); the heavier models spill)
  static constexpr bool HEX_STAGED = true;   // node-staged HEX8 row gather: 1.32 -> 1.07 ms on H(80) (tools/hex_ab.py)
  static constexpr bool HEX_REF_GRADS = true;
  static constexpr int AUX_LOCAL_NODE = -1;  // aux nodal values of every element node are read
  static constexpr int NELEM = 0;  // per-element input doubles
  RDC_HD static void grad_post(const HccK&, double (*)[3], const double*) {}
  static constexpr int NV = 3, NG = 1, NAUX = 0;
  static constexpr int FAST_EXP_MODE = 3;
  using K = HccK;
  using C = Coef<NV, NG>;
  RDC_HD static constexpr int grad_src(int) { return 1; }
  RDC_HD static constexpr int row_order(int x) { return x; }
  RDC_HD static constexpr bool hasA(int a, int b) { return !(a == 1 && b == 2); }
  RDC_HD static constexpr bool hasB(int a, int b, int) { return a == 1 && b <= 1; }
  RDC_HD static constexpr bool hasD(int a, int b) { return a == 1 && b == 1; }
  RDC_HD static constexpr bool hasRG(int a, int) { return a == 1; }

  static inline K derive(const rdc_hcc_params& p) {
    K k;
    k.DT2 = p.time_step / 2.0;
    k.Lambda = p.cells_min_capacity;
    k.Kappa = p.cells_max_capacity;
    k.ek = p.cells_max_capacity_exponent;
    k.prod_l = p.produce_l; k.dif_c = p.diffuse_c; k.prod_c = p.produce_c;
    k.nec_l = p.necrosis_l / k.Kappa;  // :459-460
    k.nec_c = p.necrosis_c / k.Kappa;
    return k;
  }
  static inline double exponent(const K& k) { return k.ek; }

  struct Pt { double l, c, n, Tau, dT, dif_c; };

  template <int EXP_MODE>
  RDC_HD static void point(const K& k, const double* u, const double* /*aux*/, Pt& s) {
    s.l = u[0]; s.c = u[1]; s.n = u[2];
    const double Te = (s.l + s.c + s.n) / k.Kappa;  // :513
    if (Te <= 0.0) { s.Tau = 1.0; s.dT = 0.0; }
    else if (Te >= 1.0) { s.Tau = 0.0; s.dT = 0.0; }
    else {
      double p, pm1;
      pow_pair<EXP_MODE>(1.0 - Te, k.ek, p, pm1);
      s.Tau = p;
      s.dT = (-k.ek / k.Kappa) * pm1;
    }
    s.dif_c = (s.c > k.Lambda ? k.dif_c : 0.0);  // :534
  }

  RDC_HD static void coef(const K& k, const Pt& s, C& o) {
    o.zero();
    const double T = k.DT2;
    const double pl = k.prod_l * s.dT * s.l, pc = k.prod_c * s.dT * s.c;
    o.R[0] = s.l + T * (k.prod_l * s.Tau * s.l - k.nec_l * s.l * s.n);               // :540-546
    o.A[0][0] = 1.0 - T * (k.prod_l * s.Tau + pl - k.nec_l * s.n);                   // :569-576
    o.A[0][1] = 1.0 - T * pl;                                                        // :577-582
    o.A[0][2] = 1.0 - T * (pl - k.nec_l * s.l);                                      // :583-589
    o.R[1] = s.c + T * (k.prod_c * s.Tau * s.c - k.nec_c * s.c * s.n);               // :548-556
    o.RG[1][0] = -T * s.dif_c * s.Tau;
    o.A[1][0] = 1.0 - T * pc;                                                        // :591-598
    o.B[1][0][0] = T * s.dif_c * s.dT;
    o.A[1][1] = (1.0 - T * (k.prod_c * s.Tau + pc - k.nec_c * s.n))                  // :599-610
              + (1.0 - T * (pc - k.nec_c * s.c));                                    // :611-619
    o.B[1][1][0] = 2.0 * (T * s.dif_c * s.dT);
    o.D[1][1] = T * s.dif_c * s.Tau;
    o.R[2] = s.n + T * (k.nec_l * s.l * s.n + k.nec_c * s.c * s.n);                  // :558-564
    o.A[2][0] = -T * k.nec_l * s.n;                                                  // :621-625
    o.A[2][1] = -T * k.nec_c * s.n;                                                  // :626-630
    o.A[2][2] = 1.0 - T * (k.nec_l * s.l + k.nec_c * s.c);                           // :631-637
  }
};

// HCC with every rate zero (produce/l, produce/c, diffuse/c, necrosis/l, necrosis/c = 0 exactly): the shipped
// run/Coupled/HCC/input.dat, which gives only the capacity keys.  What is left of src/coupled_hcc.C:540-637 are the
// mass-type blocks with upstream's constants -- including the spurious capacity blocks [0][1], [0][2], [1][0] = 1 and
// the doubled [1][1] = 2 (App. D.1-2) -- and rhs = old solution; the blocks [2][0], [2][1] are -T*0*n = -0.  No
// gradient, no pow.  Exact for finite states; the host selects it from the parameter values.
struct HccMassOnly : Hcc {
  static constexpr int HEX_CL_POINTS = 1;   // stays at 148 registers = three workgroups per CU
  static inline bool applies(const rdc_hcc_params& p) {
    return p.produce_l == 0.0 && p.produce_c == 0.0 && p.diffuse_c == 0.0 && p.necrosis_l == 0.0 && p.necrosis_c == 0.0;
  }
  RDC_HD static constexpr bool hasA(int a, int b) { return a == 0 || (a == 1 && b <= 1) || (a == 2 && b == 2); }
  RDC_HD static constexpr bool hasB(int, int, int) { return false; }
  RDC_HD static constexpr bool hasD(int, int) { return false; }
  RDC_HD static constexpr bool hasRG(int, int) { return false; }
  struct Pt { double l, c, n; };
  template <int EXP_MODE>
  RDC_HD static void point(const K&, const double* u, const double* /*aux*/, Pt& s) { s.l = u[0]; s.c = u[1]; s.n = u[2]; }
  RDC_HD static void coef(const K&, const Pt& s, C& o) {
    o.zero();
    o.R[0] = s.l; o.R[1] = s.c; o.R[2] = s.n;
    o.A[0][0] = 1.0; o.A[0][1] = 1.0; o.A[0][2] = 1.0;
    o.A[1][0] = 1.0; o.A[1][1] = 2.0;
    o.A[2][2] = 1.0;
  }
};

// exponent -> EXP_MODE (0 = general pow)
// =========================================================================================
// ADPM (src/adpm.C:324-652): unknowns (PrP, A_b, Tau).  Gradient fields: 0 = grad A_b, 1 = grad Tau,
// 2 = tract_A_b, 3 = tract_Tau -- the element's tract vector (the "Tracts" system, :448-453) with the sign of
// its projection on the unit gradient when that exceeds cos(angle), else zero (:474-493); they enter the weak
// form exactly like gradient fields, so grad_post() writes them into the same slots.
// Rates are the piecewise functions of src/utils.h:100-187.
// =========================================================================================
RDC_HD double pw_Pi(double C, const double* p) {       // rectangular pulse, utils.h:100-110
  if (0.0 >= p[0]) return 0.0;
  if (C < p[1]) return 0.0;
  else if (C < p[2]) return p[0];
  return 0.0;
}
// the ramps divide by a constant width: iw = magnitude / width is formed once on the host (<= 1 ulp off the quotient)
RDC_HD double pw_SD(double C, const double* p, double iw) {       // step decay, utils.h:112-122
  if (0.0 >= p[0]) return 0.0;
  if (C < p[1]) return p[0];
  else if (C < p[2]) return (p[2] - C) * iw;
  return 0.0;
}
RDC_HD double pw_dSD(double C, const double* p, double iw) {      // utils.h:123-133
  if (0.0 >= p[0]) return 0.0;
  if (C < p[1]) return 0.0;
  else if (C < p[2]) return -iw;
  return 0.0;
}
RDC_HD double pw_Tr(double C, const double* p, double iw_up, double iw_dn) {   // trapezoid, utils.h:158-172
  if (0.0 >= p[0]) return 0.0;
  if (C < p[1]) return 0.0;
  else if (C < p[2]) return (C - p[1]) * iw_up;
  else if (C < p[3]) return p[0];
  else if (C < p[4]) return (p[4] - C) * iw_dn;
  return 0.0;
}
RDC_HD double pw_dTr(double C, const double* p, double iw_up, double iw_dn) {  // utils.h:173-187
  if (0.0 >= p[0]) return 0.0;
  if (C < p[1]) return 0.0;
  else if (C < p[2]) return iw_up;
  else if (C < p[3]) return 0.0;
  else if (C < p[4]) return -iw_dn;
  return 0.0;
}

struct AdpmK {
  double DT2;
  double decay_PrP[3], transform_A_b[5], transform_Tau[5];
  double diffuse_A_b[3], taxis1_A_b[3], taxis2_A_b[3], produce_A_b[3], decay_A_b[3];
  double diffuse_Tau[3], taxis1_Tau[3], taxis2_Tau[3], produce_Tau[3], decay_Tau[3];
  double omega_A_b, omega_Tau;  // cos(angle), :412-413
  double iw_prodA, iw_prodT, iw_trA_up, iw_trA_dn, iw_trT_up, iw_trT_dn;  // magnitude / ramp width
};

struct Adpm {
  static constexpr int HEX_CL_POINTS = 1;   // k_hex8_cl: quadrature points per workgroup barrier
  static constexpr bool HEX_STAGED = true;   // node-staged HEX8 row gather: 2.85 -> 2.12 ms on H(80) (tools/hex_ab.py)
  static constexpr bool HEX_REF_GRADS = true;
  static constexpr int AUX_LOCAL_NODE = -1;  // aux nodal values of every element node are read
  static constexpr int NELEM = 3;   // tract vector of the element
  static constexpr int NV = 3, NG = 4, NAUX = 0;
  static constexpr int FAST_EXP_MODE = 1;  // no power law in this model
  using K = AdpmK;
  using C = Coef<NV, NG>;
  // fields 2, 3 are not gradients of nodal data: any value >= NV makes the evaluators skip them
  RDC_HD static constexpr int grad_src(int k) { return k == 0 ? 1 : (k == 1 ? 2 : NV); }
  RDC_HD static constexpr int row_order(int x) { return x; }
  RDC_HD static constexpr bool hasA(int a, int b) { return (a == 0) || (b == 0) || (a == b); }
  RDC_HD static constexpr bool hasB(int a, int b, int k) { return a == b && a >= 1 && k >= 2; }
  RDC_HD static constexpr bool hasD(int a, int b) { return a == b && a >= 1; }
  RDC_HD static constexpr bool hasRG(int a, int k) { return (a == 1 && k != 1) || (a == 2 && k != 0); }

  static inline K derive(const rdc_adpm_params& p) {
    K k;
    k.DT2 = p.time_step / 2.0;                                                    // :365
    for (int i = 0; i < 3; i++) {
      k.decay_PrP[i] = p.decay_PrP[i];
      k.diffuse_A_b[i] = p.diffuse_A_b[i]; k.taxis1_A_b[i] = p.taxis1_A_b[i]; k.taxis2_A_b[i] = p.taxis2_A_b[i];
      k.produce_A_b[i] = p.produce_A_b[i]; k.decay_A_b[i] = p.decay_A_b[i];
      k.diffuse_Tau[i] = p.diffuse_Tau[i]; k.taxis1_Tau[i] = p.taxis1_Tau[i]; k.taxis2_Tau[i] = p.taxis2_Tau[i];
      k.produce_Tau[i] = p.produce_Tau[i]; k.decay_Tau[i] = p.decay_Tau[i];
    }
    k.decay_PrP[0] = p.decay_PrP[0] * pow(p.time, p.decay_PrP_time_exponent);     // :369-370
    for (int i = 0; i < 5; i++) { k.transform_A_b[i] = p.transform_A_b[i]; k.transform_Tau[i] = p.transform_Tau[i]; }
    k.omega_A_b = cos(p.taxis_A_b_angle);                                         // :412-413
    k.omega_Tau = cos(p.taxis_Tau_angle);
    k.iw_prodA = k.produce_A_b[0] / (k.produce_A_b[2] - k.produce_A_b[1]);
    k.iw_prodT = k.produce_Tau[0] / (k.produce_Tau[2] - k.produce_Tau[1]);
    k.iw_trA_up = k.transform_A_b[0] / (k.transform_A_b[2] - k.transform_A_b[1]);
    k.iw_trA_dn = k.transform_A_b[0] / (k.transform_A_b[4] - k.transform_A_b[3]);
    k.iw_trT_up = k.transform_Tau[0] / (k.transform_Tau[2] - k.transform_Tau[1]);
    k.iw_trT_dn = k.transform_Tau[0] / (k.transform_Tau[4] - k.transform_Tau[3]);
    return k;
  }
  static inline double exponent(const K&) { return 1.0; }

  RDC_HD static void tract_of(const double (&g)[3], const double* t, double omega, double (&o)[3]) {
    o[0] = 0.0; o[1] = 0.0; o[2] = 0.0;
    const double nrm = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);             // :473
    if (nrm != 0.0) {
      const double in = rcp(nrm);
      const double d = (g[0] * in) * t[0] + (g[1] * in) * t[1] + (g[2] * in) * t[2];  // :479-481
      if (d > +omega) { o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; }
      else if (d < -omega) { o[0] = -t[0]; o[1] = -t[1]; o[2] = -t[2]; }
    }
  }
  RDC_HD static void grad_post(const K& k, double (*GF)[3], const double* ed) {
    const double gA[3] = {GF[0][0], GF[0][1], GF[0][2]}, gT[3] = {GF[1][0], GF[1][1], GF[1][2]};
    double tA[3], tT[3];
    tract_of(gA, ed, k.omega_A_b, tA);
    tract_of(gT, ed, k.omega_Tau, tT);
    for (int d = 0; d < 3; d++) { GF[2][d] = tA[d]; GF[3][d] = tT[d]; }
  }

  // Only the interpolated unknowns are kept per point: the piecewise rates are a few compares each, and
  // evaluating them inside coef() (where the compiler drops what the requested equation row does not use)
  // keeps the per-point state at 3 doubles instead of 20 -- the factored TET4 kernels hold 5 points at once.
  struct Pt { double PrP, A_b, Tau; };

  template <int EXP_MODE>
  RDC_HD static void point(const K&, const double* u, const double* /*aux*/, Pt& s) {
    s.PrP = u[0]; s.A_b = u[1]; s.Tau = u[2];
  }

  RDC_HD static void coef(const K& k, const Pt& p, C& o) {
    o.zero();
    const double T = k.DT2;
    struct {
      double PrP, A_b, Tau, TrA, dTrA, TrT, dTrT, PiP, SDA, dSDA, decA, difA, t1A, t2A, SDT, dSDT, decT, difT, t1T, t2T;
    } s;
    s.PrP = p.PrP; s.A_b = p.A_b; s.Tau = p.Tau;
    s.TrA = pw_Tr(s.A_b, k.transform_A_b, k.iw_trA_up, k.iw_trA_dn); s.dTrA = pw_dTr(s.A_b, k.transform_A_b, k.iw_trA_up, k.iw_trA_dn);
    s.TrT = pw_Tr(s.Tau, k.transform_Tau, k.iw_trT_up, k.iw_trT_dn); s.dTrT = pw_dTr(s.Tau, k.transform_Tau, k.iw_trT_up, k.iw_trT_dn);
    s.PiP = pw_Pi(s.PrP, k.decay_PrP);
    s.SDA = pw_SD(s.A_b, k.produce_A_b, k.iw_prodA); s.dSDA = pw_dSD(s.A_b, k.produce_A_b, k.iw_prodA);
    s.decA = pw_Pi(s.A_b, k.decay_A_b); s.difA = pw_Pi(s.A_b, k.diffuse_A_b);
    s.t1A = pw_Pi(s.A_b, k.taxis1_A_b); s.t2A = pw_Pi(s.Tau, k.taxis2_A_b);      // taxis_2 of A_b is gated by Tau, :516
    s.SDT = pw_SD(s.Tau, k.produce_Tau, k.iw_prodT); s.dSDT = pw_dSD(s.Tau, k.produce_Tau, k.iw_prodT);
    s.decT = pw_Pi(s.Tau, k.decay_Tau); s.difT = pw_Pi(s.Tau, k.diffuse_Tau);
    s.t1T = pw_Pi(s.Tau, k.taxis1_Tau); s.t2T = pw_Pi(s.A_b, k.taxis2_Tau);      // :530
    // PrP, :497-504 and :537-556
    o.R[0] = s.PrP + T * (-s.TrA * s.PrP - s.TrT * s.PrP - s.PiP * s.PrP);
    o.A[0][0] = 1.0 - T * (-s.TrA - s.TrT - s.PiP);
    o.A[0][1] = -T * (-s.dTrA * s.PrP);
    o.A[0][2] = -T * (-s.dTrT * s.PrP);
    // A_b, :506-518 and :558-577
    o.R[1] = s.A_b + T * (s.SDA * s.A_b + s.TrA * s.PrP - s.decA * s.A_b);
    o.RG[1][0] = -T * s.difA;
    o.RG[1][2] = -T * s.t1A * s.A_b;
    o.RG[1][3] = T * s.t2A * s.A_b;
    o.A[1][0] = -T * s.TrA;
    o.A[1][1] = 1.0 - T * (s.SDA + s.dSDA * s.A_b + s.dTrA * s.PrP - s.decA);
    o.D[1][1] = T * s.difA;
    o.B[1][1][2] = T * s.t1A;
    o.B[1][1][3] = -T * s.t2A;
    // Tau, :520-532 and :579-598
    o.R[2] = s.Tau + T * (s.SDT * s.Tau + s.TrT * s.PrP - s.decT * s.Tau);
    o.RG[2][1] = -T * s.difT;
    o.RG[2][3] = -T * s.t1T * s.Tau;
    o.RG[2][2] = T * s.t2T * s.Tau;
    o.A[2][0] = -T * s.TrT;
    o.A[2][2] = 1.0 - T * (s.SDT + s.dSDT * s.Tau + s.dTrT * s.PrP - s.decT);
    o.D[2][2] = T * s.difT;
    o.B[2][2][3] = T * s.t1T;
    o.B[2][2][2] = -T * s.t2T;
  }
};

// ADPM with only the decay terms on: transform, production, A_b decay, diffusion and both taxis magnitudes <= 0 (every
// such piecewise rate returns 0, src/utils.h:100-187) -- the shipped run/HCP102513/input.dat as the code reads it.
// Left: the PrP decay pulse and the Tau decay, i.e. three diagonal mass-type blocks; no gradient field, no tract
// vector.  Exact for finite states; the host selects it from the parameter values.
struct AdpmDecayOnly : Adpm {
  static constexpr int NELEM = 0;  // the tract vectors are not read
  RDC_HD static void grad_post(const AdpmK&, double (*)[3], const double*) {}
  static inline bool applies(const rdc_adpm_params& p) {
    return p.transform_A_b[0] <= 0.0 && p.transform_Tau[0] <= 0.0 && p.produce_A_b[0] <= 0.0 && p.produce_Tau[0] <= 0.0 &&
           p.decay_A_b[0] <= 0.0 && p.diffuse_A_b[0] <= 0.0 && p.diffuse_Tau[0] <= 0.0 && p.taxis1_A_b[0] <= 0.0 &&
           p.taxis2_A_b[0] <= 0.0 && p.taxis1_Tau[0] <= 0.0 && p.taxis2_Tau[0] <= 0.0;
  }
  RDC_HD static constexpr bool hasA(int a, int b) { return a == b; }
  RDC_HD static constexpr bool hasB(int, int, int) { return false; }
  RDC_HD static constexpr bool hasD(int, int) { return false; }
  RDC_HD static constexpr bool hasRG(int, int) { return false; }
  RDC_HD static void coef(const K& k, const Pt& p, C& o) {
    o.zero();
    const double T = k.DT2;
    const double PiP = pw_Pi(p.PrP, k.decay_PrP), decT = pw_Pi(p.Tau, k.decay_Tau);
    o.R[0] = p.PrP + T * (-PiP * p.PrP);          // :497-504 with TrA = TrT = 0
    o.A[0][0] = 1.0 - T * (-PiP);
    o.R[1] = p.A_b;                               // :506-518 with every rate 0
    o.A[1][1] = 1.0;
    o.R[2] = p.Tau + T * (-decT * p.Tau);         // :520-532
    o.A[2][2] = 1.0 - T * (-decT);
  }
};

// =========================================================================================
// PROTEAS (src/proteas.C:338-705): unknowns (hos, tum, nec, vsc, oed); gradient fields 0 = grad tum,
// 1 = grad hos, 2 = grad oed.  Quirks kept: the dose RTD at a point is phi_1 * (AUX variable 0 at local node 1)
// (:481: index [0][1] of the AUX dof table) -- expressed here as the interpolation of an aux nodal field that is
// masked to local node 1 (AUX_LOCAL_NODE); dKappa = -1/T_max also where Kappa is clamped (:494); the
// d(vsc_prol)/d(tum) column omits rho_v*Kappa*vsc (:661-670).
// =========================================================================================
struct ProteasK {
  double DT2, T_max, i_RT_max;
  double rho_h, u_h, delta_h, a_RT_h, b_RT_h, nu_h;
  double D_c, D_c_h, rho_c, u_c, delta_c, a_RT_c, b_RT_c, nu_c;
  double psi_n, k_n, u_n, rho_v, nu_v;
  double D_e, rho_e, u_e, xi_e, p_RT_e, psi_e;
};

struct Proteas {
  static constexpr int HEX_CL_POINTS = 1;   // k_hex8_cl: quadrature points per workgroup barrier
  static constexpr bool HEX_STAGED = false;  // node-staged HEX8 row gather measured slower for this model (tools/hex_ab.py)
  static constexpr bool HEX_REF_GRADS = true;
  static constexpr int AUX_LOCAL_NODE = 1;
  static constexpr int NELEM = 0;
  RDC_HD static void grad_post(const ProteasK&, double (*)[3], const double*) {}
  static constexpr int NV = 5, NG = 3, NAUX = 3;
  static constexpr int FAST_EXP_MODE = 1;  // oedema/RT_exp = 1 (the default): (RTD/RT_max)^1
  using K = ProteasK;
  using C = Coef<NV, NG>;
  RDC_HD static constexpr int grad_src(int k) { return k == 0 ? 1 : (k == 1 ? 0 : 4); }
  RDC_HD static constexpr int row_order(int x) { return x; }
  RDC_HD static constexpr bool hasA(int a, int b) { return a < 4 ? b < 4 : (b == 1 || b == 4); }
  RDC_HD static constexpr bool hasB(int a, int b, int k) { return a == 1 && b < 4 && (k == 0 || (k == 1 && b != 1)); }
  RDC_HD static constexpr bool hasD(int a, int b) { return (a == 1 && b <= 1) || (a == 4 && b == 4); }
  RDC_HD static constexpr bool hasRG(int a, int k) { return (a == 1 && k <= 1) || (a == 4 && k == 2); }

  static inline K derive(const rdc_proteas_params& p) {
    K k;
    k.DT2 = p.time_step / 2.0;
    k.T_max = p.cells_total_capacity; k.i_RT_max = 1.0 / p.RT_max_dosage;
    k.rho_h = p.host_proliferation; k.u_h = p.host_vsc_threshold; k.delta_h = p.host_RT_death_rate;
    k.a_RT_h = p.host_RT_exp_a; k.b_RT_h = p.host_RT_exp_b; k.nu_h = p.host_necrosis_rate;
    k.D_c = p.tumour_diffusion; k.D_c_h = p.tumour_diffusion_host; k.rho_c = p.tumour_proliferation;
    k.u_c = p.tumour_vsc_threshold; k.delta_c = p.tumour_RT_death_rate; k.a_RT_c = p.tumour_RT_exp_a;
    k.b_RT_c = p.tumour_RT_exp_b; k.nu_c = p.tumour_necrosis_rate;
    k.psi_n = p.necrosis_clearance; k.k_n = p.necrosis_slope; k.u_n = p.necrosis_vsc_threshold;
    k.rho_v = p.vascular_proliferation; k.nu_v = p.vascular_necrosis_rate;
    k.D_e = p.oedema_diffusion; k.rho_e = p.oedema_proliferation; k.u_e = p.oedema_vsc_threshold;
    k.xi_e = p.oedema_RT_coeff; k.p_RT_e = p.oedema_RT_exp; k.psi_e = p.oedema_reabsorption_rate;
    return k;
  }
  static inline double exponent(const K& k) { return k.p_RT_e; }

  // per point: the unknowns and the transcendental results only (see Ripf::Pt)
  struct Pt {
    double hos, tum, nec, vsc, oed;
    double hRT, tRT, nclr, dnclr, oRT;
  };

  template <int EXP_MODE>
  RDC_HD static void point(const K& k, const double* u, const double* aux, Pt& s) {
    s.hos = u[0]; s.tum = u[1]; s.nec = u[2]; s.vsc = u[3]; s.oed = u[4];
    const double RTD = aux[0];                                               // masked interpolation, see above
    s.hRT = k.delta_h * (1.0 - exp(-k.a_RT_h * RTD - k.b_RT_h * (RTD * RTD)));  // :497
    s.tRT = k.delta_c * (1.0 - exp(-k.a_RT_c * RTD - k.b_RT_c * (RTD * RTD)));  // :502
    const double arg = k.k_n * s.vsc - k.u_n;
    s.nclr = k.psi_n * (1.0 - tanh(arg));                                    // :506
    const double ch = cosh(arg);
    s.dnclr = k.psi_n * -k.k_n / (ch * ch);                                  // :507
    const double x = RTD * k.i_RT_max;
    s.oRT = k.xi_e * (EXP_MODE == 1 ? x : rdc_pow(x, k.p_RT_e));             // :515
  }

  RDC_HD static void coef(const K& k, const Pt& p, C& o) {
    o.zero();
    const double T = k.DT2, dK = -1.0 / k.T_max;                             // :493
    struct { double hos, tum, nec, vsc, oed, Kap, hp, dhp, hRT, tp, dtp, tRT, nclr, dnclr, dvp, oRT, oclr; } s;
    s.hos = p.hos; s.tum = p.tum; s.nec = p.nec; s.vsc = p.vsc; s.oed = p.oed;
    s.hRT = p.hRT; s.tRT = p.tRT; s.nclr = p.nclr; s.dnclr = p.dnclr; s.oRT = p.oRT;
    {
      const double Tt = s.hos + s.tum + s.nec + s.vsc;                       // :490
      double Kap = 1.0 - Tt / k.T_max;                                       // :491-492
      Kap = (Kap > 0.0 ? Kap : 0.0);
      Kap = (Kap < 1.0 ? Kap : 1.0);
      s.Kap = Kap;
      const double Hh = (s.vsc - k.u_h > 0.0 ? 1.0 : 0.0), Hc = (s.vsc - k.u_c > 0.0 ? 1.0 : 0.0);
      s.hp = k.rho_h * Kap * Hh; s.dhp = k.rho_h * dK * Hh;                  // :495-496
      s.tp = k.rho_c * Kap * Hc; s.dtp = k.rho_c * dK * Hc;                  // :500-501
      s.dvp = k.rho_v * dK * s.tum;                                          // :510
      s.oclr = k.psi_e * (1.0 - (s.vsc - k.u_e > 0.0 ? 1.0 : 0.0));          // :516
    }
    const double hn = k.nu_h * s.nec, tn = k.nu_c * s.nec;                   // :498,503
    const double np_ = k.nu_h * s.hos + k.nu_c * s.tum + k.nu_v * s.vsc;     // :505
    const double vp = k.rho_v * s.Kap * s.tum, vn = k.nu_v * s.nec;          // :509,511
    const double op = k.rho_e * s.tum * (1.0 - s.tum), dop = k.rho_e * (1.0 - 2.0 * s.tum);  // :513-514
    const double hh = s.hos * (1.0 - s.hos);
    // host, :522-529 and :575-601
    o.R[0] = s.hos + T * (s.hp * hh - s.hRT * s.hos - hn * s.hos);
    o.A[0][0] = 1.0 - T * (s.dhp * hh + s.hp * (1.0 - 2.0 * s.hos) - s.hRT - hn);
    o.A[0][1] = -T * (s.dhp * hh);
    o.A[0][2] = -T * (s.dhp * hh - k.nu_h * s.hos);
    o.A[0][3] = -T * (s.dhp * hh);
    // tumour, :531-541 and :603-640
    o.R[1] = s.tum + T * (s.tp * s.tum - s.tRT * s.tum - tn * s.tum);
    o.RG[1][0] = -T * k.D_c * s.Kap;
    o.RG[1][1] = -T * k.D_c_h * s.Kap * s.tum;
    const double bt = T * k.D_c * dK, bh = T * k.D_c_h * dK * s.tum;
    o.A[1][0] = -T * (s.dtp * s.tum);
    o.B[1][0][0] = bt; o.B[1][0][1] = bh;
    o.D[1][0] = T * k.D_c_h * s.Kap * s.tum;
    o.A[1][1] = 1.0 - T * (s.dtp * s.tum + s.tp - s.tRT - tn);
    o.B[1][1][0] = bt;
    o.D[1][1] = T * k.D_c * s.Kap;
    o.A[1][2] = -T * (s.dtp * s.tum - k.nu_c * s.tum);
    o.B[1][2][0] = bt; o.B[1][2][1] = bh;
    o.A[1][3] = -T * (s.dtp * s.tum);
    o.B[1][3][0] = bt; o.B[1][3][1] = bh;
    // necrotic, :543-550 and :642-665
    o.R[2] = s.nec + T * (np_ * s.nec - s.nclr * s.nec);
    o.A[2][0] = -T * (k.nu_h * s.nec);
    o.A[2][1] = -T * (k.nu_c * s.nec);
    o.A[2][2] = 1.0 - T * (np_ - s.nclr);
    o.A[2][3] = -T * (k.nu_v * s.nec - s.dnclr * s.nec);
    // vascular, :552-559 and :667-692
    o.R[3] = s.vsc + T * (vp * s.vsc - vn * s.vsc);
    o.A[3][0] = -T * (s.dvp * s.vsc);
    o.A[3][1] = -T * (s.dvp * s.vsc);
    o.A[3][2] = -T * (s.dvp * s.vsc - k.nu_v * s.vsc);
    o.A[3][3] = 1.0 - T * (s.dvp * s.vsc + vp - vn);
    // oedema, :561-570 and :694-709
    o.R[4] = s.oed + T * (op * s.oed - s.oRT * s.oed - s.oclr * s.oed);
    o.RG[4][2] = -T * k.D_e;
    o.A[4][1] = -T * (dop * s.oed);
    o.A[4][4] = 1.0 - T * (op - s.oRT - s.oclr);
    o.D[4][4] = T * k.D_e;
  }
};

static inline int exp_mode_of(double e) {
  if (e == 1.0) return 1;
  if (e == 2.0) return 2;
  if (e == 3.0) return 3;
  if (e == 4.0) return 4;
  if (e == 2.5) return 25;
  return 0;
}

}  // namespace rdc
#endif
