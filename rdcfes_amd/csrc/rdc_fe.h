// rdc_fe.h — FIRST LAGRANGE TET4 / HEX8 shape data and QGauss(THIRD) rules on the device.
// libMesh facts restated: TET4 5-point rule with the negative centroid
// weight; HEX8 2x2x2 tensor Gauss, x fastest; node orders coincide with Gmsh.
#ifndef RDC_FE_H
#define RDC_FE_H

#include "rdc_integrands.h"

namespace rdc {

template <int NEN> struct Ref;

template <> struct Ref<4> {
  static constexpr int NQP = 5;
  RDC_HD static void qpoint(int q, double xi[3], double& w) {
    const double s = 1.0 / 6.0;
    xi[0] = (q == 0) ? 0.25 : (q == 1 ? 0.5 : s);
    xi[1] = (q == 0) ? 0.25 : (q == 2 ? 0.5 : s);
    xi[2] = (q == 0) ? 0.25 : (q == 3 ? 0.5 : s);
    w = (q == 0) ? (-2.0 / 15.0) : 0.075;
  }
  RDC_HD static void shape(const double xi[3], double N[4], double dN[4][3]) {
    N[0] = 1.0 - xi[0] - xi[1] - xi[2]; N[1] = xi[0]; N[2] = xi[1]; N[3] = xi[2];
    dN[0][0] = -1.0; dN[0][1] = -1.0; dN[0][2] = -1.0;
    dN[1][0] = 1.0; dN[1][1] = 0.0; dN[1][2] = 0.0;
    dN[2][0] = 0.0; dN[2][1] = 1.0; dN[2][2] = 0.0;
    dN[3][0] = 0.0; dN[3][1] = 0.0; dN[3][2] = 1.0;
  }
};

template <> struct Ref<8> {
  static constexpr int NQP = 8;
  RDC_HD static double sx(int n) { return ((n & 3) == 1 || (n & 3) == 2) ? 1.0 : -1.0; }
  RDC_HD static double sy(int n) { return ((n & 3) >= 2) ? 1.0 : -1.0; }
  RDC_HD static double sz(int n) { return (n >= 4) ? 1.0 : -1.0; }
  RDC_HD static void qpoint(int q, double xi[3], double& w) {
    const double g = 0.57735026918962576451;
    xi[0] = (q & 1) ? g : -g;
    xi[1] = (q & 2) ? g : -g;
    xi[2] = (q & 4) ? g : -g;
    w = 1.0;
  }
  RDC_HD static void shape(const double xi[3], double N[8], double dN[8][3]) {
#pragma unroll
    for (int n = 0; n < 8; n++) {
      const double a = 1.0 + sx(n) * xi[0], b = 1.0 + sy(n) * xi[1], c = 1.0 + sz(n) * xi[2];
      N[n] = 0.125 * a * b * c;
      dN[n][0] = 0.125 * sx(n) * b * c;
      dN[n][1] = 0.125 * a * sy(n) * c;
      dN[n][2] = 0.125 * a * b * sz(n);
    }
  }
  // the same values at the 8 Gauss points as compile-time data: with the quadrature loop kept rolled the point index is
  // wave-uniform, so these become scalar loads / scalar operands instead of ~100 FP64 operations and 64 VGPRs per point
  struct Tab { double N[8][8], dN[8][8][3]; };
  static constexpr Tab make_tab() {
    Tab t{};
    const double g = 0.57735026918962576451;
    for (int q = 0; q < 8; q++) {
      const double x = (q & 1) ? g : -g, y = (q & 2) ? g : -g, z = (q & 4) ? g : -g;
      for (int n = 0; n < 8; n++) {
        const double s0 = ((n & 3) == 1 || (n & 3) == 2) ? 1.0 : -1.0, s1 = ((n & 3) >= 2) ? 1.0 : -1.0, s2 = (n >= 4) ? 1.0 : -1.0;
        const double a = 1.0 + s0 * x, b = 1.0 + s1 * y, c = 1.0 + s2 * z;
        t.N[q][n] = 0.125 * a * b * c;
        t.dN[q][n][0] = 0.125 * s0 * b * c;
        t.dN[q][n][1] = 0.125 * a * s1 * c;
        t.dN[q][n][2] = 0.125 * a * b * s2;
      }
    }
    return t;
  }
};
static constexpr Ref<8>::Tab kHex8Tab = Ref<8>::make_tab();
__attribute__((unused)) static const void* const kHex8TabRef = &kHex8Tab;

// physical shape data at one quadrature point: phi, grad phi, JxW = det(J) * w
template <int NEN>
RDC_HD void fe_point(const double (&X)[NEN][3], int q, double (&N)[NEN], double (&G)[NEN][3], double& JxW) {
  double xi[3], w, dN[NEN][3];
  Ref<NEN>::qpoint(q, xi, w);
  if constexpr (NEN == 8) {
#pragma unroll
    for (int n = 0; n < 8; n++) {
      N[n] = kHex8Tab.N[q][n];
      dN[n][0] = kHex8Tab.dN[q][n][0]; dN[n][1] = kHex8Tab.dN[q][n][1]; dN[n][2] = kHex8Tab.dN[q][n][2];
    }
  } else {
    Ref<NEN>::shape(xi, N, dN);
  }
  double J[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double s = 0.0;
#pragma unroll
      for (int n = 0; n < NEN; n++) s += X[n][r] * dN[n][c];
      J[r][c] = s;
    }
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
  const double s = rcp(det);  // v_rcp_f64 + 2 Newton steps: <= 1 ulp
  double Ji[3][3];  // Ji[c][r] = d xi_c / d x_r
  Ji[0][0] = c00 * s;
  Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * s;
  Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * s;
  Ji[1][0] = c01 * s;
  Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * s;
  Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * s;
  Ji[2][0] = c02 * s;
  Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * s;
  Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * s;
#pragma unroll
  for (int n = 0; n < NEN; n++)
#pragma unroll
    for (int r = 0; r < 3; r++) G[n][r] = dN[n][0] * Ji[0][r] + dN[n][1] * Ji[1][r] + dN[n][2] * Ji[2][r];
  JxW = det * w;
}

// HEX8: inverse Jacobian and JxW only.  The generic row evaluator needs the physical gradient of ONE node (the row
// node) and of the interpolated fields, so it applies Ji to those few reference gradients instead of forming
// grad phi of all eight nodes (72 multiply-adds per point, rdc_row.h).
RDC_HD void fe_jacobian8(const double (&X)[8][3], int q, double (&Ji)[3][3], double& JxW) {
  double J[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double s = 0.0;
#pragma unroll
      for (int n = 0; n < 8; n++) s += X[n][r] * kHex8Tab.dN[q][n][c];
      J[r][c] = s;
    }
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
  const double s = rcp(det);  // v_rcp_f64 + 2 Newton steps: <= 1 ulp
  Ji[0][0] = c00 * s;
  Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * s;
  Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * s;
  Ji[1][0] = c01 * s;
  Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * s;
  Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * s;
  Ji[2][0] = c02 * s;
  Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * s;
  Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * s;
  JxW = det;  // Gauss weight 1
}

}  // namespace rdc
#endif
