// rdc_hex8_cl.h — HEX8 reaction-diffusion assembly on node clusters with producer and consumer waves (three-unknown
// models: coupled HCC src/coupled_hcc.C:433-645, RIPF src/ripf.C:382-553, ADPM src/adpm.C:370-528).
//
// k_rowgather_staged repeats everything that belongs to an (element, point) -- Jacobian, inverse, interpolation of the
// unknowns and their gradients, pow() -- in each of the element's eight (node, element) pairs, and re-reads the eight
// node records from LDS at every point.  Here, as in rdc_solid_cl.hip:
//   * a workgroup owns a cluster of <= CW * 8 owned nodes (rdc_prep_cl.cpp);
//   * PW producer waves: one lane per element touching the cluster, whose coordinates and unknowns stay in its registers;
//     per quadrature point it evaluates the POINT RECORD once: S = Ji Ji^T (6), JxW, h_g = Ji grad f_g for the model's
//     gradient fields (3 each), and the interpolated unknowns / aux fields, through LDS;
//   * CW consumer waves: one lane per pair; per point it reads the record, forms what depends on its row node
//     (b_i = S dN_i, grad phi_j . grad phi_i = dN_j . b_i, grad f_g . grad phi_i = h_g . dN_i), evaluates M::point and
//     M::coef (cheap next to the geometry; on the producer they made its instruction stream the longest of the
//     workgroup) and accumulates with the generic evaluator's own accumulation (rd_point_accum): NV^2 x 8 accumulators;
//   * epilogue: rows added into an LDS image of the cluster's CSR rows (ds_add_f64), image copied out in runs.
// The record functions are host + device so that the CPU suite replays the kernel from the same lists (tests/host_shim.cpp).
#ifndef RDC_HEX8_CL_H
#define RDC_HEX8_CL_H
#include "rdc_row.h"

namespace rdc {

template <class M>
struct Hex8Rec {
  static constexpr int NA = (M::NAUX > 0 ? M::NAUX : 0);
  static constexpr int HG = 7, UQ = 7 + 3 * M::NG, AQ = UQ + M::NV, N = AQ + NA;
  static constexpr int STRIDE = N | 1;   // odd: the records of 32 consecutive elements start in 32 different double-banks
};

// point record of element (X, U, AX, ED) at quadrature point q (follows rd_point_setup, HEX8 reference-gradient branch)
template <class M>
RDC_HD void hex8_cl_produce(const typename M::K& k, const double (&X)[8][3], const double (&U)[8][M::NV],
                            const double (&AX)[8][M::NAUX > 0 ? M::NAUX : 1], const double* ED, int q, double* rec) {
  constexpr int NV = M::NV, NG = M::NG, NA = (M::NAUX > 0 ? M::NAUX : 1);
  using R = Hex8Rec<M>;
  double Ji[3][3], W;
  fe_jacobian8(X, q, Ji, W);
  const int V[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {1, 2}, {0, 2}};
#pragma unroll
  for (int p = 0; p < 6; p++) {
    const int a = V[p][0], b = V[p][1];
    rec[p] = Ji[a][0] * Ji[b][0] + Ji[a][1] * Ji[b][1] + Ji[a][2] * Ji[b][2];
  }
  rec[6] = W;
  double GF[NG][3];
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
  double uq[NV], aq[NA];
#pragma unroll
  for (int v = 0; v < NV; v++) {
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < 8; l++) s += kHex8Tab.N[q][l] * U[l][v];
    uq[v] = s;
  }
#pragma unroll
  for (int v = 0; v < NA; v++) {
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < 8; l++) s += kHex8Tab.N[q][l] * AX[l][v];
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
#pragma unroll
  for (int g = 0; g < NG; g++)
#pragma unroll
    for (int c = 0; c < 3; c++) rec[R::HG + 3 * g + c] = Ji[c][0] * GF[g][0] + Ji[c][1] * GF[g][1] + Ji[c][2] * GF[g][2];
#pragma unroll
  for (int v = 0; v < NV; v++) rec[R::UQ + v] = uq[v];
#pragma unroll
  for (int v = 0; v < R::NA; v++) rec[R::AQ + v] = aq[v];
}

// what the point whose record is `rec` contributes to the row of local node irow, before the accumulation
template <class M, int EXP_MODE>
RDC_HD void hex8_cl_point(const typename M::K& k, const double* rec, int q, int irow, RowPoint<M, 8>& P) {
  constexpr int NG = M::NG;
  using R = Hex8Rec<M>;
  P.W = rec[6];
#pragma unroll
  for (int n = 0; n < 8; n++) P.N[n] = kHex8Tab.N[q][n];
  // the row node's reference shape data from its corner signs (irow differs per lane)
  double xi[3], wq;
  Ref<8>::qpoint(q, xi, wq);
  const double sx = Ref<8>::sx(irow), sy = Ref<8>::sy(irow), sz = Ref<8>::sz(irow);
  const double fa = 1.0 + sx * xi[0], fb = 1.0 + sy * xi[1], fc = 1.0 + sz * xi[2];
  P.Ni = 0.125 * fa * fb * fc;
  const double dNi[3] = {0.125 * sx * fb * fc, 0.125 * fa * sy * fc, 0.125 * fa * fb * sz};
  const double bi[3] = {rec[0] * dNi[0] + rec[3] * dNi[1] + rec[5] * dNi[2],
                        rec[3] * dNi[0] + rec[1] * dNi[1] + rec[4] * dNi[2],
                        rec[5] * dNi[0] + rec[4] * dNi[1] + rec[2] * dNi[2]};
#pragma unroll
  for (int j = 0; j < 8; j++)
    P.dd[j] = kHex8Tab.dN[q][j][0] * bi[0] + kHex8Tab.dN[q][j][1] * bi[1] + kHex8Tab.dN[q][j][2] * bi[2];
#pragma unroll
  for (int g = 0; g < NG; g++)
    P.gi[g] = rec[R::HG + 3 * g] * dNi[0] + rec[R::HG + 3 * g + 1] * dNi[1] + rec[R::HG + 3 * g + 2] * dNi[2];
  double uq[M::NV], aq[M::NAUX > 0 ? M::NAUX : 1] = {0.0};
#pragma unroll
  for (int v = 0; v < M::NV; v++) uq[v] = rec[R::UQ + v];
#pragma unroll
  for (int v = 0; v < R::NA; v++) aq[v] = rec[R::AQ + v];
  typename M::Pt pt;
  M::template point<EXP_MODE>(k, uq, aq, pt);
  M::coef(k, pt, P.c);
}

// contribution of that point to the row of local node irow: all equation rows (three unknowns) ...
template <class M, int EXP_MODE>
RDC_HD void hex8_cl_consume(const typename M::K& k, const double* rec, int q, int irow, double (&acc)[M::NV][M::NV][8],
                            double (&fe)[M::NV]) {
  RowPoint<M, 8> P;
  hex8_cl_point<M, EXP_MODE>(k, rec, q, irow, P);
  rd_point_accum<M, 8>(P, acc, fe);
}
// ... or equation row A only (five unknowns: the NV x NV x 8 accumulator does not fit the register file; everything of
// coef() that row A does not read is dead code)
template <class M, int EXP_MODE, int A>
RDC_HD void hex8_cl_consume_row(const typename M::K& k, const double* rec, int q, int irow, double (&acc)[M::NV][8], double& fe) {
  RowPoint<M, 8> P;
  hex8_cl_point<M, EXP_MODE>(k, rec, q, irow, P);
  rd_point_accum_row<M, 8, A>(P, acc, fe);
}

}  // namespace rdc
#endif
