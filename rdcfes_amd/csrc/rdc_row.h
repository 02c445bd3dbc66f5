// rdc_row.h — evaluation of ONE row-node of the element matrices of a reaction-diffusion model
// over all quadrature points (generic in model and element type).  Host+device so that the
// CPU-only test-suite can exercise exactly the code the kernels run (tests/host_shim.cpp);
// the shipped library only ever instantiates it inside HIP kernels.
#ifndef RDC_ROW_H
#define RDC_ROW_H
#include "rdc_fe.h"

namespace rdc {

// ---- one row (local node `irow`) of Ke and Fe over all quadrature points ------------------
template <class M, int NEN, int EXP_MODE>
RDC_HD void rd_row(const typename M::K& k, const double (&X)[NEN][3],
                                       const double (&U)[NEN][M::NV],
                                       const double (&AX)[NEN][M::NAUX > 0 ? M::NAUX : 1], int irow,
                                       double (&acc)[M::NV][M::NV][NEN], double (&fe)[M::NV],
                                       const double* ED = nullptr /* M::NELEM per-element inputs */) {
  constexpr int NV = M::NV, NG = M::NG, NA = (M::NAUX > 0 ? M::NAUX : 1);
#pragma unroll
  for (int a = 0; a < NV; a++) {
    fe[a] = 0.0;
#pragma unroll
    for (int b = 0; b < NV; b++)
#pragma unroll
      for (int j = 0; j < NEN; j++) acc[a][b][j] = 0.0;
  }
#pragma unroll 1
  for (int q = 0; q < Ref<NEN>::NQP; q++) {
    double N[NEN], G[NEN][3], W;
    fe_point<NEN>(X, q, N, G, W);
    // old solution, aux fields and gradient fields at the point (src/pihna.C:429-442)
    double uq[NV], aq[NA], GF[NG][3];
#pragma unroll
    for (int v = 0; v < NV; v++) {
      double s = 0.0;
#pragma unroll
      for (int l = 0; l < NEN; l++) s += N[l] * U[l][v];
      uq[v] = s;
    }
#pragma unroll
    for (int v = 0; v < NA; v++) {
      double s = 0.0;
#pragma unroll
      for (int l = 0; l < NEN; l++) s += N[l] * AX[l][v];
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
        for (int l = 0; l < NEN; l++) s += G[l][d] * (src >= 0 ? U[l][(src >= 0 && src < NV) ? src : 0] : AX[l][src < 0 ? (-1 - src) % NA : 0]);
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
    typename M::C c;
    M::coef(k, pt, c);
    // shape data of the row node
    double Ni = 0.0, Gi[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int n = 0; n < NEN; n++)
      if (n == irow) { Ni = N[n]; Gi[0] = G[n][0]; Gi[1] = G[n][1]; Gi[2] = G[n][2]; }
    double gi[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) gi[g] = GF[g][0] * Gi[0] + GF[g][1] * Gi[1] + GF[g][2] * Gi[2];
#pragma unroll
    for (int a = 0; a < NV; a++) {
      double r = c.R[a] * Ni;
#pragma unroll
      for (int g = 0; g < NG; g++) r += c.RG[a][g] * gi[g];
      fe[a] += W * r;
    }
#pragma unroll
    for (int j = 0; j < NEN; j++) {
      const double pp = N[j] * Ni;
      const double dd = G[j][0] * Gi[0] + G[j][1] * Gi[1] + G[j][2] * Gi[2];
#pragma unroll
      for (int a = 0; a < NV; a++)
#pragma unroll
        for (int b = 0; b < NV; b++) {
          double bg = 0.0;
#pragma unroll
          for (int g = 0; g < NG; g++) bg += c.B[a][b][g] * gi[g];
          acc[a][b][j] += W * (c.A[a][b] * pp + N[j] * bg + c.D[a][b] * dd);
        }
    }
  }
}

}  // namespace rdc
#endif
