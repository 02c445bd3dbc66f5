// host_shim.cpp — TEST-ONLY host build of the product's host/device headers, so the CPU suite
// (-m "not gpu") can check the exact row-evaluation code the HIP kernels run, and the mesh
// preparation, against the oracle without a GPU.  It is compiled by tests/conftest.py with g++
// into tests/_build/ and is never part of librdc_assembly.so (the product has no CPU path).
#include <cstring>
#include <string>

#include "../rdcfes_amd/csrc/rdc_prep.h"
#include "../rdcfes_amd/csrc/rdc_row.h"
#include "../rdcfes_amd/csrc/rdc_tet4_fast.h"
#include "../rdcfes_amd/csrc/rdc_tet4_pihna_moments.h"
#include "../rdcfes_amd/csrc/rdc_tet4_ev.h"
#include "../rdcfes_amd/csrc/rdc_hex8_cl.h"
#include "../rdcfes_amd/csrc/rdc_tet4_evc.h"

using namespace rdc;

namespace {

template <class M, int NEN, int EM>
void row_generic(const typename M::K& k, const double* Xp, const double* Up, const double* Ap, int irow, double* acc,
                 double* fe, const double* ED) {
  constexpr int NV = M::NV, NA = (M::NAUX > 0 ? M::NAUX : 1);
  double X[NEN][3], U[NEN][NV], AX[NEN][NA];
  for (int i = 0; i < NEN; i++) {
    for (int d = 0; d < 3; d++) X[i][d] = Xp[3 * i + d];
    for (int v = 0; v < NV; v++) U[i][v] = Up[NV * i + v];
    for (int v = 0; v < NA; v++)
      AX[i][v] = (M::NAUX > 0 && Ap && (M::AUX_LOCAL_NODE < 0 || i == M::AUX_LOCAL_NODE)) ? Ap[M::NAUX * i + v] : 0.0;
  }
  double a[NV][NV][NEN], f[NV];
  rd_row<M, NEN, EM>(k, X, U, AX, irow, a, f, ED);
  std::memcpy(acc, a, sizeof(a));
  std::memcpy(fe, f, sizeof(f));
}

template <class M>
struct ArraySink {
  double* acc;  // [NV][NV][4] in ORIGINAL column order
  double* fe;
  int irow;
  void ke(int a, int b, int j, double v) { acc[(a * M::NV + b) * 4 + (j ^ irow)] = v; }
  void fe_(int a, double v) { fe[a] = v; }
};
template <class M>
struct ArraySinkAdapter {
  ArraySink<M> s;
  void ke(int a, int b, int j, double v) { s.ke(a, b, j, v); }
  void fe(int a, double v) { s.fe_(a, v); }
};

template <class M, int EM>
void row_fast(const typename M::K& k, const double* Xp, const double* Up, const double* Ap, int irow, double* acc,
              double* fe, const double* ED = nullptr) {
  constexpr int NV = M::NV, NA = (M::NAUX > 0 ? M::NAUX : 1);
  double X[4][3], U[4][NV], AX[4][NA];
  for (int j = 0; j < 4; j++) {  // rotate the row node to local 0: j -> j ^ irow (rdc_tet4_fast.h)
    const int jo = j ^ irow;
    for (int d = 0; d < 3; d++) X[j][d] = Xp[3 * jo + d];
    for (int v = 0; v < NV; v++) U[j][v] = Up[NV * jo + v];
    for (int v = 0; v < NA; v++)
      AX[j][v] = (M::NAUX > 0 && Ap && (M::AUX_LOCAL_NODE < 0 || jo == M::AUX_LOCAL_NODE)) ? Ap[M::NAUX * jo + v] : 0.0;
  }
  ArraySinkAdapter<M> sink{{acc, fe, irow}};
  tet4_row0<M, EM>(k, X, U, AX, sink, ED);
}

template <class M, class P>
int run(const P* p, int nen, int fast, int force_general_pow, const double* X, const double* U, const double* A,
        int irow, double* acc, double* fe, const double* ED = nullptr) {
  const typename M::K k = M::derive(*p);
  // the dedicated-exponent instantiation the product would pick (Pihna/Hcc: 3, Ripf: 2.5 via sqrt)
  constexpr int FE = M::FAST_EXP_MODE;
  const bool cube = !force_general_pow && exp_mode_of(M::exponent(k)) == FE;
  if (fast) {
    if (nen != 4) return 1;
    if (cube) row_fast<M, FE>(k, X, U, A, irow, acc, fe, ED); else row_fast<M, 0>(k, X, U, A, irow, acc, fe, ED);
    return 0;
  }
  if (nen == 4) {
    if (cube) row_generic<M, 4, FE>(k, X, U, A, irow, acc, fe, ED); else row_generic<M, 4, 0>(k, X, U, A, irow, acc, fe, ED);
  } else if (nen == 8) {
    if (cube) row_generic<M, 8, FE>(k, X, U, A, irow, acc, fe, ED); else row_generic<M, 8, 0>(k, X, U, A, irow, acc, fe, ED);
  } else return 1;
  return 0;
}

template <class M, class P>
int masks(const P* p, const double* u, const double* aux, double* worst) {
  const typename M::K k = M::derive(*p);
  typename M::Pt pt;
  M::template point<0>(k, u, aux, pt);
  typename M::C c;
  M::coef(k, pt, c);
  double w = 0.0;
  for (int a = 0; a < M::NV; a++) {
    for (int g = 0; g < M::NG; g++)
      if (!M::hasRG(a, g) && c.RG[a][g] != 0.0) w = 1.0;
    for (int b = 0; b < M::NV; b++) {
      if (!M::hasA(a, b) && c.A[a][b] != 0.0) w = 1.0;
      if (!M::hasD(a, b) && c.D[a][b] != 0.0) w = 1.0;
      for (int g = 0; g < M::NG; g++)
        if (!M::hasB(a, b, g) && c.B[a][b][g] != 0.0) w = 1.0;
    }
  }
  *worst = w;
  return 0;
}

HostPrep g_prep;
HostPrepEv g_ev;
int g_ev_bg = 1;   // shim_ev_assemble: background visits skip their zero moments (rdc_tet4_ev.h, bg), as the device does per wave
HostPrepCl g_cl;
std::vector<uint32_t> g_conn;
SolidGather g_gather;
std::string g_err;

// host emulation of one workgroup's LDS accumulation in k_tet4_ev
struct EvHostSink {
  double* M;
  double* R;
  int blk[4][4], nloc[4];
  void mom(int m, int i, int j, double v) { M[m * ev::NBP + blk[i][j]] += v; }
  void rhs(int a, int i, double v) { R[a * ev::MAXN + nloc[i]] += v; }
};


// k_hex8_cl replayed on the host from the cluster lists of the last shim_cl_build and the SAME record functions
// (hex8_cl_produce / hex8_cl_consume): per cluster the point records of its elements, then per pair the row over the eight
// points, added into the CSR rows at the slots of the pair list.  val / rhs: the caller pre-fills NaN; covered entries are
// first zeroed here (every owned row belongs to exactly one cluster).
template <class M, class P>
int cl_replay(const P* p, const double* xyz, const double* u, const double* aux, const double* elem, double* val, double* rhs) {
  constexpr int NV = M::NV, NA = (M::NAUX > 0 ? M::NAUX : 1);
  using R = Hex8Rec<M>;
  if (!g_cl.ok || g_prep.nen != 8 || g_prep.nvar != NV) return 1;
  const typename M::K k = M::derive(*p);
  const bool fastexp = exp_mode_of(M::exponent(k)) == M::FAST_EXP_MODE;
  const HostPrepCl& C = g_cl;
  std::vector<double> rec;
  for (size_t w = 0; w < C.desc.size(); w++) {
    const HostPrepCl::Desc& d = C.desc[w];
    rec.assign((size_t)d.nelem * 8 * R::STRIDE, 0.0);
    for (int le = 0; le < d.nelem; le++) {
      const uint32_t e = C.eid[w * C.lim.max_elems + le];
      double X[8][3], U[8][NV], AX[8][NA];
      for (int n = 0; n < 8; n++) {
        const uint32_t I = g_conn[(size_t)e * 8 + n];
        for (int c = 0; c < 3; c++) X[n][c] = xyz[3 * (size_t)I + c];
        for (int v = 0; v < NV; v++) U[n][v] = u[NV * (size_t)I + v];
        for (int v = 0; v < NA; v++) AX[n][v] = (M::NAUX > 0 && aux) ? aux[(size_t)M::NAUX * I + (v % (M::NAUX > 0 ? M::NAUX : 1))] : 0.0;
      }
      const double* ED = M::NELEM > 0 ? elem + (size_t)e * M::NELEM : nullptr;
      for (int q = 0; q < 8; q++) {
        double* r = &rec[((size_t)le * 8 + q) * R::STRIDE];
        hex8_cl_produce<M>(k, X, U, AX, ED, q, r);
      }
    }
    for (int a = 0; a < d.nown; a++) {
      const HostPrepCl::Node& nd = C.ntab[w * C.lim.max_nodes + a];
      for (int x = 0; x < NV * NV * (int)nd.len; x++) val[(size_t)NV * NV * nd.bptr + x] = 0.0;
      for (int v = 0; v < NV; v++) rhs[(size_t)NV * nd.node + v] = 0.0;
    }
    for (int x = 0; x < d.npair; x++) {
      const uint32_t pr = C.pair[w * C.lim.max_pairs + x];
      const uint32_t le = pr & 0xFF, li = (pr >> 8) & 0xFF, a = (pr >> 16) & 0xFF;
      double acc[NV][NV][8], fe[NV];
      rd_row_zero<M, 8>(acc, fe);
      for (int q = 0; q < 8; q++) {
        const double* r = &rec[((size_t)le * 8 + q) * R::STRIDE];
        if (fastexp) hex8_cl_consume<M, M::FAST_EXP_MODE>(k, r, q, (int)li, acc, fe); else hex8_cl_consume<M, 0>(k, r, q, (int)li, acc, fe);
      }
      const HostPrepCl::Node& nd = C.ntab[w * C.lim.max_nodes + a];
      for (int j = 0; j < 8; j++) {
        const uint32_t sl = (C.pslot[(w * C.lim.max_pairs + x) * 2 + j / 4] >> (8 * (j % 4))) & 0xFF;
        for (int aa = 0; aa < NV; aa++)
          for (int b = 0; b < NV; b++) val[(size_t)NV * NV * nd.bptr + (size_t)aa * NV * nd.len + NV * sl + b] += acc[aa][b][j];
      }
      for (int v = 0; v < NV; v++) rhs[(size_t)NV * nd.node + v] += fe[v];
    }
  }
  return 0;
}


// k_tet4_evc replayed on the host from the element-visit lists of the last shim_ev_build and the SAME visit function
// (tet4_visit): entries accumulated per cluster in the slice layout of the kernel ([entry][slot][node]), then moved to the
// CSR rows.  val / rhs: the caller pre-fills NaN; entries the lists do not cover stay NaN.
template <class M>
struct EvcHostSink {
  double* S;        // [NE][NBP]
  double* R;        // [NV][MAXN]
  int blk[4][4], nloc[4];
  void ke(int a, int b, int i, int j, double v) { S[(size_t)evc_index<M>(a, b) * ev::NBP + blk[i][j]] += v; }
  void fe(int a, int i, double v) { R[a * ev::MAXN + nloc[i]] += v; }
};
template <class M, class P>
int evc_replay(const P* p, const double* xyz, const double* u, const double* aux, double* val, double* rhs) {
  constexpr int NV = M::NV, NA = (M::NAUX > 0 ? M::NAUX : 1), NE = evc_blocks<M>();
  if (!g_ev.ok || g_prep.nvar != NV) return 1;
  const typename M::K k = M::derive(*p);
  const bool fastexp = exp_mode_of(M::exponent(k)) == M::FAST_EXP_MODE;
  const HostPrepEv& E = g_ev;
  std::vector<double> S((size_t)NE * ev::NBP), R((size_t)NV * ev::MAXN);
  for (size_t w = 0; w < E.desc.size(); w++) {
    const HostPrepEv::Desc& d = E.desc[w];
    const HostPrepEv::Node* nt = &E.ntab[w * HostPrepEv::MAXN];
    std::fill(S.begin(), S.end(), 0.0);
    std::fill(R.begin(), R.end(), 0.0);
    for (int x = 0; x < HostPrepEv::BLOCK; x++) {
      const uint32_t pl = E.vloc[w * HostPrepEv::BLOCK + (size_t)x];
      if (pl == 0xFFFFFFFFu) continue;
      double X[4][3], U[4][NV], AX[4][NA];
      EvcHostSink<M> sink{S.data(), R.data(), {}, {}};
      int li[4], r = 0;
      for (int j = 0; j < 4; j++) {
        li[j] = (int)((pl >> (8 * j)) & 0xFF);
        const uint32_t n = E.nlist[w * (size_t)E.nls + (size_t)li[j]];
        for (int c = 0; c < 3; c++) X[j][c] = xyz[3 * (size_t)n + c];
        for (int v = 0; v < NV; v++) U[j][v] = u[NV * (size_t)n + v];
        for (int v = 0; v < NA; v++) AX[j][v] = (M::NAUX > 0 && aux) ? aux[(size_t)M::NAUX * n + (M::NAUX > 0 ? v : 0)] : 0.0;
        r += li[j] < (int)d.nown;
      }
      for (int i = 0; i < 4; i++) {
        sink.nloc[i] = i < r ? li[i] : 0;
        const uint32_t word = E.vslot[(w * HostPrepEv::BLOCK + (size_t)x) * 2 + (size_t)(i >> 1)] >> (16 * (i & 1));
        for (int j = 0; j < 4; j++) sink.blk[i][j] = i < r ? (int)((word >> (4 * j)) & 0xF) * ev::MAXN + li[i] : 0;
      }
      if (fastexp) tet4_visit<M, M::FAST_EXP_MODE>(k, X, U, AX, r, sink); else tet4_visit<M, 0>(k, X, U, AX, r, sink);
    }
    for (uint32_t t = 0; t < (uint32_t)ev::NBP; t++) {
      const uint32_t bn = t & (ev::MAXN - 1), s2 = t >> 4;
      if (bn >= d.nown || s2 >= nt[bn].len) continue;
      const HostPrepEv::Node& nd = nt[bn];
      for (int a = 0; a < NV; a++)
        for (int b = 0; b < NV; b++)
          val[(size_t)NV * NV * nd.bptr + (size_t)a * NV * nd.len + (size_t)NV * s2 + b] =
              evc_block<M>(a, b) ? S[(size_t)evc_index<M>(a, b) * ev::NBP + t] : 0.0;
    }
    for (uint32_t n = 0; n < d.nown; n++)
      for (int a = 0; a < NV; a++) rhs[(size_t)NV * nt[n].node + a] = R[a * ev::MAXN + n];
  }
  return 0;
}

}  // namespace

extern "C" {

// model: 0 PIHNA, 1 RIPF, 2 HCC, 4 ADPM (ED = tract vector).  acc: [NV][NV][nen] (a, b, column node j in ORIGINAL local order)
int shim_row(int model, int nen, int fast, int force_general_pow, const void* params, const double* X, const double* U,
             const double* A, int irow, double* acc, double* fe, const double* ED) {
  switch (model) {
    case 5: return run<Proteas>((const rdc_proteas_params*)params, nen, fast, force_general_pow, X, U, A, irow, acc, fe, ED);
    case 4: return run<Adpm>((const rdc_adpm_params*)params, nen, fast, force_general_pow, X, U, A, irow, acc, fe, ED);
    case 0: return run<Pihna>((const rdc_pihna_params*)params, nen, fast, force_general_pow, X, U, A, irow, acc, fe);
    case 1: return run<Ripf>((const rdc_ripf_params*)params, nen, fast, force_general_pow, X, U, A, irow, acc, fe);
    case 2: return run<Hcc>((const rdc_hcc_params*)params, nen, fast, force_general_pow, X, U, A, irow, acc, fe);
    case 3:  // PIHNA, cell-transport-off variant: only legal when the parameters allow it
      if (!PihnaNoCellTransport::applies(*(const rdc_pihna_params*)params)) return 3;
      return run<PihnaNoCellTransport>((const rdc_pihna_params*)params, nen, fast, force_general_pow, X, U, A, irow, acc, fe);
    case 8:  // HCC with every rate zero (run/Coupled/HCC)
      if (!HccMassOnly::applies(*(const rdc_hcc_params*)params)) return 3;
      return run<HccMassOnly>((const rdc_hcc_params*)params, nen, fast, force_general_pow, X, U, A, irow, acc, fe);
    case 9:  // ADPM, decay terms only (run/HCP102513)
      if (!AdpmDecayOnly::applies(*(const rdc_adpm_params*)params)) return 3;
      return run<AdpmDecayOnly>((const rdc_adpm_params*)params, nen, fast, force_general_pow, X, U, A, irow, acc, fe, ED);
    case 7:  // RIPF, reduced variant (growth / HU rates / second source / radiotaxis off): only legal for such parameters
      if (!RipfReduced::applies(*(const rdc_ripf_params*)params)) return 3;
      return run<RipfReduced>((const rdc_ripf_params*)params, nen, fast, force_general_pow, X, U, A, irow, acc, fe);
    case 6:  // the same in moment form (rdc_tet4_pihna_moments.h); TET4 factored row only
      if (!PihnaNoCellTransport::applies(*(const rdc_pihna_params*)params)) return 3;
      if (!fast || nen != 4) return 1;
      return run<PihnaNoCellTransportMoments>((const rdc_pihna_params*)params, nen, fast, force_general_pow, X, U, A, irow, acc, fe);
  }
  return 2;
}

// 1.0 in *worst if coef() produced a non-zero outside the structural masks
int shim_masks(int model, const void* params, const double* u, const double* aux, double* worst) {
  switch (model) {
    case 0: return masks<Pihna>((const rdc_pihna_params*)params, u, aux, worst);
    case 1: return masks<Ripf>((const rdc_ripf_params*)params, u, aux, worst);
    case 2: return masks<Hcc>((const rdc_hcc_params*)params, u, aux, worst);
    case 3: return masks<PihnaNoCellTransport>((const rdc_pihna_params*)params, u, aux, worst);
    case 7: return masks<RipfReduced>((const rdc_ripf_params*)params, u, aux, worst);
    case 4: return masks<Adpm>((const rdc_adpm_params*)params, u, aux, worst);
    case 5: return masks<Proteas>((const rdc_proteas_params*)params, u, aux, worst);
  }
  return 2;
}

// ---- mesh preparation ---------------------------------------------------------------------
int shim_prep_build(int nen, int64_t n_elem, int64_t n_node, int64_t n_owned, const uint32_t* conn, int nvar,
                    int64_t lds_budget, int block) {
  g_err = prep_build(nen, n_elem, n_node, n_owned, conn, nvar, (size_t)lds_budget, block, g_prep);
  g_conn.assign(conn, conn + n_elem * nen);
  return g_err.empty() ? 0 : 1;
}

// element-visit lists (rdc_prep_ev.cpp) of the mesh of the last shim_prep_build; stats[6] = workgroups, visits, rows,
// list stride, largest CSR image (doubles), owned nodes covered
int shim_ev_build(int64_t lds_budget, int64_t* stats) {
  g_err = prep_build_ev(g_prep, g_conn.data(), (size_t)lds_budget, g_ev);
  if (!g_err.empty()) return 1;
  int64_t covered = 0;
  for (const auto& d : g_ev.desc) covered += d.nown;
  stats[0] = (int64_t)g_ev.desc.size(); stats[1] = g_ev.n_visits; stats[2] = g_ev.n_rows; stats[3] = g_ev.nls;
  stats[4] = (int64_t)g_ev.max_out_doubles; stats[5] = covered;
  return 0;
}

// cluster lists of the producer / consumer kernels (rdc_prep_cl.cpp) of the mesh of the last shim_prep_build, and a
// structural check of them: every (owned node, incident element) pair is listed exactly once, with the local element,
// local row node, owned-node index and column slots the kernel will use.
// stats[8] = workgroups, element visits, pairs, largest one-row image (doubles), owned nodes covered, largest cluster,
// 16-lane groups of pairs, groups in which a node occurs twice
static int64_t g_cl_interior = -1;   // "interior_nodes" of the next shim_cl_build (two-part assembly), -1 = none
void shim_cl_set_interior(int64_t n) { g_cl_interior = n; }
// two-part assembly: out[3] = leading clusters of interior nodes, rows complete after them, clusters mixing the two kinds (must be 0)
int shim_cl_interior_stats(int64_t* out) {
  const HostPrepCl& C = g_cl;
  out[0] = C.n_wg_interior; out[1] = C.part1_nodes; out[2] = 0;
  const int max_nodes = C.lim.max_nodes;
  for (size_t w = 0; w < C.desc.size(); w++) {
    int n_in = 0;
    for (int a = 0; a < C.desc[w].nown; a++) n_in += (int64_t)C.ntab[w * max_nodes + a].node < g_cl_interior;
    const bool interior = n_in == C.desc[w].nown;
    if (n_in != 0 && !interior) out[2]++;                                   // a cluster holds both kinds
    if (interior != ((int64_t)w < C.n_wg_interior)) return 1;               // the interior clusters are exactly the leading ones
    if (!interior)
      for (int a = 0; a < C.desc[w].nown; a++) if ((int64_t)C.ntab[w * max_nodes + a].node < C.part1_nodes) return 2;   // a row below the bound is not complete after part 1
  }
  return 0;
}
int shim_cl_build(int max_nodes, int max_pairs, int max_elems, int max_row_doubles, int pair_order, int64_t* stats) {
  HostPrepCl::Limits lim;
  stats[5] = 0;
  lim.max_nodes = max_nodes; lim.max_pairs = max_pairs; lim.max_elems = max_elems; lim.max_row_doubles = max_row_doubles; lim.pair_order = pair_order;
  g_err = prep_build_cl(g_prep, g_conn.data(), lim, g_cl, g_cl_interior);
  if (!g_err.empty()) return 1;
  const HostPrepCl& C = g_cl;
  const int nen = g_prep.nen;
  std::vector<int32_t> npairs_of((size_t)g_prep.n_owned, 0), seen((size_t)g_prep.n_owned, 0);
  int64_t covered = 0, largest = 0, groups = 0, groups_twice = 0;
  for (size_t w = 0; w < C.desc.size(); w++) {
    const HostPrepCl::Desc& d = C.desc[w];
    if (d.nown > lim.max_nodes || d.npair > lim.max_pairs || d.nelem > lim.max_elems || (int)d.row_doubles > lim.max_row_doubles) { g_err = "limits"; return 2; }
    covered += d.nown;
    largest = std::max<int64_t>(largest, d.nown);
    uint32_t off = 0;
    for (int a = 0; a < d.nown; a++) {
      const HostPrepCl::Node& nd = C.ntab[w * lim.max_nodes + a];
      if ((int64_t)nd.node >= g_prep.n_owned || seen[nd.node]++) { g_err = "node listed twice"; return 3; }
      if ((off ^ (uint32_t)(g_prep.nvar * g_prep.nvar * g_prep.bptr[nd.node])) & 1u) off++;   // the segment has the 16-byte phase of its CSR segment
      if (nd.bptr != (uint32_t)g_prep.bptr[nd.node] || nd.len != g_prep.bptr[nd.node + 1] - g_prep.bptr[nd.node] || nd.off != off) { g_err = "node table"; return 4; }
      off += (uint32_t)(g_prep.nvar * g_prep.nvar * nd.len);
    }
    if (off != d.row_doubles) { g_err = "row_doubles"; return 5; }
    for (int x = 0; x < lim.max_pairs; x++) {
      const uint32_t pr = C.pair[w * lim.max_pairs + x];
      if (x >= d.npair) { if (pr != 0xFFFFFFFFu) { g_err = "pair beyond npair"; return 6; } continue; }
      const uint32_t le = pr & 0xFF, li = (pr >> 8) & 0xFF, a = (pr >> 16) & 0xFF;
      if (le >= d.nelem || (int)li >= nen || a >= d.nown) { g_err = "pair fields"; return 7; }
      const uint32_t e = C.eid[w * lim.max_elems + le];
      const uint32_t n = C.ntab[w * lim.max_nodes + a].node;
      if (e >= (uint32_t)g_prep.n_elem || g_conn[(size_t)e * nen + li] != n) { g_err = "pair does not match the mesh"; return 8; }
      for (int j = 0; j < nen; j++) {
        const uint32_t s = (C.pslot[(w * lim.max_pairs + x) * (nen / 4) + j / 4] >> (8 * (j % 4))) & 0xFF;
        if (s != g_prep.eslot[(size_t)e * nen * nen + li * nen + j]) { g_err = "slot"; return 9; }
      }
      npairs_of[n]++;
    }
    // the 16 lanes of an atomic pass: count the groups in which a node occurs twice (statistics only)
    for (int x = 0; x + 16 <= d.npair; x += 16) {
      uint64_t m[4] = {0, 0, 0, 0};
      bool twice = false;
      for (int y = x; y < x + 16; y++) {
        const uint32_t a = (C.pair[w * lim.max_pairs + y] >> 16) & 0xFF;
        if (m[a >> 6] & (1ull << (a & 63))) twice = true;
        m[a >> 6] |= 1ull << (a & 63);
      }
      groups++;
      groups_twice += twice;
    }
  }
  // incident elements per owned node
  std::vector<int32_t> inc((size_t)g_prep.n_owned, 0);
  for (size_t x = 0; x < g_conn.size(); x++) if ((int64_t)g_conn[x] < g_prep.n_owned) inc[g_conn[x]]++;
  for (int64_t n = 0; n < g_prep.n_owned; n++)
    if (inc[n] != npairs_of[n] || seen[n] != 1) { g_err = "pair coverage"; return 10; }
  stats[0] = (int64_t)C.desc.size(); stats[1] = C.n_elem_visits; stats[2] = C.n_pairs; stats[3] = (int64_t)C.max_row_doubles;
  stats[4] = covered; stats[5] = largest; stats[6] = groups; stats[7] = groups_twice;
  return 0;
}

// model: 1 RIPF, 2 HCC, 4 ADPM, 7 RIPF reduced, 8 HCC mass only, 9 ADPM decay only (as shim_row)
int shim_cl_assemble(int model, const void* params, const double* xyz, const double* u, const double* aux, const double* elem,
                     double* val, double* rhs) {
  switch (model) {
    case 1: return cl_replay<Ripf>((const rdc_ripf_params*)params, xyz, u, aux, elem, val, rhs);
    case 2: return cl_replay<Hcc>((const rdc_hcc_params*)params, xyz, u, aux, elem, val, rhs);
    case 4: return cl_replay<Adpm>((const rdc_adpm_params*)params, xyz, u, aux, elem, val, rhs);
    case 7: if (!RipfReduced::applies(*(const rdc_ripf_params*)params)) return 3;
            return cl_replay<RipfReduced>((const rdc_ripf_params*)params, xyz, u, aux, elem, val, rhs);
    case 8: if (!HccMassOnly::applies(*(const rdc_hcc_params*)params)) return 3;
            return cl_replay<HccMassOnly>((const rdc_hcc_params*)params, xyz, u, aux, elem, val, rhs);
    case 9: if (!AdpmDecayOnly::applies(*(const rdc_adpm_params*)params)) return 3;
            return cl_replay<AdpmDecayOnly>((const rdc_adpm_params*)params, xyz, u, aux, elem, val, rhs);
  }
  return 2;
}

// model: 1 RIPF, 2 HCC, 7 RIPF reduced, 8 HCC mass only (as shim_row)
int shim_evc_assemble(int model, const void* params, const double* xyz, const double* u, const double* aux, double* val, double* rhs) {
  switch (model) {
    case 1: return evc_replay<Ripf>((const rdc_ripf_params*)params, xyz, u, aux, val, rhs);
    case 2: return evc_replay<Hcc>((const rdc_hcc_params*)params, xyz, u, aux, val, rhs);
    case 7: if (!RipfReduced::applies(*(const rdc_ripf_params*)params)) return 3;
            return evc_replay<RipfReduced>((const rdc_ripf_params*)params, xyz, u, aux, val, rhs);
    case 8: if (!HccMassOnly::applies(*(const rdc_hcc_params*)params)) return 3;
            return evc_replay<HccMassOnly>((const rdc_hcc_params*)params, xyz, u, aux, val, rhs);
  }
  return 2;
}

// The kernel k_tet4_ev replayed on the host, phase by phase and workgroup by workgroup, from the SAME lists and the
// SAME device functions (pihna_visit, pihna_expand): moments accumulated per cluster, expanded per node block into
// the LDS image of the CSR segments, segments copied out.  val / rhs must be pre-filled by the caller (entries the
// lists do not cover stay as they are, which the test detects).
void shim_ev_set_background(int on) { g_ev_bg = on; }
}  // extern "C"
template <bool GEN>
static int ev_assemble_impl(const rdc_pihna_params* p, const double* xyz, const double* u, double* val, double* rhs) {
  constexpr int NMT = GEN ? ev::NMG : ev::NM;
  if (!g_ev.ok) return 1;
  const PihnaK k = Pihna::derive(*p);
  const bool cube = exp_mode_of(k.ek) == 3;
  const HostPrepEv& E = g_ev;
  std::vector<double> M((size_t)NMT * ev::NBP), R(5 * ev::MAXN), img;
  for (size_t w = 0; w < E.desc.size(); w++) {
    const HostPrepEv::Desc& d = E.desc[w];
    const HostPrepEv::Node* nt = &E.ntab[w * HostPrepEv::MAXN];
    std::fill(M.begin(), M.end(), 0.0);
    std::fill(R.begin(), R.end(), 0.0);
    for (int x = 0; x < HostPrepEv::BLOCK; x++) {
      const uint32_t pl = E.vloc[w * HostPrepEv::BLOCK + (size_t)x];
      if (pl == 0xFFFFFFFFu) continue;
      double X[4][3], U[4][5];
      EvHostSink sink{M.data(), R.data(), {}, {}};
      int li[4], r = 0;
      for (int j = 0; j < 4; j++) {
        li[j] = (int)((pl >> (8 * j)) & 0xFF);
        if (li[j] >= (int)d.ntouch) return 4;
        const uint32_t n = E.nlist[w * (size_t)E.nls + (size_t)li[j]];
        for (int c = 0; c < 3; c++) X[j][c] = xyz[3 * (size_t)n + c];
        for (int v = 0; v < 5; v++) U[j][v] = u[5 * (size_t)n + v];
        r += li[j] < (int)d.nown;
      }
      for (int i = 0; i < 4; i++) {
        sink.nloc[i] = i < r ? li[i] : 0;
        const uint32_t word = E.vslot[(w * HostPrepEv::BLOCK + (size_t)x) * 2 + (size_t)(i >> 1)] >> (16 * (i & 1));   // 4-bit slots, two rows per word
        for (int j = 0; j < 4; j++) {
          const int slot = (int)((word >> (4 * j)) & 0xF);
          sink.blk[i][j] = i < r ? slot * ev::MAXN + li[i] : 0;     // block (node a, slot s) of the slice: s * 16 + a
          if (i < r && slot >= (int)nt[li[i]].len) return 5;
        }
      }
      const bool bg = g_ev_bg && ev::pihna_background(U);   // per visit here, per wave on the device
      if (cube) ev::pihna_visit<3, EvHostSink, true, GEN>(k, X, U, r, sink, bg); else ev::pihna_visit<0, EvHostSink, true, GEN>(k, X, U, r, sink, bg);
    }
    img.assign(d.out_doubles, 0.0);
    for (uint32_t t = 0; t < (uint32_t)ev::NBP; t++) {
      const uint32_t bn = t & (ev::MAXN - 1), s2 = t >> 4;
      if (bn >= d.nown || s2 >= nt[bn].len) continue;
      double e[NMT], o[25];
      for (int m = 0; m < NMT; m++) e[m] = M[(size_t)m * ev::NBP + t];
      const uint32_t mir = E.bpart[w * ev::NBP + t];
      if (mir != t)
        for (int m = 0; m < NMT; m++)
          if (GEN ? ev::symmetric_moment_gen(m) : ev::symmetric_moment(m)) e[m] += M[(size_t)m * ev::NBP + mir];
      ev::pihna_expand(k, e, o);
      const HostPrepEv::Node& nd = nt[bn];
      for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++) {
          const size_t at = (size_t)nd.obase + (size_t)a * 5 * nd.len + 5 * s2 + (size_t)b;
          if (at >= img.size()) return 7;
          img[at] = o[a * 5 + b];
        }
    }
    for (uint32_t n = 0; n < d.nown; n++) {
      const HostPrepEv::Node& nd = nt[n];
      if (((25ull * nd.bptr) & 1) != (nd.obase & 1)) return 8;   // image and memory share the 16-byte phase
      for (uint32_t x = 0; x < 25u * nd.len; x++) val[25 * (size_t)nd.bptr + x] = img[(size_t)nd.obase + x];
      for (int a = 0; a < 5; a++) rhs[5 * (size_t)nd.node + a] = R[(size_t)a * ev::MAXN + n];
    }
  }
  return 0;
}
extern "C" {
// shipped parameter pattern: the 16-moment kernel; anything else: all 22 moments (rdc_tet4_ev.h, GEN); gen = 1 forces the latter
int shim_ev_assemble_gen(const rdc_pihna_params* p, const double* xyz, const double* u, double* val, double* rhs, int gen) {
  if (gen || !PihnaNoCellTransport::applies(*p)) return ev_assemble_impl<true>(p, xyz, u, val, rhs);
  return ev_assemble_impl<false>(p, xyz, u, val, rhs);
}
int shim_ev_assemble(const rdc_pihna_params* p, const double* xyz, const double* u, double* val, double* rhs) {
  return shim_ev_assemble_gen(p, xyz, u, val, rhs, 0);
}
const char* shim_prep_error() { return g_err.c_str(); }
// gather lists of the two-pass solid assembly, from the last shim_prep_build
int shim_solid_gather_build() {
  g_err = solid_gather_build(g_prep, g_gather);
  return g_err.empty() ? 0 : 1;
}
int64_t shim_prep_size(int what) {
  switch (what) {
    case 0: return (int64_t)g_prep.bptr.size();
    case 1: return (int64_t)g_prep.bcol.size();
    case 2: return (int64_t)g_prep.eslot.size();
    case 3: return (int64_t)g_prep.colour.size();
    case 4: return (int64_t)g_prep.elem_order.size();
    case 5: return (int64_t)g_prep.colour_ptr.size();
    case 6: return (int64_t)g_prep.first_mask.size();
    case 7: return (int64_t)g_prep.first_rhs.size();
    case 8: return (int64_t)g_prep.pair_elem.size();
    case 9: return (int64_t)g_prep.pair_local.size();
    case 10: return (int64_t)g_prep.node_pair_ptr.size();
    case 11: return (int64_t)g_prep.wg_node_ptr.size();
    case 12: return (int64_t)(g_prep.wg2.size() * sizeof(HostPrep::WgDesc));
    case 13: return (int64_t)g_prep.pair_rec.size();
    case 14: return (int64_t)g_prep.pair_aux.size();
    case 15: return (int64_t)(g_prep.chunk.size() * sizeof(HostPrep::Chunk));
    case 16: return (int64_t)(g_prep.sdesc.size() * sizeof(HostPrep::StoreDesc));
    case 17: return (int64_t)g_prep.contrib.size();
    case 20: return (int64_t)g_gather.gptr.size();
    case 30: return (int64_t)g_ev.vloc.size();
    case 31: return (int64_t)(g_ev.desc.size() * sizeof(HostPrepEv::Desc));
    case 32: return (int64_t)g_ev.n_group_rows;
    case 33: return (int64_t)g_ev.n_conflicts;
    case 34: return (int64_t)g_ev.n_pass_instr;
    case 21: return (int64_t)g_gather.gsrc.size();
    case 22: return (int64_t)g_gather.brow.size();
    case 100: return g_prep.n_colours;
    case 101: return g_prep.rowgather_ok ? 1 : 0;
    case 102: return (int64_t)g_prep.rg_lds_bytes;
    case 103: return g_prep.rg2_ok ? 1 : 0;
    case 104: return (int64_t)g_prep.rg2_lds_bytes;
    case 105: return g_prep.rg2_block;
  }
  return -1;
}
int shim_prep_copy(int what, void* dst) {
#define CP(v) std::memcpy(dst, g_prep.v.data(), g_prep.v.size() * sizeof(g_prep.v[0])); return 0
  switch (what) {
    case 0: CP(bptr);
    case 1: CP(bcol);
    case 2: CP(eslot);
    case 3: CP(colour);
    case 4: CP(elem_order);
    case 5: CP(colour_ptr);
    case 6: CP(first_mask);
    case 7: CP(first_rhs);
    case 8: CP(pair_elem);
    case 9: CP(pair_local);
    case 10: CP(node_pair_ptr);
    case 11: CP(wg_node_ptr);
    case 12: CP(wg2);
    case 13: CP(pair_rec);
    case 14: CP(pair_aux);
    case 15: CP(chunk);
    case 16: CP(sdesc);
    case 17: CP(contrib);
  }
#undef CP
  if (what == 30) { std::memcpy(dst, g_ev.vloc.data(), g_ev.vloc.size() * 4); return 0; }
  if (what == 31) { std::memcpy(dst, g_ev.desc.data(), g_ev.desc.size() * sizeof(HostPrepEv::Desc)); return 0; }
#define CG(v) std::memcpy(dst, g_gather.v.data(), g_gather.v.size() * sizeof(g_gather.v[0])); return 0
  switch (what) {
    case 20: CG(gptr);
    case 21: CG(gsrc);
    case 22: CG(brow);
  }
#undef CG
  return 1;
}

}  // extern "C"
