// rdc_prep_ev.cpp — host preparation of the element-visit kernel (rdc_tet4_ev.hip): node clusters and their work lists.
//
// Clusters are grown greedily over the mesh graph: starting from the lowest unassigned owned node, the unassigned
// owned node that shares the most elements with the cluster joins it, until a limit is hit (16 nodes, 256 element
// visits, 255 distinct nodes, 256 node blocks, the LDS budget of the CSR image).  The more elements a cluster holds
// completely, the more rows an element visit serves (the per-element work is done once per visit).
#include <algorithm>
#include <cstring>
#include <numeric>

#include "rdc_prep.h"

namespace rdc {

std::string prep_build_ev(const HostPrep& P, const uint32_t* conn, size_t lds_budget_bytes, HostPrepEv& E) {
  E = HostPrepEv();
  if (P.nen != 4 || P.nvar != 5) return "element-visit lists exist for TET4 with 5 unknowns only";
  const int64_t n_elem = P.n_elem, n_node = P.n_node, n_owned = P.n_owned;
  if (n_owned <= 0) return "no owned nodes";
  if (P.bptr[(size_t)n_owned] >= ((int64_t)1 << 32)) return "more than 2^32 node blocks";
  constexpr int MAXN = HostPrepEv::MAXN, NBP = HostPrepEv::NBP, BLOCK = HostPrepEv::BLOCK;
  // node -> incident elements (owned nodes only need them)
  std::vector<int64_t> inc_ptr((size_t)n_node + 1, 0);
  for (int64_t x = 0; x < n_elem * 4; x++) inc_ptr[conn[x] + 1]++;
  for (int64_t n = 0; n < n_node; n++) inc_ptr[n + 1] += inc_ptr[n];
  std::vector<uint32_t> inc((size_t)inc_ptr[n_node]);
  {
    std::vector<int64_t> fill(inc_ptr.begin(), inc_ptr.end() - 1);
    for (int64_t e = 0; e < n_elem; e++)
      for (int i = 0; i < 4; i++) inc[fill[conn[e * 4 + i]]++] = (uint32_t)e;
  }
  const size_t budget_doubles = lds_budget_bytes / sizeof(double);
  // ---- greedy clustering -----------------------------------------------------------------------------------------
  std::vector<int32_t> cluster_of((size_t)n_owned, -1);   // -1 unassigned, -2 rejected for the cluster being grown
  std::vector<std::vector<uint32_t>> clusters;
  {
    std::vector<uint32_t> emark((size_t)n_elem, 0), nmark((size_t)n_node, 0), gain((size_t)n_owned, 0), gstamp((size_t)n_owned, 0);
    uint32_t stamp = 0;
    std::vector<uint32_t> cand, tmp, rejected;
    for (int64_t seed = 0; seed < n_owned; seed++) {
      if (cluster_of[seed] >= 0) continue;
      stamp++;
      std::vector<uint32_t> cl;
      int64_t nvis = 0, ntouch = 0, nb = 0;
      size_t img = 0;
      cand.clear();
      rejected.clear();
      auto cost_of = [&](uint32_t n, int64_t& dv, int64_t& dt) {   // new visits / new distinct nodes if n joined
        dv = 0; dt = 0;
        tmp.clear();
        for (int64_t k = inc_ptr[n]; k < inc_ptr[n + 1]; k++) {
          const uint32_t e = inc[k];
          if (emark[e] == stamp) continue;
          dv++;
          for (int j = 0; j < 4; j++) {
            const uint32_t m = conn[(int64_t)e * 4 + j];
            if (nmark[m] != stamp && std::find(tmp.begin(), tmp.end(), m) == tmp.end()) { tmp.push_back(m); dt++; }
          }
        }
      };
      auto add = [&](uint32_t n) {
        cluster_of[n] = (int32_t)clusters.size();
        cl.push_back(n);
        nb += P.bptr[n + 1] - P.bptr[n];
        img += (size_t)25 * (size_t)(P.bptr[n + 1] - P.bptr[n]) + 1;
        for (int64_t k = inc_ptr[n]; k < inc_ptr[n + 1]; k++) {
          const uint32_t e = inc[k];
          if (emark[e] == stamp) continue;
          emark[e] = stamp;   // a new visit: every unassigned owned node of it shares one more element with the cluster
          nvis++;
          for (int j = 0; j < 4; j++) {
            const uint32_t m = conn[(int64_t)e * 4 + j];
            if (nmark[m] != stamp) { nmark[m] = stamp; ntouch++; }
            if ((int64_t)m < n_owned && cluster_of[m] == -1) {
              if (gstamp[m] != stamp) { gstamp[m] = stamp; gain[m] = 0; cand.push_back(m); }
              gain[m]++;
            }
          }
        }
      };
      add((uint32_t)seed);
      while ((int)cl.size() < MAXN) {
        // best candidate: shares the most elements with the cluster, then the lowest id
        int best = -1;
        for (size_t x = 0; x < cand.size(); x++) {
          const uint32_t c = cand[x];
          if (cluster_of[c] != -1) continue;
          if (best < 0 || gain[c] > gain[cand[(size_t)best]] || (gain[c] == gain[cand[(size_t)best]] && c < cand[(size_t)best])) best = (int)x;
        }
        if (best < 0) break;
        const uint32_t c = cand[(size_t)best];
        int64_t dv, dt;
        cost_of(c, dv, dt);
        const int64_t lenc = P.bptr[c + 1] - P.bptr[c];
        if (nvis + dv > BLOCK || ntouch + dt > 255 || nb + lenc > NBP || img + (size_t)25 * (size_t)lenc + 1 > budget_doubles) {
          cluster_of[c] = -2;   // does not fit: out of the race until this cluster is closed
          rejected.push_back(c);
          continue;
        }
        add(c);
      }
      for (uint32_t c : rejected) cluster_of[c] = -1;
      if (nvis > BLOCK || nb > NBP || img > budget_doubles) return "a single node exceeds the element-visit limits";
      clusters.push_back(std::move(cl));
    }
  }
  const int64_t nwg = (int64_t)clusters.size();
  // ---- per-workgroup lists -----------------------------------------------------------------------------------------
  // first pass: list strides
  std::vector<int32_t> ntouch_w((size_t)nwg, 0);
  E.desc.resize((size_t)nwg);
  int fail = 0;
  std::vector<std::vector<uint32_t>> touched((size_t)nwg), visits((size_t)nwg);
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t w = 0; w < nwg; w++) {
    const std::vector<uint32_t>& cl = clusters[(size_t)w];
    std::vector<uint32_t>& vis = visits[(size_t)w];
    for (uint32_t n : cl)
      for (int64_t k = inc_ptr[n]; k < inc_ptr[n + 1]; k++) vis.push_back(inc[k]);
    std::sort(vis.begin(), vis.end());
    vis.erase(std::unique(vis.begin(), vis.end()), vis.end());
    std::vector<uint32_t>& t = touched[(size_t)w];
    for (uint32_t e : vis)
      for (int j = 0; j < 4; j++) t.push_back(conn[(int64_t)e * 4 + j]);
    std::sort(t.begin(), t.end());
    t.erase(std::unique(t.begin(), t.end()), t.end());
    // owned nodes of the cluster first (in cluster order), then the rest ascending
    std::vector<uint32_t> ordered(cl.begin(), cl.end());
    for (uint32_t m : t) if (!((int64_t)m < n_owned && cluster_of[m] == (int32_t)w)) ordered.push_back(m);
    t.swap(ordered);
    ntouch_w[(size_t)w] = (int32_t)t.size();
    if (t.size() > 255 || vis.size() > (size_t)BLOCK) fail = 1;
  }
  if (fail) return "internal: cluster limits violated";
  int mx = 1;
  for (int64_t w = 0; w < nwg; w++) mx = std::max(mx, (int)ntouch_w[(size_t)w]);
  E.nls = (mx + 63) & ~63;
  E.nlist.assign((size_t)nwg * E.nls, 0);
  E.vloc.assign((size_t)nwg * BLOCK, 0xFFFFFFFFu);
  E.vslot.assign((size_t)nwg * BLOCK * 4, 0);
  E.ntab.assign((size_t)nwg * MAXN, HostPrepEv::Node{0, 0, 0, 0, 0});
  E.btab.assign((size_t)nwg * NBP, 0);
  std::vector<int64_t> rows_w((size_t)nwg, 0);
  std::vector<size_t> img_w((size_t)nwg, 0);
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t w = 0; w < nwg; w++) {
    const std::vector<uint32_t>& cl = clusters[(size_t)w];
    const std::vector<uint32_t>& t = touched[(size_t)w];
    const std::vector<uint32_t>& vis = visits[(size_t)w];
    HostPrepEv::Desc& d = E.desc[(size_t)w];
    d.nown = (uint32_t)cl.size(); d.nvis = (uint32_t)vis.size(); d.ntouch = (uint32_t)t.size(); d.pad = 0;
    d.min_node = *std::min_element(cl.begin(), cl.end());
    d.max_node = *std::max_element(cl.begin(), cl.end());
    uint32_t* nl = &E.nlist[(size_t)w * E.nls];
    for (int x = 0; x < E.nls; x++) nl[x] = t[std::min<size_t>((size_t)x, t.size() - 1)];
    // node table: moment-slice blocks and the CSR image, node after node
    uint32_t blk = 0, ob = 0;
    for (size_t x = 0; x < cl.size(); x++) {
      const uint32_t n = cl[x];
      HostPrepEv::Node& nd = E.ntab[(size_t)w * MAXN + x];
      const int64_t len = P.bptr[n + 1] - P.bptr[n];
      nd.bptr = (uint32_t)P.bptr[n]; nd.len = (uint16_t)len; nd.blk0 = (uint16_t)blk; nd.node = n;
      const uint32_t phase = (uint32_t)((25 * P.bptr[n]) & 1);
      if ((ob & 1) != phase) ob++;          // the image of a segment starts at the 16-byte phase it has in memory
      nd.obase = ob;
      ob += (uint32_t)(25 * len);
      for (int64_t s2 = 0; s2 < len; s2++) E.btab[(size_t)w * NBP + blk + (uint32_t)s2] = (uint16_t)(x | ((uint32_t)s2 << 8));
      blk += (uint32_t)len;
      if (len > 255) fail = 1;
    }
    d.nb = blk;
    d.out_doubles = (ob + 1) & ~1u;
    img_w[(size_t)w] = d.out_doubles;
    if (blk > (uint32_t)NBP) fail = 1;
    // visits: vertices permuted owned-first; sorted by the number of owned vertices (descending) so that the waves of
    // the kernel are (nearly) uniform in the number of rows they emit
    struct V { uint32_t e; int r; int perm[4]; };
    std::vector<V> vv(vis.size());
    for (size_t x = 0; x < vis.size(); x++) {
      V& v = vv[x];
      v.e = vis[x];
      int own[4], rest[4], no = 0, nr = 0;
      for (int j = 0; j < 4; j++) {
        const uint32_t m = conn[(int64_t)v.e * 4 + j];
        if ((int64_t)m < n_owned && cluster_of[m] == (int32_t)w) own[no++] = j; else rest[nr++] = j;
      }
      v.r = no;
      for (int j = 0; j < no; j++) v.perm[j] = own[j];
      for (int j = 0; j < nr; j++) v.perm[no + j] = rest[j];
    }
    std::stable_sort(vv.begin(), vv.end(), [](const V& a, const V& b) { return a.r > b.r; });
    for (size_t x = 0; x < vv.size(); x++) {
      const V& v = vv[x];
      uint32_t packed = 0;
      uint32_t li[4];
      for (int j = 0; j < 4; j++) {
        const uint32_t m = conn[(int64_t)v.e * 4 + v.perm[j]];
        const uint32_t pos = (uint32_t)(std::find(t.begin(), t.end(), m) - t.begin());
        li[j] = pos;
        packed |= pos << (8 * j);
      }
      E.vloc[(size_t)w * BLOCK + x] = packed;
      rows_w[(size_t)w] += v.r;
      for (int i = 0; i < v.r; i++) {
        uint32_t word = 0;
        for (int j = 0; j < 4; j++)
          word |= (uint32_t)P.eslot[(size_t)v.e * 16 + (size_t)v.perm[i] * 4 + (size_t)v.perm[j]] << (8 * j);
        E.vslot[((size_t)w * BLOCK + x) * 4 + (size_t)i] = word;
        if (li[i] >= cl.size()) fail = 1;
      }
    }
  }
  if (fail) return "internal: element-visit list construction failed";
  for (int64_t w = 0; w < nwg; w++) {
    E.max_out_doubles = std::max(E.max_out_doubles, img_w[(size_t)w]);
    E.n_visits += (int64_t)E.desc[(size_t)w].nvis;
    E.n_rows += rows_w[(size_t)w];
  }
  E.ok = true;
  return std::string();
}

}  // namespace rdc
