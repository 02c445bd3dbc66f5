// rdc_prep_ev.cpp — host preparation of the element-visit kernel (rdc_tet4_ev.hip): node clusters and their work lists.
//
// Clusters are grown greedily over the mesh graph: starting from the lowest unassigned owned node, the unassigned
// owned node that shares the most elements with the cluster joins it, until a limit is hit (16 nodes, 256 element
// visits, 255 distinct nodes, 256 node blocks, the LDS budget of the CSR image).  The more elements a cluster holds
// completely, the more rows an element visit serves (the per-element work is done once per visit).
//
// LDS layout and schedule.  gfx950 executes a ds_add_f64 wave instruction as 4 groups of 16 lanes; inside a group, lanes
// whose double index is equal modulo 16 are serialised (tools/lds_bank_model.hip), and so are lanes that hit the same
// address.  The moment slice is therefore laid out [moment][slot in the row][owned node]: block (node a, slot s) has
// index s * 16 + a, i.e. its bank is the ROW NODE's index a.  One instruction of the kernel adds moment m of block
// (vertex i, vertex j) for all lanes, so it is free of bank and address conflicts iff, inside every 16-lane group, the
// lanes that execute it have DIFFERENT row nodes at vertex position i.  The vertex order inside the owned set is free,
// and so is the lane of a visit inside its wave: the schedule below places every visit in the 16-lane group and with
// the vertex order that collide least (ideally: every owned node appears at most once per group and position).
#include <algorithm>
#include <array>
#include <cstring>
#include <numeric>

#include "rdc_prep.h"

namespace rdc {

std::string prep_build_ev(const HostPrep& P, const uint32_t* conn, size_t lds_budget_bytes, HostPrepEv& E, int64_t n_interior) {
  E = HostPrepEv();
  if (P.nen != 4 || (P.nvar != 5 && P.nvar != 3)) return "element-visit lists exist for TET4 with 3 or 5 unknowns only";
  const size_t nv2 = (size_t)P.nvar * P.nvar;   // CSR values per node block
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
    // Seeds: always the unassigned node with the FEWEST unassigned neighbours (lowest id among equals).  Clusters then
    // grow along the front of what is already assigned and pick up nodes before they are cut off; taking the seeds in
    // id order instead leaves ~5% one-node clusters behind on a Kuhn mesh, each costing a whole workgroup.
    // free_nb[n] = unassigned owned neighbours of n (node graph = the block pattern); lazy bucket queue on it.
    std::vector<int32_t> free_nb((size_t)n_owned, 0);
    int max_nb = 0;
    for (int64_t n = 0; n < n_owned; n++) {
      // the moment slice holds NBP / MAXN = 16 node blocks per row: a node of higher valence cannot be placed -- reject the
      // mesh before any list is written (the caller falls back to the pair kernels)
      if (P.bptr[n + 1] - P.bptr[n] > NBP / MAXN) return "a node has more than 16 node blocks in its row (valence > 15): no element-visit lists";
      int c = 0;
      for (int64_t b = P.bptr[n]; b < P.bptr[n + 1]; b++) c += (P.bcol[b] != (int32_t)n && (int64_t)P.bcol[b] < n_owned);
      free_nb[n] = c;
      max_nb = std::max(max_nb, c);
    }
    std::vector<std::vector<uint32_t>> bucket((size_t)max_nb + 1);
    for (int64_t n = n_owned - 1; n >= 0; n--) bucket[(size_t)free_nb[n]].push_back((uint32_t)n);   // popped from the back: lowest id first
    auto assigned_update = [&](uint32_t n) {     // n has just been assigned: its neighbours lose a free neighbour
      for (int64_t b = P.bptr[n]; b < P.bptr[n + 1]; b++) {
        const int32_t m = P.bcol[b];
        if (m == (int32_t)n || (int64_t)m >= n_owned || cluster_of[m] >= 0) continue;
        free_nb[m]--;
        bucket[(size_t)free_nb[m]].push_back((uint32_t)m);   // lazy: stale entries in higher buckets are skipped when popped
      }
    };
    int64_t n_assigned = 0;
    while (n_assigned < n_owned) {
      int64_t seed = -1;
      for (size_t k = 0; k < bucket.size() && seed < 0; k++) {
        while (!bucket[k].empty()) {
          const uint32_t n = bucket[k].back();
          bucket[k].pop_back();
          if (cluster_of[n] < 0 && free_nb[n] == (int32_t)k) { seed = n; break; }
        }
      }
      if (seed < 0) return "internal: seed queue ran dry";
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
        n_assigned++;
        assigned_update(n);
        nb += P.bptr[n + 1] - P.bptr[n];
        img += nv2 * (size_t)(P.bptr[n + 1] - P.bptr[n]) + 1;
        for (int64_t k = inc_ptr[n]; k < inc_ptr[n + 1]; k++) {
          const uint32_t e = inc[k];
          if (emark[e] == stamp) continue;
          emark[e] = stamp;   // a new visit: every unassigned owned node of it shares one more element with the cluster
          nvis++;
          for (int j = 0; j < 4; j++) {
            const uint32_t m = conn[(int64_t)e * 4 + j];
            if (nmark[m] != stamp) { nmark[m] = stamp; ntouch++; }
            // two-part assembly: a cluster never mixes interior nodes (rows assembled before the halo exchange has
            // landed) with the others, so the interior rows stay complete clusters
            if ((int64_t)m < n_owned && cluster_of[m] == -1 && (n_interior < 0 || (((int64_t)m < n_interior) == (seed < n_interior)))) {
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
          if (best < 0) { best = (int)x; continue; }
          const uint32_t bc = cand[(size_t)best];
          // most shared elements; then the node with the fewest unassigned neighbours (it would be cut off otherwise)
          if (gain[c] > gain[bc] || (gain[c] == gain[bc] && (free_nb[c] < free_nb[bc] || (free_nb[c] == free_nb[bc] && c < bc)))) best = (int)x;
        }
        if (best < 0) break;
        const uint32_t c = cand[(size_t)best];
        int64_t dv, dt;
        cost_of(c, dv, dt);
        const int64_t lenc = P.bptr[c + 1] - P.bptr[c];
        if (nvis + dv > BLOCK || ntouch + dt > 255 || nb + lenc > NBP || img + nv2 * (size_t)lenc + 1 > budget_doubles) {
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
    if (t.size() > 255 || vis.size() > (size_t)BLOCK) {
#pragma omp atomic write
      fail = 1;
    }
  }
  if (fail) return "internal: cluster limits violated";
  int mx = 1;
  for (int64_t w = 0; w < nwg; w++) mx = std::max(mx, (int)ntouch_w[(size_t)w]);
  E.nls = (mx + 63) & ~63;
  E.nlist.assign((size_t)nwg * E.nls, 0);
  E.vloc.assign((size_t)nwg * BLOCK, 0xFFFFFFFFu);
  E.vslot.assign((size_t)nwg * BLOCK * 2, 0);
  E.ntab.assign((size_t)nwg * MAXN, HostPrepEv::Node{0, 0, 0, 0, 0});
  E.bpart.assign((size_t)nwg * NBP, 0);
  std::vector<int64_t> rows_w((size_t)nwg, 0), conf_w((size_t)nwg, 0), pass_w((size_t)nwg, 0), instr_w((size_t)nwg, 0);
  std::vector<size_t> img_w((size_t)nwg, 0);
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t w = 0; w < nwg; w++) {
    const std::vector<uint32_t>& cl = clusters[(size_t)w];
    const std::vector<uint32_t>& t = touched[(size_t)w];
    const std::vector<uint32_t>& vis = visits[(size_t)w];
    HostPrepEv::Desc& d = E.desc[(size_t)w];
    bool wfail = false;
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
      nd.bptr = (uint32_t)P.bptr[n]; nd.len = (uint16_t)len; nd.blk0 = (uint16_t)x; nd.node = n;   // block (x, slot s) lives at s * MAXN + x
      const uint32_t phase = (uint32_t)(((int64_t)nv2 * P.bptr[n]) & 1);
      if ((ob & 1) != phase) ob++;          // the image of a segment starts at the 16-byte phase it has in memory
      nd.obase = ob;
      ob += (uint32_t)(nv2 * (size_t)len);
      blk += (uint32_t)len;
      if (len > NBP / MAXN) wfail = true;
    }
    d.nb = blk;
    if (wfail) {   // a row does not fit the slice: write nothing more for this workgroup (checked up front; kept as a guard)
#pragma omp atomic write
      fail = 1;
      continue;
    }
    // mirror blocks: block (x, slot s) whose column node is cluster node x2 != x <-> block (x2, slot of node x in the row of x2)
    {
      uint8_t* bp = &E.bpart[(size_t)w * NBP];
      for (int t2 = 0; t2 < NBP; t2++) bp[t2] = (uint8_t)t2;
      for (size_t x = 0; x < cl.size(); x++) {
        const uint32_t n = cl[x];
        for (int64_t b = P.bptr[n]; b < P.bptr[n + 1]; b++) {
          const uint32_t c2 = (uint32_t)P.bcol[b];
          if (c2 == n || (int64_t)c2 >= n_owned || cluster_of[c2] != (int32_t)w) continue;
          const size_t x2 = (size_t)(std::find(cl.begin(), cl.end(), c2) - cl.begin());
          const int32_t* row2 = &P.bcol[P.bptr[c2]];
          const int64_t len2 = P.bptr[c2 + 1] - P.bptr[c2];
          const int64_t s2 = std::find(row2, row2 + len2, (int32_t)n) - row2;
          if (s2 >= len2) { wfail = true; continue; }   // the node graph is symmetric
          bp[(size_t)(b - P.bptr[n]) * MAXN + x] = (uint8_t)(s2 * MAXN + (int64_t)x2);
        }
      }
    }
    d.out_doubles = (ob + 1) & ~1u;
    img_w[(size_t)w] = d.out_doubles;
    // visits: vertices permuted owned-first; sorted by the number of owned vertices (descending) so that the waves of
    // the kernel are (nearly) uniform in the number of rows they emit; inside a wave, conflict-aware placement (above)
    struct V { uint32_t e; int r; int own[4]; int rest[4]; };
    std::vector<V> vv(vis.size());
    for (size_t x = 0; x < vis.size(); x++) {
      V& v = vv[x];
      v.e = vis[x];
      int no = 0, nr = 0;
      for (int j = 0; j < 4; j++) {
        const uint32_t m = conn[(int64_t)v.e * 4 + j];
        if ((int64_t)m < n_owned && cluster_of[m] == (int32_t)w) v.own[no++] = j; else v.rest[nr++] = j;
      }
      v.r = no;
    }
    std::stable_sort(vv.begin(), vv.end(), [](const V& a, const V& b) { return a.r > b.r; });
    // owned-node index (position in cl) of a local vertex
    auto own_index = [&](const V& v, int j) {
      const uint32_t m = conn[(int64_t)v.e * 4 + j];
      return (int)(std::find(cl.begin(), cl.end(), m) - cl.begin());
    };
    static const int PERM4[24][4] = {{0,1,2,3},{0,1,3,2},{0,2,1,3},{0,2,3,1},{0,3,1,2},{0,3,2,1},{1,0,2,3},{1,0,3,2},{1,2,0,3},{1,2,3,0},{1,3,0,2},{1,3,2,0},
                                     {2,0,1,3},{2,0,3,1},{2,1,0,3},{2,1,3,0},{2,3,0,1},{2,3,1,0},{3,0,1,2},{3,0,2,1},{3,1,0,2},{3,1,2,0},{3,2,0,1},{3,2,1,0}};
    // the 1 / 2 / 6 / 24 orders of the r owned vertices = the PERM4 entries that keep the tail fixed
    static const int NPERM[5] = {1, 1, 2, 6, 24};
    static const int PERM_R[5][24] = {{0}, {0}, {0, 6}, {0, 2, 6, 8, 12, 14},
                                      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23}};
    // ---- placement -----------------------------------------------------------------------------------------------------
    // What the kernel pays for.  A wave executes row position i (15 (4 - i) + 9 LDS atomics in the PIHNA kernel: PASS_COST)
    // when ANY of its lanes has more than i rows, whatever the number of active lanes; and inside a 16-lane group two
    // lanes with the same row node at the same position hit one bank with every atomic of that position (a COLLISION,
    // worth ~17 atomic instructions).  The schedule therefore first picks a PROFILE -- how many row positions each of the
    // four waves issues, e.g. (4, 2, 1, 1) or (4, 2, 1, 0) -- cheapest first among those with enough collision-free lane
    // slots (a group holds a node once per position: 4 min(16, nown) lanes per wave), then places the visits, deepest
    // first, each into the shallowest wave that can take it, in the group and with the vertex order that collide with
    // nothing; a visit that finds no free place ejects one lane that does (one level), and only then accepts a collision.
    // K(n) meshes: 405 instead of 500 atomic wave-instructions per cluster and no collisions (4.4 before).
    const int nlanes = (int)vv.size();
    std::vector<std::array<int, 4>> oiv((size_t)nlanes);
    for (int x = 0; x < nlanes; x++)
      for (int i = 0; i < 4; i++) oiv[(size_t)x][(size_t)i] = i < vv[(size_t)x].r ? own_index(vv[(size_t)x], vv[(size_t)x].own[i]) : 0;
    const int* PASS_COST = P.nvar == 5 ? HostPrepEv::PASS_COST_PIHNA : HostPrepEv::PASS_COST_FLAT;
    auto prefix_cost = [&](int np) { int c = 0; for (int i = 0; i < np; i++) c += PASS_COST[i]; return c; };
    // demand[i][a]: lanes that can put node a at position i ONLY IF it is free there, weighted by how few positions they
    // can use: a lane with k rows spreads 1 / k over the positions 0 .. k - 1 of each of its nodes (scaled by 12)
    std::array<std::array<int, 16>, 4> demand;
    for (auto& dm : demand) dm.fill(0);
    for (int x = 0; x < nlanes; x++) {
      const int r = vv[(size_t)x].r;
      for (int i = 0; i < r; i++)
        for (int p2 = 0; p2 < r; p2++) demand[(size_t)p2][(size_t)oiv[(size_t)x][(size_t)i]] += 12 / r;
    }
    struct Placement { std::vector<int8_t> grp, q; long instr = 0, coll = 0; bool ok = false; };
    auto try_profile = [&](const int (&prof)[4], Placement& out) {
      std::array<std::array<uint16_t, 4>, 16> used;
      for (auto& u2 : used) u2.fill(0);
      int fill_g[16] = {0};
      std::vector<int8_t>& grp = out.grp; std::vector<int8_t>& qq = out.q;
      grp.assign((size_t)nlanes, -1); qq.assign((size_t)nlanes, 0);
      auto conflicts = [&](int g, int x, int q) {
        const int r = vv[(size_t)x].r; const int* pm = PERM4[PERM_R[r][q]];
        int c = 0;
        for (int i = 0; i < r; i++) c += (used[(size_t)g][(size_t)i] >> oiv[(size_t)x][(size_t)pm[i]]) & 1;
        return c;
      };
      auto put = [&](int x, int g, int q) {
        const int r = vv[(size_t)x].r; const int* pm = PERM4[PERM_R[r][q]];
        for (int i = 0; i < r; i++) used[(size_t)g][(size_t)i] |= (uint16_t)(1u << oiv[(size_t)x][(size_t)pm[i]]);
        fill_g[g]++; grp[(size_t)x] = (int8_t)g; qq[(size_t)x] = (int8_t)q;
      };
      // NOTE: a collision leaves the node bit set by two lanes; `take` of one of them would clear it for both, so lanes
      // are only ever taken out of groups in which they do not collide (the repair below moves collision-free lanes only)
      auto take = [&](int x) {
        const int g = grp[(size_t)x]; const int r = vv[(size_t)x].r; const int* pm = PERM4[PERM_R[r][(int)qq[(size_t)x]]];
        for (int i = 0; i < r; i++) used[(size_t)g][(size_t)i] &= (uint16_t)~(1u << oiv[(size_t)x][(size_t)pm[i]]);
        fill_g[g]--; grp[(size_t)x] = -1;
      };
      // best free (group, order) of lane x: shallowest wave first, then the order that keeps the nodes the shallower lanes
      // will ask for out of the low positions (a lane with k rows can only use positions 0 .. k - 1), then the fullest group
      auto find_free = [&](int x, int& g_out, int& q_out) {
        const int r = vv[(size_t)x].r;
        long best = -1;
        for (int g = 0; g < 16; g++) {
          if (prof[g >> 2] < r || fill_g[g] >= 16) continue;
          for (int q = 0; q < NPERM[r]; q++)
            if (conflicts(g, x, q) == 0) {
              const int* pm = PERM4[PERM_R[r][q]];
              long low = 0;   // demand of the shallower lanes for the (position, node) slots this order takes
              for (int i = 0; i < r; i++) low += demand[(size_t)i][(size_t)oiv[(size_t)x][(size_t)pm[i]]];
              const long cost = (long)(prof[g >> 2] - r) * 100000 + low * 20 + (16 - fill_g[g]);
              if (best < 0 || cost < best) { best = cost; g_out = g; q_out = q; }
            }
        }
        return best >= 0;
      };
      std::vector<int> pending;
      for (int x = 0; x < nlanes; x++) {   // vv is sorted by r, descending
        int g, q;
        if (find_free(x, g, q)) put(x, g, q); else pending.push_back(x);
      }
      out.coll = 0;
      std::vector<uint8_t> collided((size_t)nlanes, 0);
      for (int x : pending) {
        const int r = vv[(size_t)x].r;
        bool done = false;
        // eject one collision-free lane y of a group in which x then fits, and find y a free place elsewhere
        for (int g = 0; g < 16 && !done; g++) {
          if (prof[g >> 2] < r) continue;
          for (int y = 0; y < nlanes && !done; y++) {
            if (grp[(size_t)y] != g || collided[(size_t)y]) continue;
            const int qy = qq[(size_t)y];
            take(y);
            int qx = -1;
            for (int q = 0; q < NPERM[r]; q++) if (conflicts(g, x, q) == 0) { qx = q; break; }
            if (qx >= 0) {
              put(x, g, qx);
              int g2, q2;
              if (find_free(y, g2, q2)) { put(y, g2, q2); done = true; }
              else take(x);
            }
            if (!done) put(y, g, qy);
          }
        }
        if (done) continue;
        // nothing helps: the place with the fewest collisions
        long best = -1; int bg = -1, bq = 0;
        for (int g = 0; g < 16; g++) {
          if (prof[g >> 2] < r || fill_g[g] >= 16) continue;
          for (int q = 0; q < NPERM[r]; q++) {
            const long c = conflicts(g, x, q);
            if (best < 0 || c < best) { best = c; bg = g; bq = q; }
          }
        }
        if (bg < 0) { out.ok = false; return; }
        put(x, bg, bq);
        collided[(size_t)x] = 1;
        for (int y = 0; y < nlanes; y++) if (grp[(size_t)y] == bg && y != x) collided[(size_t)y] = 1;   // conservative: nobody leaves this group any more
        out.coll += best;
      }
      int maxr_w[4] = {0, 0, 0, 0};
      for (int x = 0; x < nlanes; x++) maxr_w[grp[(size_t)x] >> 2] = std::max(maxr_w[grp[(size_t)x] >> 2], vv[(size_t)x].r);
      out.instr = 0;
      for (int wv = 0; wv < 4; wv++) out.instr += prefix_cost(maxr_w[wv]);
      out.ok = true;
    };
    Placement best_pl;
    {
      int cnt_ge[5] = {0, 0, 0, 0, 0};   // visits with at least k rows
      int rmax = 0;
      for (const V& v : vv) { for (int k = 1; k <= v.r; k++) cnt_ge[k]++; rmax = std::max(rmax, v.r); }
      struct Cand { int cost; int prof[4]; };
      std::vector<Cand> cands;
      const int caps[2] = {4 * std::min(16, (int)cl.size()), 64};
      for (int pass = 0; pass < 2 && cands.empty(); pass++)
        for (int a = rmax; a <= rmax; a++)
          for (int b = 0; b <= a; b++)
            for (int c2 = 0; c2 <= b; c2++)
              for (int d2 = 0; d2 <= c2; d2++) {
                const int prof[4] = {a, b, c2, d2};
                bool fits = true;
                for (int k = 1; k <= 4 && fits; k++) {
                  int slots = 0;
                  for (int wv = 0; wv < 4; wv++) if (prof[wv] >= k) slots += caps[pass];
                  fits = cnt_ge[k] <= slots;
                }
                if (!fits) continue;
                Cand cd; cd.cost = 0;
                for (int wv = 0; wv < 4; wv++) { cd.prof[wv] = prof[wv]; cd.cost += prefix_cost(prof[wv]); }
                cands.push_back(cd);
              }
      std::stable_sort(cands.begin(), cands.end(), [](const Cand& x, const Cand& y) { return x.cost < y.cost; });
      long best_total = -1;
      int tried = 0;
      for (const Cand& cd : cands) {
        if (best_total >= 0 && cd.cost >= best_total) break;
        if (tried++ >= 6) break;
        Placement pl;
        try_profile(cd.prof, pl);
        if (!pl.ok) continue;
        const long total = pl.instr + HostPrepEv::COLLISION_COST * pl.coll;
        if (best_total < 0 || total < best_total) { best_total = total; best_pl = std::move(pl); }
        if (best_pl.coll == 0) break;
      }
      if (best_total < 0) {   // no profile could take the visits (cannot happen: (rmax, rmax, rmax, rmax) holds 256 lanes)
        const int prof[4] = {4, 4, 4, 4};
        try_profile(prof, best_pl);
      }
    }
    if (!best_pl.ok) wfail = true;
    std::vector<int> fill_g(16, 0), maxr_g(16, 0);
    for (int x = 0; x < nlanes && !wfail; x++) {
      const V& v = vv[(size_t)x];
      const int best_g = best_pl.grp[(size_t)x];
      const int* pm = PERM4[PERM_R[v.r][(int)best_pl.q[(size_t)x]]];
      int perm[4];
      for (int i = 0; i < v.r; i++) perm[i] = v.own[pm[i]];
      for (int j = v.r; j < 4; j++) perm[j] = v.rest[j - v.r];
      maxr_g[best_g] = std::max(maxr_g[best_g], v.r);
      const size_t lane = (size_t)best_g * 16 + (size_t)fill_g[best_g]++;
      uint32_t packed = 0;
      uint32_t li[4];
      for (int j = 0; j < 4; j++) {
        const uint32_t m = conn[(int64_t)v.e * 4 + perm[j]];
        const uint32_t pos = (uint32_t)(std::find(t.begin(), t.end(), m) - t.begin());
        li[j] = pos;
        packed |= pos << (8 * j);
      }
      E.vloc[(size_t)w * BLOCK + lane] = packed;
      rows_w[(size_t)w] += v.r;
      for (int i = 0; i < v.r; i++) {
        uint32_t half = 0;   // four 4-bit slots (a row has at most 16 node blocks): 16 bits per row, two rows per word
        for (int j = 0; j < 4; j++) {
          const uint32_t sl = P.eslot[(size_t)v.e * 16 + (size_t)perm[i] * 4 + (size_t)perm[j]];
          if (sl > 15) wfail = true;
          half |= (sl & 15u) << (4 * j);
        }
        E.vslot[((size_t)w * BLOCK + lane) * 2 + (size_t)(i >> 1)] |= half << (16 * (i & 1));
        if (li[i] >= cl.size()) wfail = true;
      }
    }
    conf_w[(size_t)w] = best_pl.coll;
    instr_w[(size_t)w] = best_pl.instr;
    for (int g = 0; g < 16; g++) pass_w[(size_t)w] += maxr_g[g];
    if (wfail) {
#pragma omp atomic write
      fail = 1;
    }
  }
  if (fail) return "internal: element-visit list construction failed";
  for (int64_t w = 0; w < nwg; w++) {
    E.max_out_doubles = std::max(E.max_out_doubles, img_w[(size_t)w]);
    E.n_visits += (int64_t)E.desc[(size_t)w].nvis;
    E.n_rows += rows_w[(size_t)w];
    E.n_conflicts += conf_w[(size_t)w];
    E.n_group_rows += pass_w[(size_t)w];
    E.n_pass_instr += instr_w[(size_t)w];
  }
  E.ok = true;
  return std::string();
}

}  // namespace rdc
