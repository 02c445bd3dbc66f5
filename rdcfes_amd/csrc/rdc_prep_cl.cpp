// rdc_prep_cl.cpp — host preparation of the producer / consumer cluster kernels (rdc_solid_cl.hip): node clusters, their
// element lists (producer lanes) and (node, element) pair lists (consumer lanes).
//
// Clusters are grown greedily over the mesh graph exactly as for the element-visit kernel (rdc_prep_ev.cpp): seed = the
// unassigned owned node with the fewest unassigned neighbours (clusters grow along the front of what is assigned), then
// the unassigned node sharing the most elements with the cluster joins until a limit binds (nodes, pairs, distinct
// elements, LDS image of the cluster's CSR rows).  The fewer distinct elements per owned node, the less per-point work the
// producers redo (a HEX8 brick of 4 x 3 x 2 nodes touches 60 elements = 2.5 per node, against 8 pairs per node).
//
// Pair order.  Two LDS access patterns depend on it:
//  (a) a consumer lane reads the 47-double point record of its element at every point (376 ds_read_b64 per pair, the bulk
//      of the kernel's LDS traffic).  Records are 49 doubles apart, so the records of 32 CONSECUTIVE local elements start in
//      32 different double-banks; lanes reading the same record broadcast;
//  (b) the consumers add their rows into the LDS image with ds_add_f64, executed in four groups of 16 lanes; lanes of a
//      group hitting the same address (same node, same column) or the same double-bank serialise (tools/lds_bank_model.hip).
// pair_order 1 (default): elements sorted by (colour, id) -- elements of one colour share no node -- and the pairs listed
// element after element: the 32 lanes of a read pass cover a run of consecutive records (conflict-free), and the ~5
// elements of a 16-lane group are node-disjoint, so no two lanes of an atomic pass share a row.
// pair_order 0: "k-th pair of every node" major (16 consecutive pairs = 16 different, consecutive nodes: atomics free of
// conflicts, but a read pass then touches ~20 scattered records and 39% of the LDS cycles were bank conflicts).
#include <algorithm>
#include <cstring>

#include "rdc_prep.h"

namespace rdc {

std::string prep_build_cl(const HostPrep& P, const uint32_t* conn, const HostPrepCl::Limits& lim, HostPrepCl& C, int64_t n_interior) {
  C = HostPrepCl();
  C.lim = lim;
  C.nvar = P.nvar;
  const int nen = P.nen;
  const int64_t nv2 = (int64_t)P.nvar * P.nvar;
  const int64_t ipb = lim.img_per_block > 0 ? lim.img_per_block : nv2;   // image doubles per node block
  const int64_t pad = ipb == nv2 ? 1 : 0;   // whole-row images: up to one double per node to give its segment the 16-byte phase of its CSR segment
  if (nen != 8 && nen != 4) return "cluster lists: TET4 / HEX8 only";
  const int64_t n_elem = P.n_elem, n_node = P.n_node, n_owned = P.n_owned;
  if (n_owned <= 0) return "no owned nodes";
  if (P.bptr[(size_t)n_owned] >= ((int64_t)1 << 32)) return "more than 2^32 node blocks";
  if (lim.max_nodes > 255 || lim.max_elems > 255 || lim.max_pairs > 65535) return "cluster limits out of range";
  std::vector<int64_t> inc_ptr((size_t)n_node + 1, 0);
  for (int64_t x = 0; x < n_elem * nen; x++) inc_ptr[conn[x] + 1]++;
  for (int64_t n = 0; n < n_node; n++) inc_ptr[n + 1] += inc_ptr[n];
  std::vector<uint32_t> inc((size_t)inc_ptr[n_node]);
  {
    std::vector<int64_t> fill(inc_ptr.begin(), inc_ptr.end() - 1);
    for (int64_t e = 0; e < n_elem; e++)
      for (int i = 0; i < nen; i++) inc[fill[conn[e * nen + i]]++] = (uint32_t)e;   // ascending element ids per node
  }
  for (int64_t n = 0; n < n_owned; n++) {
    if (P.bptr[n + 1] - P.bptr[n] > 255) return "a row has more than 255 node blocks";
    if (inc_ptr[n + 1] - inc_ptr[n] > lim.max_pairs || inc_ptr[n + 1] - inc_ptr[n] > lim.max_elems) return "a node has more incident elements than a cluster may hold";
    if (ipb * (P.bptr[n + 1] - P.bptr[n]) + pad > lim.max_row_doubles) return "the rows of one node exceed the image budget";
  }
  // ---- greedy clustering ------------------------------------------------------------------------------------------
  std::vector<int32_t> cluster_of((size_t)n_owned, -1);   // -1 unassigned, -2 rejected for the cluster being grown
  std::vector<std::vector<uint32_t>> clusters;
  {
    std::vector<uint32_t> emark((size_t)n_elem, 0), gain((size_t)n_owned, 0), gstamp((size_t)n_owned, 0);
    uint32_t stamp = 0;
    std::vector<uint32_t> cand, rejected;
    std::vector<int32_t> free_nb((size_t)n_owned, 0);
    int max_nb = 0;
    for (int64_t n = 0; n < n_owned; n++) {
      int c = 0;
      for (int64_t b = P.bptr[n]; b < P.bptr[n + 1]; b++) c += (P.bcol[b] != (int32_t)n && (int64_t)P.bcol[b] < n_owned);
      free_nb[n] = c;
      max_nb = std::max(max_nb, c);
    }
    std::vector<std::vector<uint32_t>> bucket((size_t)max_nb + 1);
    for (int64_t n = n_owned - 1; n >= 0; n--) bucket[(size_t)free_nb[n]].push_back((uint32_t)n);
    auto assigned_update = [&](uint32_t n) {
      for (int64_t b = P.bptr[n]; b < P.bptr[n + 1]; b++) {
        const int32_t m = P.bcol[b];
        if (m == (int32_t)n || (int64_t)m >= n_owned || cluster_of[m] >= 0) continue;
        free_nb[m]--;
        bucket[(size_t)free_nb[m]].push_back((uint32_t)m);
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
      int64_t npair = 0, nel = 0, img = 0;
      cand.clear();
      rejected.clear();
      auto new_elems = [&](uint32_t n) {
        int64_t d = 0;
        for (int64_t k = inc_ptr[n]; k < inc_ptr[n + 1]; k++) d += (emark[inc[k]] != stamp);
        return d;
      };
      auto add = [&](uint32_t n) {
        cluster_of[n] = (int32_t)clusters.size();
        cl.push_back(n);
        n_assigned++;
        assigned_update(n);
        npair += inc_ptr[n + 1] - inc_ptr[n];
        img += ipb * (P.bptr[n + 1] - P.bptr[n]) + pad;
        for (int64_t k = inc_ptr[n]; k < inc_ptr[n + 1]; k++) {
          const uint32_t e = inc[k];
          if (emark[e] == stamp) continue;
          emark[e] = stamp;
          nel++;
          for (int j = 0; j < nen; j++) {
            const uint32_t m = conn[(int64_t)e * nen + j];
            // two-part assembly: a cluster never mixes interior nodes (rows assembled before the halo exchange has landed) with the others
            if ((int64_t)m < n_owned && cluster_of[m] == -1 && (n_interior < 0 || (((int64_t)m < n_interior) == (seed < n_interior)))) {
              if (gstamp[m] != stamp) { gstamp[m] = stamp; gain[m] = 0; cand.push_back(m); }
              gain[m]++;
            }
          }
        }
      };
      add((uint32_t)seed);
      while ((int)cl.size() < lim.max_nodes) {
        int best = -1;
        for (size_t x = 0; x < cand.size(); x++) {
          const uint32_t c = cand[x];
          if (cluster_of[c] != -1) continue;
          if (best < 0) { best = (int)x; continue; }
          const uint32_t bc = cand[(size_t)best];
          if (gain[c] > gain[bc] || (gain[c] == gain[bc] && (free_nb[c] < free_nb[bc] || (free_nb[c] == free_nb[bc] && c < bc)))) best = (int)x;
        }
        if (best < 0) break;
        const uint32_t c = cand[(size_t)best];
        if (npair + (inc_ptr[c + 1] - inc_ptr[c]) > lim.max_pairs || nel + new_elems(c) > lim.max_elems ||
            img + ipb * (P.bptr[c + 1] - P.bptr[c]) + pad > lim.max_row_doubles) {
          cluster_of[c] = -2;
          rejected.push_back(c);
          continue;
        }
        add(c);
      }
      for (uint32_t c : rejected) cluster_of[c] = -1;
      clusters.push_back(std::move(cl));
    }
  }
  // two-part assembly: the clusters of interior nodes first (the kernels launch a leading / trailing sub-range of the lists);
  // cluster_of follows the new numbering
  if (n_interior >= 0) {
    std::stable_partition(clusters.begin(), clusters.end(), [&](const std::vector<uint32_t>& cl) { return (int64_t)cl[0] < n_interior; });
    int64_t bound = n_interior;
    for (size_t w = 0; w < clusters.size(); w++) {
      const bool interior = (int64_t)clusters[w][0] < n_interior;
      if (interior) C.n_wg_interior = (int64_t)w + 1;
      for (uint32_t n : clusters[w]) {
        cluster_of[n] = (int32_t)w;
        if (!interior) bound = std::min<int64_t>(bound, (int64_t)n);
      }
    }
    C.part1_nodes = bound;   // rows [0, bound) are complete once the interior clusters have run
  }
  const int64_t nwg = (int64_t)clusters.size();
  // ---- per-workgroup lists ------------------------------------------------------------------------------------------
  C.desc.assign((size_t)nwg, HostPrepCl::Desc{0, 0, 0, 0, 0, 0});
  C.ntab.assign((size_t)nwg * lim.max_nodes, HostPrepCl::Node{0, 0, 0, 0, 0});
  C.eid.assign((size_t)nwg * lim.max_elems, 0xFFFFFFFFu);
  C.pair.assign((size_t)nwg * lim.max_pairs, 0xFFFFFFFFu);
  const int wpp = nen / 4;
  C.pslot.assign((size_t)nwg * lim.max_pairs * wpp, 0);
  int fail = 0;
  std::vector<uint32_t> rowd((size_t)nwg, 0);
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t w = 0; w < nwg; w++) {
    const std::vector<uint32_t>& cl = clusters[(size_t)w];
    std::vector<uint32_t> el;
    size_t maxinc = 0;
    for (uint32_t n : cl) {
      for (int64_t k = inc_ptr[n]; k < inc_ptr[n + 1]; k++) el.push_back(inc[k]);
      maxinc = std::max(maxinc, (size_t)(inc_ptr[n + 1] - inc_ptr[n]));
    }
    std::sort(el.begin(), el.end());
    el.erase(std::unique(el.begin(), el.end()), el.end());
    if (lim.pair_order == 1)
      std::sort(el.begin(), el.end(), [&](uint32_t x, uint32_t y) { return P.colour[x] != P.colour[y] ? P.colour[x] < P.colour[y] : x < y; });
    if ((int)el.size() > lim.max_elems || (int)cl.size() > lim.max_nodes) { fail = 1; continue; }
    HostPrepCl::Desc& d = C.desc[(size_t)w];
    d.nown = (uint16_t)cl.size();
    d.nelem = (uint16_t)el.size();
    std::copy(el.begin(), el.end(), C.eid.begin() + (size_t)w * lim.max_elems);
    uint32_t off = 0;
    for (size_t a = 0; a < cl.size(); a++) {
      const uint32_t n = cl[a];
      HostPrepCl::Node& nd = C.ntab[(size_t)w * lim.max_nodes + a];
      const int64_t len = P.bptr[n + 1] - P.bptr[n];
      // whole-row images: the node's segment gets the 16-byte phase its CSR segment has in memory (copied out with 16-byte stores)
      if (ipb == nv2 && ((off ^ (uint32_t)(nv2 * P.bptr[n])) & 1u)) off++;
      nd.bptr = (uint32_t)P.bptr[n]; nd.len = (uint16_t)len; nd.off = (uint16_t)off; nd.node = n;
      off += (uint32_t)(ipb * len);
    }
    if (off > 0xFFFFu) { fail = 1; continue; }
    d.row_doubles = off;
    rowd[(size_t)w] = off;
    uint32_t np = 0;
    auto emit = [&](uint32_t le, int li, size_t a) {
      const uint32_t e = el[le];
      if (np >= (uint32_t)lim.max_pairs) { fail = 1; return; }
      C.pair[(size_t)w * lim.max_pairs + np] = le | ((uint32_t)li << 8) | ((uint32_t)a << 16);
      for (int j = 0; j < nen; j++) {
        const uint32_t s = P.eslot[(size_t)e * nen * nen + (size_t)li * nen + j];
        if (s > 255) fail = 1;
        C.pslot[((size_t)w * lim.max_pairs + np) * wpp + j / 4] |= (s & 0xFFu) << (8 * (j % 4));
      }
      np++;
    };
    if (lim.pair_order == 1) {
      // element after element (a node listed twice in one element is not a valid mesh: one pair per (element, local node))
      for (uint32_t le = 0; le < (uint32_t)el.size(); le++)
        for (int li = 0; li < nen; li++) {
          const uint32_t n = conn[(int64_t)el[le] * nen + li];
          if ((int64_t)n >= n_owned || cluster_of[n] != (int32_t)w) continue;
          emit(le, li, (size_t)(std::find(cl.begin(), cl.end(), n) - cl.begin()));
        }
    } else {
      // k-th incident element of every node, k-major
      for (size_t k = 0; k < maxinc; k++)
        for (size_t a = 0; a < cl.size(); a++) {
          const uint32_t n = cl[a];
          if ((int64_t)k >= inc_ptr[n + 1] - inc_ptr[n]) continue;
          const uint32_t e = inc[inc_ptr[n] + (int64_t)k];
          int li = -1;
          for (int j = 0; j < nen; j++) if (conn[(int64_t)e * nen + j] == n) { li = j; break; }
          const uint32_t le = (uint32_t)(std::find(el.begin(), el.end(), e) - el.begin());
          if (li < 0) { fail = 1; break; }
          emit(le, li, a);
        }
    }
    d.npair = (uint16_t)np;
  }
  if (fail) return "internal: cluster list construction failed";
  for (int64_t w = 0; w < nwg; w++) {
    C.max_row_doubles = std::max(C.max_row_doubles, (size_t)rowd[(size_t)w]);
    C.n_elem_visits += C.desc[(size_t)w].nelem;
    C.n_pairs += C.desc[(size_t)w].npair;
  }
  C.ok = true;
  return std::string();
}

}  // namespace rdc
