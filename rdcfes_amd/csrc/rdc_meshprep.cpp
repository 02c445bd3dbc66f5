// rdc_meshprep.cpp — one-time host preparation of a mesh partition:
//   * node-block CSR pattern of the owned rows (the sparsity libMesh's es.init() builds,
//     src/pihna.C:48), dof = node*nvar + var
//   * per-element slot map (where Ke(i,j) lands in row i) -- replaces MatSetValues' row search
//   * greedy element colouring + first-writer masks (coloured scatter needs no memset, no atomics)
//   * (node, incident element) pairs and workgroup ranges for the row-gather kernels
#include <algorithm>
#include <array>
#include <cstring>
#include <numeric>

#include "rdc_prep.h"

namespace rdc {

std::string prep_build(int nen, int64_t n_elem, int64_t n_node, int64_t n_owned, const uint32_t* conn,
                       int nvar, size_t lds_budget_bytes, int block, HostPrep& P, bool conflict_aware) {
  if (nen != 4 && nen != 8) return "element type must be TET4 (4) or HEX8 (8)";
  if (n_elem < 0 || n_node <= 0 || n_owned < 0 || n_owned > n_node) return "bad mesh sizes";
  if (n_node >= (int64_t)1 << 31) return "more than 2^31 local nodes not supported";
  if (n_elem >= (int64_t)1 << 32) return "more than 2^32 local elements not supported";
  if (nvar < 1 || nvar > 8) return "nvar must be in 1..8";
  P = HostPrep();
  P.nen = nen; P.nvar = nvar; P.n_elem = n_elem; P.n_node = n_node; P.n_owned = n_owned;
  for (int64_t x = 0; x < n_elem * nen; x++)
    if (conn[x] >= (uint64_t)n_node) return "connectivity entry out of range";
  for (int64_t e = 0; e < n_elem; e++)
    for (int i = 0; i < nen; i++)
      for (int j = i + 1; j < nen; j++)
        if (conn[e * nen + i] == conn[e * nen + j]) return "degenerate element (repeated node)";

  // ---- node -> incident (element, local index) lists, counting sort ------------------------
  std::vector<int64_t> inc_ptr((size_t)n_node + 1, 0);
  for (int64_t x = 0; x < n_elem * nen; x++) inc_ptr[conn[x] + 1]++;
  for (int64_t n = 0; n < n_node; n++) inc_ptr[n + 1] += inc_ptr[n];
  std::vector<uint32_t> inc_elem((size_t)inc_ptr[n_node]);
  std::vector<uint8_t> inc_loc((size_t)inc_ptr[n_node]);
  {
    std::vector<int64_t> fill(inc_ptr.begin(), inc_ptr.end() - 1);
    for (int64_t e = 0; e < n_elem; e++)
      for (int i = 0; i < nen; i++) {
        const int64_t n = conn[e * nen + i];
        inc_elem[fill[n]] = (uint32_t)e;
        inc_loc[fill[n]] = (uint8_t)i;
        fill[n]++;
      }
  }

  // ---- node-block pattern of the owned rows -------------------------------------------------
  P.bptr.assign((size_t)n_owned + 1, 0);
  {
    std::vector<int64_t> len((size_t)n_owned, 0);
#pragma omp parallel
    {
      std::vector<int32_t> tmp;
#pragma omp for schedule(dynamic, 4096)
      for (int64_t n = 0; n < n_owned; n++) {
        tmp.clear();
        tmp.push_back((int32_t)n);
        for (int64_t k = inc_ptr[n]; k < inc_ptr[n + 1]; k++) {
          const uint32_t* c = conn + (int64_t)inc_elem[k] * nen;
          for (int i = 0; i < nen; i++) tmp.push_back((int32_t)c[i]);
        }
        std::sort(tmp.begin(), tmp.end());
        len[n] = std::unique(tmp.begin(), tmp.end()) - tmp.begin();
      }
    }
    for (int64_t n = 0; n < n_owned; n++) P.bptr[n + 1] = P.bptr[n] + len[n];
    P.bcol.resize((size_t)P.bptr[n_owned]);
#pragma omp parallel
    {
      std::vector<int32_t> tmp;
#pragma omp for schedule(dynamic, 4096)
      for (int64_t n = 0; n < n_owned; n++) {
        tmp.clear();
        tmp.push_back((int32_t)n);
        for (int64_t k = inc_ptr[n]; k < inc_ptr[n + 1]; k++) {
          const uint32_t* c = conn + (int64_t)inc_elem[k] * nen;
          for (int i = 0; i < nen; i++) tmp.push_back((int32_t)c[i]);
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        std::copy(tmp.begin(), tmp.end(), P.bcol.begin() + P.bptr[n]);
      }
    }
  }
  int64_t max_len = 0;
  for (int64_t n = 0; n < n_owned; n++) max_len = std::max(max_len, P.bptr[n + 1] - P.bptr[n]);
  if (max_len >= 0xFFFF) return "node valence too large for 16-bit slot map";
  if ((int64_t)nvar * nvar * P.bptr[n_owned] >= ((int64_t)1 << 40)) return "matrix too large";

  // ---- slot map ----------------------------------------------------------------------------
  P.eslot.assign((size_t)n_elem * nen * nen, 0xFFFF);
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < n_elem; e++) {
    const uint32_t* c = conn + e * nen;
    for (int i = 0; i < nen; i++) {
      const int64_t I = c[i];
      if (I >= n_owned) continue;
      const int32_t* b = P.bcol.data() + P.bptr[I];
      const int32_t* be = P.bcol.data() + P.bptr[I + 1];
      for (int j = 0; j < nen; j++) {
        const int32_t* it = std::lower_bound(b, be, (int32_t)c[j]);
        P.eslot[(size_t)e * nen * nen + i * nen + j] = (uint16_t)(it - b);
      }
    }
  }

  // ---- greedy colouring: elements of one colour share no node ----------------------------
  {
    constexpr int W = 4;  // up to 256 colours
    std::vector<uint64_t> used((size_t)n_node * W, 0);
    P.colour.assign((size_t)n_elem, -1);
    int ncol = 0;
    for (int64_t e = 0; e < n_elem; e++) {
      const uint32_t* c = conn + e * nen;
      uint64_t m[W] = {0, 0, 0, 0};
      for (int i = 0; i < nen; i++)
        for (int w = 0; w < W; w++) m[w] |= used[(size_t)c[i] * W + w];
      int col = -1;
      for (int w = 0; w < W && col < 0; w++)
        if (~m[w]) col = w * 64 + __builtin_ctzll(~m[w]);
      if (col < 0) return "mesh needs more than 256 colours";
      P.colour[e] = col;
      ncol = std::max(ncol, col + 1);
      for (int i = 0; i < nen; i++) used[(size_t)c[i] * W + (col >> 6)] |= 1ull << (col & 63);
    }
    P.n_colours = ncol;
    P.colour_ptr.assign((size_t)ncol + 1, 0);
    for (int64_t e = 0; e < n_elem; e++) P.colour_ptr[P.colour[e] + 1]++;
    for (int c = 0; c < ncol; c++) P.colour_ptr[c + 1] += P.colour_ptr[c];
    P.elem_order.resize((size_t)n_elem);
    std::vector<int64_t> fill(P.colour_ptr.begin(), P.colour_ptr.end() - 1);
    for (int64_t e = 0; e < n_elem; e++) P.elem_order[fill[P.colour[e]]++] = (uint32_t)e;
  }

  // ---- first-writer masks in colour order --------------------------------------------------
  {
    P.first_mask.assign((size_t)n_elem, 0);
    P.first_rhs.assign((size_t)n_elem, 0);
    std::vector<uint8_t> touched((size_t)P.bptr[n_owned], 0), touched_r((size_t)n_owned, 0);
    for (int64_t x = 0; x < n_elem; x++) {
      const int64_t e = P.elem_order[x];
      const uint32_t* c = conn + e * nen;
      uint64_t fm = 0;
      uint8_t fr = 0;
      for (int i = 0; i < nen; i++) {
        const int64_t I = c[i];
        if (I >= n_owned) continue;
        if (!touched_r[I]) { touched_r[I] = 1; fr |= (uint8_t)(1u << i); }
        for (int j = 0; j < nen; j++) {
          const int64_t pos = P.bptr[I] + P.eslot[(size_t)e * nen * nen + i * nen + j];
          if (!touched[pos]) { touched[pos] = 1; fm |= 1ull << (i * nen + j); }
        }
      }
      P.first_mask[e] = fm;
      P.first_rhs[e] = fr;
    }
    // every owned row must be reached by at least one element, or it would stay unwritten
    for (int64_t n = 0; n < n_owned; n++)
      if (!touched_r[n]) return "owned node without any incident element";
  }

  // ---- row-gather pairs and workgroup ranges ------------------------------------------------
  {
    P.node_pair_ptr.assign((size_t)n_owned + 1, 0);
    for (int64_t n = 0; n < n_owned; n++) P.node_pair_ptr[n + 1] = inc_ptr[n + 1];  // owned nodes come first
    const int64_t npairs = inc_ptr[n_owned];
    P.pair_elem.assign(inc_elem.begin(), inc_elem.begin() + npairs);
    P.pair_local.assign(inc_loc.begin(), inc_loc.begin() + npairs);
    P.rg_block = block;
    P.rowgather_ok = true;
    P.wg_node_ptr.clear();
    P.wg_node_ptr.push_back(0);
    int64_t pairs = 0;
    size_t bytes = 0, max_bytes = 0;
    for (int64_t n = 0; n < n_owned; n++) {
      const int64_t np = inc_ptr[n + 1] - inc_ptr[n];
      const size_t nb = sizeof(double) * ((size_t)nvar * nvar * (P.bptr[n + 1] - P.bptr[n]) + nvar);
      if (nb > lds_budget_bytes) { P.rowgather_ok = false; break; }
      if (n > P.wg_node_ptr.back() && (pairs + np > block || bytes + nb > lds_budget_bytes)) {
        P.wg_node_ptr.push_back((int32_t)n);
        pairs = 0;
        bytes = 0;
      }
      pairs += np;
      bytes += nb;
      max_bytes = std::max(max_bytes, bytes);
    }
    P.wg_node_ptr.push_back((int32_t)n_owned);
    P.rg_lds_bytes = max_bytes;
    if (!P.rowgather_ok) { P.wg_node_ptr.assign(2, 0); P.wg_node_ptr[1] = 0; }
    // node lists of the workgroups for the node-staged kernel (HEX8 only: TET4 has its own factored path)
    P.hx_ok = false;
    P.hx_nl_ptr.clear(); P.hx_nlist.clear(); P.hx_ploc.clear(); P.hx_max_nodes = 0;
    if (P.rowgather_ok && nen == 8) {
      const int64_t nwg = (int64_t)P.wg_node_ptr.size() - 1;
      std::vector<std::vector<uint32_t>> lists((size_t)nwg);
      P.hx_ploc.assign((size_t)npairs * nen, 0);
      int bad = 0;
#pragma omp parallel for schedule(dynamic, 64)
      for (int64_t w = 0; w < nwg; w++) {
        const int64_t p0 = P.node_pair_ptr[P.wg_node_ptr[w]], p1 = P.node_pair_ptr[P.wg_node_ptr[w + 1]];
        std::vector<uint32_t>& l = lists[(size_t)w];
        l.reserve((size_t)(p1 - p0) * nen);
        for (int64_t p = p0; p < p1; p++)
          for (int j = 0; j < nen; j++) l.push_back(conn[(int64_t)P.pair_elem[p] * nen + j]);
        std::sort(l.begin(), l.end());
        l.erase(std::unique(l.begin(), l.end()), l.end());
        if (l.size() > 0xFFFF) { bad = 1; continue; }
        for (int64_t p = p0; p < p1; p++)
          for (int j = 0; j < nen; j++)
            P.hx_ploc[(size_t)p * nen + j] =
                (uint16_t)(std::lower_bound(l.begin(), l.end(), conn[(int64_t)P.pair_elem[p] * nen + j]) - l.begin());
      }
      if (!bad) {
        P.hx_nl_ptr.assign((size_t)nwg + 1, 0);
        for (int64_t w = 0; w < nwg; w++) {
          P.hx_nl_ptr[(size_t)w + 1] = P.hx_nl_ptr[(size_t)w] + (int64_t)lists[(size_t)w].size();
          P.hx_max_nodes = std::max(P.hx_max_nodes, (int)lists[(size_t)w].size());
        }
        P.hx_nlist.resize((size_t)P.hx_nl_ptr[(size_t)nwg]);
        for (int64_t w = 0; w < nwg; w++)
          std::copy(lists[(size_t)w].begin(), lists[(size_t)w].end(), P.hx_nlist.begin() + P.hx_nl_ptr[(size_t)w]);
        P.hx_ok = true;
      } else {
        P.hx_ploc.clear();
      }
    }
  }

  // ---- staged row gather: workgroups of <= block pairs, flat descriptors, balanced chunks ------
  {
    constexpr int CH = HostPrep::RG2_CHUNK;
    P.rg2_block = block;
    P.rg2_ok = n_owned > 0;
    // contributions per node block (= elements containing the edge / all incident elements on the
    // diagonal); needed up front to size the workgroups
    std::vector<uint8_t> bcnt((size_t)P.bptr[n_owned], 0);
    for (int64_t p = 0; p < inc_ptr[n_owned] && P.rg2_ok; p++) {
      const int64_t e = inc_elem[p];
      const int i = inc_loc[p];
      const int64_t I = conn[e * nen + i];
      for (int jo = 0; jo < nen; jo++) {
        uint8_t& c = bcnt[P.bptr[I] + P.eslot[(size_t)e * nen * nen + i * nen + jo]];
        if (c == 255) { P.rg2_ok = false; break; }
        c++;
      }
    }
    std::vector<int64_t> wgp;  // node ranges
    wgp.push_back(0);
    int64_t pairs = 0, chunks = 0;
    size_t bytes = 0, max_bytes = 0;
    for (int64_t n = 0; n < n_owned && P.rg2_ok; n++) {
      const int64_t np = inc_ptr[n + 1] - inc_ptr[n];
      const int64_t len = P.bptr[n + 1] - P.bptr[n];
      // row slice of the node; the private diagonal accumulators are a fixed-size area
      const size_t diag_area = sizeof(double) * (size_t)HostPrep::RG3_DIAG_SLOTS * (nvar * nvar + nvar);
      const size_t nbytes = sizeof(double) * ((size_t)nvar * nvar * len);
      if (nbytes + diag_area > lds_budget_bytes) { P.rg2_ok = false; break; }
      int64_t nch = 0;
      for (int64_t b = P.bptr[n]; b < P.bptr[n + 1]; b++) nch += std::max<int64_t>(1, (bcnt[b] + CH - 1) / CH);
      if (np > block || nch + 1 > block || len > 255) { P.rg2_ok = false; break; }
      // gather work items of a workgroup: its chunks plus one rhs item per node
      if (n > wgp.back() && (pairs + np > block || chunks + nch + (n - wgp.back() + 1) > block || n - wgp.back() >= 255 ||
                             n - wgp.back() >= HostPrep::RG3_DIAG_SLOTS / HostPrep::rg3_diag_copies(block) ||
                             bytes + nbytes + diag_area > lds_budget_bytes)) {
        wgp.push_back(n);
        pairs = 0;
        chunks = 0;
        bytes = 0;
      }
      pairs += np;
      chunks += nch;
      bytes += nbytes;
      max_bytes = std::max(max_bytes, bytes + diag_area);
    }
    P.rg2_lds_bytes = max_bytes;
    wgp.push_back(n_owned);
    if (P.rg2_ok) {
      const int64_t nwg = (int64_t)wgp.size() - 1;
      // stage row of a pair: [column 0 (diagonal block): nvar values][rhs value][columns 1..nen-1: nvar each];
      // odd padding keeps ds_write_b64 conflict-free
      const int stride = (nen * nvar + 1) | 1;
      P.wg2.resize((size_t)nwg);
      P.pair_rec.assign((size_t)nwg * block * nen, 0xFFFFFFFFu);
      if (nen == 4) P.pair_eid.assign((size_t)nwg * block, 0u);
      if (nen == 4) P.pair_aux.assign((size_t)nwg * block * 8, 0);
      P.node_tab.assign((size_t)n_owned * 4, 0);
      P.sdesc.resize((size_t)P.bptr[n_owned]);
      P.contrib.assign((size_t)inc_ptr[n_owned] * nen + (size_t)block * nen, 0);  // tail pad: vector loads may overrun
      // chunk offsets per workgroup
      std::vector<int64_t> ch0((size_t)nwg + 1, 0);
      for (int64_t w = 0; w < nwg; w++) {
        int64_t nch = 0;
        for (int64_t b = P.bptr[wgp[w]]; b < P.bptr[wgp[w + 1]]; b++) nch += std::max<int64_t>(1, (bcnt[b] + CH - 1) / CH);
        ch0[w + 1] = ch0[w] + nch;
      }
      P.chunk.resize((size_t)ch0[nwg]);
      int fail_flag = 0;
#pragma omp parallel for schedule(dynamic, 256)
      for (int64_t w = 0; w < nwg; w++) {
        const int64_t n0 = wgp[w], n1 = wgp[w + 1];
        HostPrep::WgDesc& d = P.wg2[w];
        d.n0 = (int32_t)n0; d.nnodes = (int32_t)(n1 - n0);
        d.bb0 = P.bptr[n0]; d.vb0 = (int64_t)nvar * nvar * d.bb0;
        d.nb = (int32_t)(P.bptr[n1] - d.bb0);
        d.np = (int32_t)(inc_ptr[n1] - inc_ptr[n0]);
        d.c0 = inc_ptr[n0] * nen;
        d.ch0 = ch0[w];
        d.nch = (int32_t)(ch0[w + 1] - ch0[w]);
        d.pad = 0;
        // contribution ranges per block (prefix of bcnt), chunks, store descriptors
        std::vector<int32_t> cbeg((size_t)d.nb + 1, 0), fill((size_t)d.nb, 0);
        for (int b = 0; b < d.nb; b++) cbeg[b + 1] = cbeg[b] + bcnt[d.bb0 + b];
        int64_t ck = d.ch0;
        int extra = d.nb;  // partial-sum slots: [0, nb) one per block, then the extra chunks
        for (int64_t n = n0; n < n1; n++) {
          const int64_t len = P.bptr[n + 1] - P.bptr[n];
          {
            const int32_t* bc = P.bcol.data() + P.bptr[n];
            const int64_t dslot = std::lower_bound(bc, bc + len, (int32_t)n) - bc;
            uint16_t* nt = &P.node_tab[(size_t)n * 4];
            nt[0] = (uint16_t)((int64_t)nvar * nvar * (P.bptr[n] - d.bb0));
            nt[1] = (uint16_t)(nvar * len);
            nt[2] = (uint16_t)(nvar * dslot);
            nt[3] = 0;
          }
          for (int64_t s2 = 0; s2 < len; s2++) {
            const int64_t gb = P.bptr[n] + s2;
            const int lb = (int)(gb - d.bb0);
            HostPrep::StoreDesc& sd = P.sdesc[gb];
            const int64_t off = (int64_t)nvar * nvar * (P.bptr[n] - d.bb0) + nvar * s2;
            if (off > 0xFFFF) fail_flag = 1;
            sd.outoff = (uint16_t)off; sd.len = (uint8_t)len;
            sd.diag = (P.bcol[gb] == (int32_t)n) ? 1 : 0;
            sd.node = (uint8_t)(n - n0);
            const int cnt = bcnt[gb];
            const int nch = std::max(1, (cnt + CH - 1) / CH);
            sd.nextra = (uint8_t)(nch - 1);
            sd.extra = (uint16_t)extra;
            for (int c = 0; c < nch; c++) {
              HostPrep::Chunk& k = P.chunk[ck++];
              k.cbeg = (uint16_t)(cbeg[lb] + c * CH);
              k.cnt = (uint16_t)std::min(CH, cnt - c * CH);
              if (cnt == 0) k.cnt = 0;
              k.dst = (uint16_t)(c == 0 ? lb : extra++);
              k.pad = 0;
            }
          }
        }
        d.nout = extra;
        if (extra > block) fail_flag = 1;
        // ---- LDS-conflict-aware schedule (TET4) ---------------------------------------------------
        // The kernel maps pair index idx -> (wave = idx % NW, lane = idx / NW).  An FP64 LDS atomic costs
        // 6 CU-cycles per wave-instruction on distinct banks, ~19 on random addresses and >50 when lanes
        // share an address (tools/lds_atomic_bench.hip), so the host chooses, per pair, the wave and the
        // ORDER of its three off-diagonal columns (any vertex permutation of a tet is legal) that adds the
        // fewest bank / address collisions to the three off-diagonal instruction streams of that wave.
        const int NW = block / 64, NCOP = HostPrep::rg3_diag_copies(block);
        const int np_w = d.np;
        std::vector<int> new_idx((size_t)np_w);
        std::vector<std::array<int, 4>> col_of((size_t)np_w);  // rotated column j -> original local index
        std::vector<int> copy_of((size_t)np_w, 0);
        bool wg_fail = false;   // this workgroup's schedule does not fit: nothing of it is written (the mesh falls back to the generic kernels)
        if (nen == 4 && conflict_aware) {
          std::vector<int> lanes_used((size_t)NW, 0);
          std::vector<std::array<std::array<uint8_t, 32>, 3>> bank((size_t)NW);
          for (auto& bw : bank) for (auto& bj : bw) bj.fill(0);
          const int naddr = nvar * nvar * d.nb + 1;
          std::vector<uint8_t> addr_seen((size_t)NW * 3 * naddr, 0);  // [wave][j][address]: already targeted
          std::vector<int> node_in_wave((size_t)NW * (size_t)(n1 - n0), 0);
          static const int PERM[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
          for (int64_t p = inc_ptr[n0]; p < inc_ptr[n1]; p++) {
            const int64_t e = inc_elem[p];
            const int i = inc_loc[p];
            const int64_t I = conn[e * nen + i];
            const int idx = (int)(p - inc_ptr[n0]);
            const int rowoff = (int)((int64_t)nvar * nvar * (P.bptr[I] - d.bb0));
            int base[3], addr[3];
            for (int c = 0; c < 3; c++) {
              base[c] = (c + 1) ^ i;  // the three non-row local indices
              addr[c] = rowoff + nvar * P.eslot[(size_t)e * 16 + i * 4 + base[c]];
            }
            int best_w = -1, best_q = 0;
            long best_cost = -1;
            for (int wv = 0; wv < NW; wv++) {
              if (lanes_used[wv] >= 64 || node_in_wave[(size_t)wv * (n1 - n0) + (I - n0)] >= NCOP) continue;
              for (int q = 0; q < 6; q++) {
                long cost = lanes_used[wv];  // mild preference for the emptiest wave (load balance)
                for (int j = 0; j < 3; j++) {
                  const int ad = addr[PERM[q][j]];
                  cost += 8L * bank[wv][j][ad & 31];
                  if (addr_seen[((size_t)wv * 3 + j) * naddr + ad]) cost += 1000;
                }
                if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_w = wv; best_q = q; }
              }
            }
            if (best_w < 0) { wg_fail = true; break; }   // no wave can take the pair (a node with more pairs than NW * copies lanes)
            new_idx[idx] = lanes_used[best_w] * NW + best_w;
            copy_of[idx] = node_in_wave[(size_t)best_w * (n1 - n0) + (I - n0)]++;
            lanes_used[best_w]++;
            col_of[idx][0] = i;
            for (int j = 0; j < 3; j++) {
              const int ad = addr[PERM[best_q][j]];
              col_of[idx][j + 1] = base[PERM[best_q][j]];
              bank[best_w][j][ad & 31]++;
              addr_seen[((size_t)best_w * 3 + j) * naddr + ad] = 1;
            }
          }
          for (int wv = 0; wv < NW; wv++)
            if (lanes_used[wv] * NW + wv - NW >= block) wg_fail = true;
        } else {
          for (int64_t p = inc_ptr[n0]; p < inc_ptr[n1]; p++) {
            const int idx = (int)(p - inc_ptr[n0]);
            const int i = inc_loc[p];
            new_idx[idx] = idx;
            copy_of[idx] = (idx / NW) % NCOP;
            for (int j = 0; j < nen && j < 4; j++) col_of[idx][j] = (nen == 4) ? (j ^ i) : 0;
          }
        }
        if (wg_fail) {
#pragma omp atomic write
          fail_flag = 1;
          continue;
        }
        // pair records and contribution entries in ascending pair order -> fixed summation order
        for (int64_t p = inc_ptr[n0]; p < inc_ptr[n1]; p++) {
          const int64_t e = inc_elem[p];
          const int i = inc_loc[p];
          const int64_t I = conn[e * nen + i];
          const int64_t idx = new_idx[(size_t)(p - inc_ptr[n0])];
          const std::array<int, 4>& cols = col_of[(size_t)(p - inc_ptr[n0])];
          uint32_t* pr = &P.pair_rec[((size_t)w * block + idx) * nen];
          if (nen == 4) {
            // element of the pair; bits 30-31: which rotated column holds the element's local node 1
            // (models whose aux field is read at that node only, Proteas::AUX_LOCAL_NODE)
            uint32_t pos1 = 0;
            for (int j = 0; j < 4; j++) if (cols[j] == 1) pos1 = (uint32_t)j;
            P.pair_eid[(size_t)w * block + idx] = (uint32_t)e | (pos1 << 30);
            if ((uint64_t)e >= (1ull << 30)) fail_flag = 1;
            uint16_t* ax = &P.pair_aux[((size_t)w * block + idx) * 8];
            const int64_t len = P.bptr[I + 1] - P.bptr[I];
            const int64_t rowoff = (int64_t)nvar * nvar * (P.bptr[I] - d.bb0);
            if (rowoff > 0xFFFF) fail_flag = 1;
            ax[0] = (uint16_t)rowoff;
            ax[1] = (uint16_t)(nvar * len);
            ax[2] = (uint16_t)((I - n0) * nvar);
            ax[3] = (uint16_t)copy_of[(size_t)(p - inc_ptr[n0])];  // private diagonal copy of this lane
            for (int j = 0; j < 4; j++) ax[4 + j] = (uint16_t)(nvar * P.eslot[(size_t)e * 16 + i * 4 + cols[j]]);
          }
          for (int j = 0; j < nen; j++) {
            const int jo = (nen == 4) ? cols[j] : ((j + i) % nen);  // row node first
            pr[j] = conn[e * nen + jo];
            const int lb = (int)(P.bptr[I] + P.eslot[(size_t)e * nen * nen + i * nen + jo] - d.bb0);
            P.contrib[d.c0 + cbeg[lb] + fill[lb]++] = (uint16_t)(idx * stride + (j == 0 ? 0 : j * nvar + 1));
          }
        }
      }
      if (fail_flag) P.rg2_ok = false;
      if (P.rg2_ok && nen == 4) {
        P.wg_ntab.assign((size_t)nwg * 16 * 4, 0);
        for (int64_t w = 0; w < nwg; w++) {
          const int nn = std::min<int>(P.wg2[(size_t)w].nnodes, 16);
          std::copy(P.node_tab.begin() + (size_t)P.wg2[(size_t)w].n0 * 4, P.node_tab.begin() + ((size_t)P.wg2[(size_t)w].n0 + nn) * 4,
                    P.wg_ntab.begin() + (size_t)w * 64);
        }
      }
      // ---- node lists of the workgroups (rg4) --------------------------------------------------
      if (P.rg2_ok && nen == 4) {
        std::vector<int32_t> nuniq((size_t)nwg, 0);
#pragma omp parallel for schedule(dynamic, 256)
        for (int64_t w = 0; w < nwg; w++) {
          std::vector<uint32_t> tmp(P.pair_rec.begin() + (size_t)w * block * 4, P.pair_rec.begin() + (size_t)(w + 1) * block * 4);
          std::sort(tmp.begin(), tmp.end());
          tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
          if (!tmp.empty() && tmp.back() == 0xFFFFFFFFu) tmp.pop_back();
          nuniq[w] = (int32_t)tmp.size();
        }
        int mx = 1;
        for (int64_t w = 0; w < nwg; w++) mx = std::max(mx, nuniq[w]);
        if (mx <= 256) {
          P.rg4_nl_stride = (mx + 63) & ~63;
          P.nlist.assign((size_t)nwg * P.rg4_nl_stride, 0);
          P.pair_loc.assign((size_t)nwg * block, 0xFFFFFFFFu);
#pragma omp parallel for schedule(dynamic, 256)
          for (int64_t w = 0; w < nwg; w++) {
            const uint32_t* pr = &P.pair_rec[(size_t)w * block * 4];
            std::vector<uint32_t> tmp(pr, pr + (size_t)block * 4);
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            if (!tmp.empty() && tmp.back() == 0xFFFFFFFFu) tmp.pop_back();
            uint32_t* nl = &P.nlist[(size_t)w * P.rg4_nl_stride];
            for (int x = 0; x < P.rg4_nl_stride; x++) nl[x] = tmp.empty() ? 0u : tmp[(size_t)std::min<size_t>(x, tmp.size() - 1)];
            for (int idx = 0; idx < block; idx++) {
              if (pr[idx * 4] == 0xFFFFFFFFu) continue;
              uint32_t packed = 0;
              for (int j = 0; j < 4; j++) {
                const uint32_t li = (uint32_t)(std::lower_bound(tmp.begin(), tmp.end(), pr[idx * 4 + j]) - tmp.begin());
                packed |= li << (8 * j);
              }
              P.pair_loc[(size_t)w * block + idx] = packed;
            }
          }
        }
      }
    }
    if (!P.rg2_ok) { P.wg2.clear(); P.pair_rec.clear(); P.chunk.clear(); P.sdesc.clear(); P.contrib.clear(); P.pair_aux.clear(); P.node_tab.clear(); P.nlist.clear(); P.pair_loc.clear(); P.pair_eid.clear(); P.wg_ntab.clear(); P.rg4_nl_stride = 0; }
  }
  return std::string();
}

std::string solid_gather_build(const HostPrep& P, SolidGather& G) {
  const int nen = P.nen;
  const int64_t nb = P.bptr.empty() ? 0 : P.bptr[(size_t)P.n_owned];
  const int64_t npairs = P.node_pair_ptr.empty() ? 0 : P.node_pair_ptr[(size_t)P.n_owned];
  if ((uint64_t)P.n_elem * nen * nen >= 0xFFFFFFFFull || (uint64_t)npairs * nen >= 0xFFFFFFFFull)
    return "mesh too large for 32-bit gather lists";
  G.gptr.assign((size_t)nb + 1, 0);
  G.brow.resize((size_t)nb);
  G.gsrc.resize((size_t)npairs * nen);
#pragma omp parallel for schedule(static)
  for (int64_t I = 0; I < P.n_owned; I++) {
    for (int64_t b = P.bptr[I]; b < P.bptr[I + 1]; b++) G.brow[(size_t)b] = (int32_t)I;
    for (int64_t p = P.node_pair_ptr[I]; p < P.node_pair_ptr[I + 1]; p++) {
      const uint16_t* es = &P.eslot[((size_t)P.pair_elem[p] * nen + P.pair_local[p]) * nen];
      for (int j = 0; j < nen; j++) G.gptr[(size_t)(P.bptr[I] + es[j]) + 1]++;
    }
  }
  for (int64_t b = 0; b < nb; b++) G.gptr[(size_t)b + 1] += G.gptr[(size_t)b];
#pragma omp parallel for schedule(static)
  for (int64_t I = 0; I < P.n_owned; I++) {
    const int64_t b0 = P.bptr[I], len = P.bptr[I + 1] - b0;
    std::vector<uint32_t> fill((size_t)len, 0);
    for (int64_t p = P.node_pair_ptr[I]; p < P.node_pair_ptr[I + 1]; p++) {
      const uint32_t base = (P.pair_elem[p] * (uint32_t)nen + P.pair_local[p]) * (uint32_t)nen;
      const uint16_t* es = &P.eslot[(size_t)base];
      for (int j = 0; j < nen; j++) G.gsrc[(size_t)G.gptr[(size_t)(b0 + es[j])] + fill[es[j]]++] = base + (uint32_t)j;
    }
  }
  return std::string();
}

}  // namespace rdc
