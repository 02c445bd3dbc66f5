// rdc_prep.h — host-side mesh preparation (pure C++, no HIP): pattern, slot map, colouring,
// first-writer masks, row-gather work lists.  Kept free of HIP headers so the CPU-only tests can
// compile rdc_meshprep.cpp directly (tests/host_shim.cpp).
#ifndef RDC_PREP_H
#define RDC_PREP_H
#include <stdint.h>
#include <string>
#include <vector>

namespace rdc {

// ---- host mesh preparation (what es.init() + DofMap give the reference callback) ----------
struct HostPrep {
  int nen = 0, nvar = 0;
  int64_t n_elem = 0, n_node = 0, n_owned = 0;
  // node-block CSR pattern of the owned rows
  std::vector<int64_t> bptr;
  std::vector<int32_t> bcol;
  std::vector<uint16_t> eslot;  // [n_elem][nen*nen]; 0xFFFF where the row node is a ghost
  // colouring
  int n_colours = 0;
  std::vector<int32_t> colour;        // per element
  std::vector<uint32_t> elem_order;   // colour-sorted
  std::vector<int64_t> colour_ptr;    // [n_colours+1]
  std::vector<uint64_t> first_mask;
  std::vector<uint8_t> first_rhs;
  // row gather
  bool rowgather_ok = false;
  int rg_block = 256;
  size_t rg_lds_bytes = 0;            // max over workgroups
  std::vector<uint32_t> pair_elem;
  std::vector<uint8_t> pair_local;
  std::vector<int64_t> node_pair_ptr;
  std::vector<int32_t> wg_node_ptr;
  // node-staged generic row gather (HEX8): the distinct nodes of a workgroup's pairs, so their coordinates and
  // unknowns can sit in LDS instead of 96 registers per lane; pairs address them by 16-bit list positions
  bool hx_ok = false;
  std::vector<int64_t> hx_nl_ptr;     // [n_wg + 1] into hx_nlist
  std::vector<uint32_t> hx_nlist;     // node ids
  std::vector<uint16_t> hx_ploc;      // [n_pairs][nen] list position of the pair's element nodes (local order)
  int hx_max_nodes = 0;               // largest list
  // staged row gather ("rg2"): flat per-workgroup descriptors so the kernel needs only two
  // dependent load levels, and balanced contribution chunks for the deterministic LDS gather
  struct WgDesc {          // 64 bytes
    int64_t vb0;           // first CSR value of the workgroup's rows  (nvar^2 * bptr[n0])
    int64_t bb0;           // first node block                        (bptr[n0])
    int64_t c0;            // first entry in `contrib`
    int64_t ch0;           // first entry in `chunk`
    int32_t n0, nnodes;    // owned nodes [n0, n0 + nnodes)
    int32_t nb, np;        // node blocks and (node, element) pairs of the workgroup
    int32_t nch, nout;     // gather chunks; partial-sum slots (blocks + extra chunks of split lists)
    int64_t pad;
  };
  struct Chunk {           // 8 bytes: one gather work item = <= RG2_CHUNK contributions to one block
    uint16_t cbeg;         // first contribution (relative to WgDesc::c0)
    uint16_t cnt;
    uint16_t dst;          // partial-sum slot (in units of nvar doubles)
    uint16_t pad;
  };
  struct StoreDesc {       // 8 bytes, one per node block
    uint16_t outoff;       // value offset of (a = 0, b = 0) relative to vb0
    uint8_t len;           // blocks in the row of the owner node
    uint8_t nextra;        // extra partial-sum slots to add (lists longer than one chunk)
    uint16_t extra;        // first extra slot
    uint8_t diag;          // 1 for the diagonal block of its node: its slot also carries the rhs sum
    uint8_t node;          // owner node index within the workgroup
  };
  static constexpr int RG2_CHUNK = 6;
  bool rg2_ok = false;
  int rg2_block = 256;
  std::vector<WgDesc> wg2;
  std::vector<uint32_t> pair_rec;   // [n_wg][block][nen] node ids, row node first (TET4: j -> j ^ i); ~0u = no pair
  // per pair, for the LDS-accumulating kernel: offsets inside the workgroup's row slice (doubles):
  // {rowoff, stride = nvar*len, rhsoff, pad, off[0..3] = nvar*slot of rotated column j}  (TET4 only)
  std::vector<uint16_t> pair_aux;   // [n_wg][block][8]
  size_t rg2_lds_bytes = 0;         // largest row slice (values + rhs + private diagonal copies) of a workgroup
  // private accumulators of a node's diagonal block + rhs: one copy per lane of the node inside a wave
  // (pairs of a node are dealt round-robin to the block/64 waves) => copies = 24 * 64 / block
  static constexpr int rg3_diag_copies(int block) { return 1536 / block; }
  static constexpr int RG3_DIAG_SLOTS = 64;  // fixed slot count (node * copies + copy) => 64 / copies nodes per workgroup
  std::vector<uint16_t> node_tab;   // [n_owned][4] {rowoff, stride, nvar*diag slot, 0}: where the diagonal block sits
  // persistent pipelined kernel ("rg4"): the distinct nodes a workgroup touches, so their records can be
  // prefetched into LDS by LDS-DMA; pairs then address nodes by an 8-bit index into that list
  int rg4_nl_stride = 0;            // list length per workgroup (multiple of 64, max over workgroups)
  std::vector<uint32_t> nlist;      // [n_wg][rg4_nl_stride] node ids, padded with the first id
  std::vector<uint32_t> pair_loc;   // [n_wg][block] four 8-bit list indices (row node in the low byte); ~0u = no pair
  std::vector<uint16_t> wg_ntab;    // [n_wg][16][4]: node_tab rows of the workgroup's nodes, for LDS-DMA by work-item index
  std::vector<uint32_t> pair_eid;   // [n_wg][block] element of the pair (uploaded only for models with per-element inputs)
  std::vector<Chunk> chunk;
  std::vector<StoreDesc> sdesc;     // [total node blocks]
  std::vector<uint16_t> contrib;    // stage index (doubles) of the contribution: pair * stride + slot(column), see rdc_meshprep.cpp
};

// Gather lists of the two-pass solid assembly (rdc_solid.hip): for every node block b of the owned rows the
// element-matrix blocks (e, i, j) that sum into it, as src = (e * nen + i) * nen + j, in ascending
// (element, local row) order => a fixed summation order.  gptr has bptr[n_owned] + 1 entries.
struct SolidGather {
  std::vector<uint32_t> gptr, gsrc;
  std::vector<int32_t> brow;   // owner node of block b
};
std::string solid_gather_build(const HostPrep& P, SolidGather& out);

// ---- work lists of the element-visit / moment kernel (rdc_tet4_ev.hip; TET4, 5 unknowns) --------------------------
// A workgroup owns a CLUSTER of <= 16 owned nodes and visits every element that touches it (<= 256) once.
struct HostPrepEv {
  static constexpr int MAXN = 16, NBP = 256, BLOCK = 256;
  struct Desc {            // 32 bytes per workgroup
    uint32_t nown, nvis, ntouch, nb;   // owned nodes, element visits, distinct nodes of the visits, node blocks of the owned rows
    uint32_t out_doubles;              // size of the CSR image of the cluster's rows in LDS (doubles, incl. phase padding)
    uint32_t min_node, max_node, pad;  // smallest / largest owned node id (two-part assembly)
  };
  struct Node {            // 16 bytes per owned node of a workgroup
    uint32_t bptr;         // first node block of the node's rows (CSR value offset = nvar^2 * bptr)
    uint16_t len, blk0;    // node blocks in the row (<= 16); index a of the node in the cluster: block (a, slot s) of the moment slice = s * 16 + a
    uint32_t obase;        // offset of the node's CSR segment inside the LDS image (same 16-byte phase as in memory)
    uint32_t node;         // node id
  };
  bool ok = false;
  int nls = 0;                       // node list stride (multiple of 64)
  size_t max_out_doubles = 0;
  std::vector<Desc> desc;
  std::vector<uint32_t> nlist;       // [n_wg][nls] node ids, the owned ones first, padded with the first
  std::vector<uint32_t> vloc;        // [n_wg][256] four 8-bit list positions of the visit's vertices, owned first; ~0u = none
  std::vector<uint32_t> vslot;       // [n_wg][256][2]: 4-bit column slot of vertex j in the row of vertex i at bits 16 (i & 1) + 4 j of word i / 2
  std::vector<Node> ntab;            // [n_wg][16]
  std::vector<uint8_t> bpart;        // [n_wg][256]: mirror block (column node -> row node) of block t = slot * 16 + node when the column node is another node of the cluster, else t itself
  // statistics: element visits and (row, visit) pairs over all workgroups
  int64_t n_visits = 0, n_rows = 0;
  int64_t n_conflicts = 0;           // rows whose node already sits at the same vertex position of their 16-lane group
  int64_t n_group_rows = 0;          // sum over 16-lane groups of the rows they emit (LDS passes per accumulated value)
  int64_t n_pass_instr = 0;          // sum over waves of the LDS atomic instructions of the row positions they issue (PASS_COST)
  // LDS atomics of row position i of the PIHNA visit (rdc_tet4_ev.h: 15 symmetric moments to the columns j >= i, 4 of the
  // non-symmetric one, 5 rhs entries); the coefficient-form kernel (three unknowns) pays the same for every position
  static constexpr int PASS_COST_PIHNA[4] = {69, 54, 39, 24};
  static constexpr int PASS_COST_FLAT[4] = {40, 40, 40, 40};
  static constexpr int COLLISION_COST = 17;   // atomic wave-instructions a colliding row is worth (one extra LDS pass for each of its atomics)
};
// needs P.bptr / P.bcol / P.eslot of prep_build; lds_budget = LDS bytes a workgroup may use (3 workgroups per CU: 53 KB)
// n_interior >= 0: owned nodes [0, n_interior) are "interior" (two-part assembly): clusters do not mix the two kinds
std::string prep_build_ev(const HostPrep& P, const uint32_t* conn, size_t lds_budget_bytes, HostPrepEv& out, int64_t n_interior = -1);

// ---- node clusters with producer / consumer work lists (HEX8: rdc_solid_cl.hip, rdc_hex8_cl.hip) ------------------------
// A workgroup owns a CLUSTER of owned nodes.  Its consumer lanes each take one (owned node, incident element) PAIR and
// accumulate that row of the element matrix over the quadrature points; its producer lanes each take one ELEMENT
// touching the cluster and evaluate the per-point data once per point for all the pairs of the element.
struct HostPrepCl {
  // max_row_doubles: LDS image of the cluster's CSR rows; img_per_block: doubles of a node block in that image (0 = nvar^2, the
  // whole rows; nvar = one equation row at a time, as the five-unknown kernel keeps it)
  struct Limits { int max_nodes = 24, max_pairs = 192, max_elems = 64, max_row_doubles = 6000, pair_order = 1, img_per_block = 0; };
  struct Desc {            // 16 bytes per workgroup
    uint16_t nown, npair, nelem, pad;
    uint32_t row_doubles;  // LDS image of the CSR rows of all owned nodes: sum of nvar^2 * len
    uint32_t min_node_max_node_pad;
  };
  struct Node {            // 16 bytes per owned node of a workgroup
    uint32_t bptr;         // first node block of the node's rows
    uint16_t len;          // node blocks in the row
    uint16_t off;          // offset (doubles) of the node's rows in the image (same layout as in the CSR array)
    uint32_t node;         // node id
    uint32_t pad;
  };
  bool ok = false;
  Limits lim;
  int nvar = 0;
  std::vector<Desc> desc;
  std::vector<Node> ntab;            // [n_wg][max_nodes]
  std::vector<uint32_t> eid;         // [n_wg][max_elems] element ids, ~0u = none
  std::vector<uint32_t> pair;        // [n_wg][max_pairs] local element | local row node << 8 | owned-node index << 16; ~0u = none (order: rdc_prep_cl.cpp)
  std::vector<uint32_t> pslot;       // [n_wg][max_pairs][nen / 4]: byte j = column slot of local node j in the pair's row
  size_t max_row_doubles = 0;
  int64_t n_wg_interior = 0;         // leading clusters whose nodes are all interior (two-part assembly)
  int64_t part1_nodes = 0;           // the rows of nodes [0, part1_nodes) are complete after those clusters
  // statistics
  int64_t n_elem_visits = 0, n_pairs = 0;
};
// n_interior >= 0: owned nodes [0, n_interior) are "interior" (two-part assembly): clusters do not mix the two kinds and the
// interior clusters come first in the lists ([0, n_wg_interior))
std::string prep_build_cl(const HostPrep& P, const uint32_t* conn, const HostPrepCl::Limits& lim, HostPrepCl& out, int64_t n_interior = -1);

// returns empty string on success, else an error message
std::string prep_build(int nen, int64_t n_elem, int64_t n_node, int64_t n_owned, const uint32_t* conn,
                       int nvar, size_t lds_budget_bytes, int block, HostPrep& out, bool conflict_aware = true);

}  // namespace rdc
#endif
