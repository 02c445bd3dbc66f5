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
  // staged row gather ("rg2"): flat per-workgroup descriptors so the kernel needs only two
  // dependent load levels, and contribution lists for the deterministic LDS gather
  struct WgDesc {          // 48 bytes
    int64_t vb0;           // first CSR value of the workgroup's rows  (nvar^2 * bptr[n0])
    int64_t bb0;           // first node block                        (bptr[n0])
    int64_t c0;            // first entry in `contrib`
    int32_t n0, nnodes;    // owned nodes [n0, n0 + nnodes)
    int32_t nb, np;        // node blocks and (node, element) pairs of the workgroup
    int64_t pad;
  };
  struct BlkInfo {         // 8 bytes, one per node block
    uint16_t cbeg;         // first contribution (relative to WgDesc::c0)
    uint8_t cnt;           // number of contributions
    uint8_t len;           // blocks in the row of the owner node
    uint16_t outoff;       // value offset of (a = 0, b = 0) relative to vb0
    uint16_t pad;
  };
  bool rg2_ok = false;
  int rg2_block = 256;
  std::vector<WgDesc> wg2;
  std::vector<uint32_t> pair_rec;   // [n_wg][block][nen] node ids, row node first (TET4: j -> j ^ i); ~0u = no pair
  std::vector<BlkInfo> blk_info;    // [total node blocks]
  std::vector<uint16_t> contrib;    // pair_index * nen + rotated column
  std::vector<uint32_t> node_info;  // [n_owned] (first pair within the workgroup) << 16 | pair count
};

// returns empty string on success, else an error message
std::string prep_build(int nen, int64_t n_elem, int64_t n_node, int64_t n_owned, const uint32_t* conn,
                       int nvar, size_t lds_budget_bytes, int block, HostPrep& out);

}  // namespace rdc
#endif
