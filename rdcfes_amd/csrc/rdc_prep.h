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
};

// returns empty string on success, else an error message
std::string prep_build(int nen, int64_t n_elem, int64_t n_node, int64_t n_owned, const uint32_t* conn,
                       int nvar, size_t lds_budget_bytes, int block, HostPrep& out);

}  // namespace rdc
#endif
