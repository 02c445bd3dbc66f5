// rdc_internal.h — context object and host-side mesh preparation shared by the C-ABI
// translation units.  Not installed; the public surface is include/rdc_assembly.h.
#ifndef RDC_INTERNAL_H
#define RDC_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <type_traits>
#include <vector>

#include "rdc_kernels.h"
#include "rdc_prep.h"

namespace rdc {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: set it once per (kernel instantiation, device
// ordinal) -- a process may hold contexts on several ordinals (rdc_device_count / rdc_ctx_create).
inline void dyn_lds_once(std::atomic<uint64_t>& done, const void* fn, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = dev >= 0 && dev < 64 ? (1ull << dev) : 0;
  if (bit && (done.load(std::memory_order_relaxed) & bit)) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (bit) done.fetch_or(bit, std::memory_order_relaxed);
}

// device view of the staged row-gather work lists (HostPrep::wg2 ...)
struct Rg2Dev {
  int n_wg = 0;
  const HostPrep::WgDesc* desc = nullptr;
  const uint32_t* pair_rec = nullptr;
  const HostPrep::Chunk* chunk = nullptr;
  const HostPrep::StoreDesc* sdesc = nullptr;
  const uint16_t* contrib = nullptr;
  const uint16_t* pair_aux = nullptr;
  const uint16_t* node_tab = nullptr;
  const uint32_t* nlist = nullptr;
  const uint32_t* pair_loc = nullptr;
  int wg_begin = 0, wg_count = -1;      // sub-range of the work items to launch (k_tet4_rg5 only; -1 = all)
  const uint16_t* wg_ntab = nullptr;    // [n_wg][16][4]: node_tab entries of the workgroup's nodes (persistent kernel)
  const uint32_t* pair_eid = nullptr;   // element of the pair (models with per-element inputs only)
  int nl_stride = 0;
  size_t lds_bytes = 0;
  int block = 256;
};

// device view of the element-visit work lists (HostPrepEv)
struct EvDev {
  int n_wg = 0;
  const HostPrepEv::Desc* desc = nullptr;
  const uint32_t* nlist = nullptr;
  const uint32_t* vloc = nullptr;
  const uint32_t* vslot = nullptr;
  const HostPrepEv::Node* ntab = nullptr;
  const uint8_t* bpart = nullptr;
  const uint32_t* wg_perm = nullptr;   // two-part assembly: workgroup order with the interior clusters first (else null)
  int wg_begin = 0, wg_count = -1;
  int nls = 0;
  size_t max_out_doubles = 0;
};

// device view of the cluster lists (HostPrepCl) of the producer / consumer HEX8 kernels (rdc_solid_cl.hip, rdc_hex8_cl_kernel.h)
struct ClDev {
  int n_wg = 0, cw = 3, pw = 1;     // consumer / producer waves per workgroup the lists were built for
  const HostPrepCl::Desc* desc = nullptr;
  const HostPrepCl::Node* ntab = nullptr;
  const uint32_t* eid = nullptr;
  const uint32_t* pair = nullptr;
  const uint32_t* pslot = nullptr;
  size_t max_row_doubles = 0;
  int grid = 0;                     // persistent workgroups to launch (0 = one workgroup per cluster)
};

// ---- kernel launch plumbing -----------------------------------------------------------------
struct LaunchArgs {
  MeshDev m;
  int nen, exp_mode, strategy;
  const double* u;
  const double* aux;
  // node-staged generic row gather (HEX8)
  const int64_t* hx_nl_ptr = nullptr;
  const uint32_t* hx_nlist = nullptr;
  const uint16_t* hx_ploc = nullptr;
  int hx_max_nodes = 0;
  int opt_staged = 1;
  const double* elem = nullptr;  // per-element inputs ([n_elem][M::NELEM]) of models that have them (ADPM tracts)
  double* packed;  // scratch for the per-node records of the TET4 fast path
  int pack_part = 0;               // 0 = pack every record; 1 = owned nodes only, then record pack_event; 2 = wait for pack_event, ghosts only
  hipEvent_t pack_event = nullptr;
  int variant;     // RDC_VARIANT_*
  int opt_occ, opt_ablate, opt_kernel, opt_special, opt_xcd, opt_grid, opt_pf, opt_slim = 0, opt_moments = 1, opt_stagger = 0, opt_ldspad = 0;  // tuning knobs (rdc_set_option)
  Rg2Dev rg2;
  EvDev ev;
  ClDev cl;              // HEX8, three unknowns: cluster lists (n_wg = 0: not available / not wanted)
  bool use_ev = false;   // element-visit kernel allowed for this call
  int opt_ev_occ = 3;
  int opt_evc_occ = 2;
  int ev_grid = 0;           // 2 x CUs: the resident element-visit kernel launches 3/2 of it (22 moments: all of it)
  int opt_ev_bg = 1;         // k_tet4_ev: waves all of whose visits are in the background state skip the moments that are sums of zeros (rdc_tet4_ev.h, bg)
  int* ev_ticket = nullptr;  // k_tet4_evq: cluster counter (one int, zeroed by the launch)
  bool ev_general = false;   // k_tet4_ev with every PIHNA term on (22 moments) instead of the shipped parameter pattern (16)
  int opt_ev_resident = 1;   // k_tet4_evq: three resident workgroups per CU walking over the clusters, next cluster fetched ahead (whole-mesh launches)
  long long* stamps = nullptr;  // diagnostic phase stamps (rdc_debug_stamps)
  double* val;
  double* rhs;
  hipStream_t stream;
  hipEvent_t ev_start;  // recorded right before the dominant kernel(s) when timing is on, else null
  const int64_t* colour_ptr;  // host
  int n_colours;
  int n_wg;
  size_t lds_bytes;
};

template <class M>
hipError_t launch_rd(const LaunchArgs& a, const typename M::K& k);
// TET4-specialised factored kernels (rdc_tet4_fast.hip)
template <class M>
hipError_t launch_tet4_fast(const LaunchArgs& a, const typename M::K& k);
// node records of the PIHNA kernels (xyz | u, 64 bytes per node), honouring the two-part pack rules (rdc_tet4_fast.hip)
hipError_t pack_nodes_pihna(const LaunchArgs& a);
// element-visit / moment kernel of the shipped-pattern PIHNA TET4 assembly (rdc_tet4_ev.hip)
hipError_t launch_tet4_ev(const LaunchArgs& a, const PihnaK& k);
// element-visit kernel in coefficient form for the three-unknown TET4 models (rdc_tet4_evc.hip: Ripf, RipfReduced, Hcc, HccMassOnly);
// the node records must have been packed
template <class M>
hipError_t launch_tet4_evc(const LaunchArgs& a, const typename M::K& k);
// which models run it: measured on K(94) against k_tet4_rg5 (tools/ab.py): Ripf with all terms on 1.41 vs 1.79 ms (the per-element
// part -- two exp, sqrt, the unit gradient -- dominates); RipfReduced 0.87 vs 0.78, Hcc 0.90 vs 0.65: the pair kernel packs the
// row atomics of a wave densely, an element visit executes a row position for as few as 10 of 64 lanes
// (Pihna with all terms on, five unknowns, was tried in round 3: tet4_visit<Pihna> needs 3.9 KB of scratch per lane -- profiles/r03_ev_ab_log.md)
template <class M> struct EvcEligible { static constexpr bool value = std::is_same<M, Ripf>::value; };

}  // namespace rdc
#endif
