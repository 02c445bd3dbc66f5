// rdc_capi.hip — implementation of the C-ABI declared in include/rdc_assembly.h.
// No CPU fallback lives here: every assemble call launches HIP kernels or fails.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <type_traits>

#include "rdc_internal.h"
#include "rdc_tet4_pihna_moments.h"
#include "rdc_tet4_ev.h"
#include "rdc_solid.h"

using namespace rdc;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool owned = true;
};

thread_local char g_create_error[512] = "";

}  // namespace

struct rdc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  char err[512] = "";
  bool have_mesh = false;
  int strategy = RDC_SCATTER_AUTO;
  int variant = RDC_VARIANT_AUTO;
  int opt_occ = 2, opt_ablate = 0, opt_kernel = 0;
  int opt_xcd = 0;      // XCD-aware workgroup order of the row-gather kernel (measured: no gain, off)
  int opt_pf = 0;       // L2 prefetch distance (workgroups) of the work lists in k_tet4_rg5; 0 = off
  int opt_grid = 0;     // persistent grid size of the pipelined kernel (0 = 2 workgroups per CU)
  int opt_sched = 1;    // LDS-conflict-aware pair schedule (takes effect at the next rdc_mesh_upload)
  int opt_special = 1;  // allow parameter-sparsity kernel variants
  int opt_part = 0;            // 0 = whole mesh, 1 = workgroups of interior nodes only, 2 = the remaining workgroups
  int64_t opt_interior = -1;   // owned nodes [0, opt_interior) have no ghost node in any of their elements
  hipEvent_t pack_event = nullptr;  // two-part assembly: recorded behind part 1's pack of the owned node records
  bool part1_packed = false;        // part 1 of the current step has packed the owned records (consumed by part 2)
  int opt_stagger = 0;
  int opt_ldspad = 0;
  int opt_moments = 1;  // PIHNA (cell transport off) TET4: moment form of the rows
  int opt_slim = 0;     // PIHNA: slim per-point state (re-derived per equation row); with occupancy=3 three waves per SIMD
  int opt_block = 256;  // workgroup size of the row-gather work lists (takes effect at the next rdc_mesh_upload)
  HostPrep prep;
  HostPrepEv prep_ev;          // element-visit lists (PIHNA TET4, shipped pattern); .ok = available
  DevBuf ev_desc, ev_nlist, ev_vloc, ev_vslot, ev_ntab, ev_bpart, ev_perm, ev_ticket;
  bool ev_tried = false;           // the element-visit lists of this mesh have been built (or found impossible)
  int64_t ev_perm_interior = -2;   // "interior_nodes" value the uploaded workgroup order was built for
  int ev_part1_wg = 0;             // leading workgroups of that order whose clusters are interior
  int scl_interior = -1, scl_n_wg_interior = 0;   // "interior_nodes" the cluster lists were built with; leading interior clusters
  int64_t scl_part1_nodes = 0;
  // chunked hand-back (rdc_csr_download_rows_async): a copy stream of the context's own and a small pool of completion events
  hipStream_t copy_stream = nullptr;
  hipEvent_t copy_fence = nullptr;            // recorded on the context's stream at call time: the copy starts behind the work enqueued so far
  static constexpr int N_TICKETS = 16;
  hipEvent_t ticket[N_TICKETS] = {};
  int next_ticket = 0;
  hipEvent_t solid_part1_event = nullptr;   // recorded behind part 1 of a two-part solid assembly (the sides of part 2 wait for it)
  bool solid_part1_pending = false;
  int64_t part1_nodes = -1;        // rows [0, part1_nodes) were complete after the LAST part-1 call (-1: none since the upload)
  int opt_ev_bg = 1;               // 1 = the element-visit kernel skips the zero moments of waves in the background state (n = c = h = a = 0), 0 = evaluates everything
  int opt_ev_general = 1;          // 1 = PIHNA / TET4 with any parameter values through the element-visit kernel with 22 moments; 0 = pair kernel (k_tet4_rg5<Pihna>)
  int opt_ev_resident = 1;         // 1 (default) = whole-mesh launches of the shipped-pattern element-visit kernel run as k_tet4_evq (resident workgroups that fetch the next cluster while this one is expanded); 0 = k_tet4_ev
  int opt_ev_occ = 3;              // launch-bound waves per SIMD of the element-visit kernel (2 or 3)
  int opt_evc_occ = 2;             // ... of the coefficient-form element-visit kernel (k_tet4_evc): 2 (default) or 3 (spills: measured 2.24 vs 1.42 ms)
  int opt_ev_lds = 54000;          // LDS bytes per workgroup the clusters are sized for (3 workgroups per CU)
  // device mesh data
  DevBuf conn, xyz, bptr, eslot, elem_order, first_mask, first_rhs, pair_elem, pair_local, node_pair_ptr,
      wg_node_ptr;
  DevBuf val, rhs, packed;
  DevBuf stamps;
  DevBuf rg2_desc, rg2_pair, rg2_chunk, rg2_sdesc, rg2_contrib, rg2_aux, rg2_ntab, rg4_nlist, rg4_ploc;
  DevBuf hx_nl_ptr, hx_nlist, hx_ploc;   // node-staged generic row gather (HEX8)
  int opt_staged = 1;
  DevBuf rg4_wgntab;
  DevBuf rg5_eid;             // pair -> element list, uploaded at the first assembly of a model with per-element inputs
  bool rg5_eid_ready = false;
  DevBuf field[RDC_FIELD_COUNT];
  int64_t field_count[RDC_FIELD_COUNT] = {0, 0, 0, 0, 0, 0, 0};
  DevBuf wg_max;  // per-workgroup maxima of rdc_ripf_check_solution
  // solid
  DevBuf elem_material, materials, side_elem, side_id, side_disp;
  DevBuf adpm_slot;
  DevBuf solid_ke, solid_fe, sg_gptr, sg_gsrc, sg_brow, solid_post;  // two-pass assembly: element matrices + gather lists
  bool solid_gather_ready = false;
  DevBuf scl_desc, scl_ntab, scl_eid, scl_pair, scl_pslot;           // fused cluster kernel (HEX8 tangent)
  int solid_cl_state = 0;    // 0 = lists not built yet, 1 = ready, -1 = not available for this mesh (two-pass is used)
  int solid_cl_waves = 31;   // consumer / producer waves the lists were built for (opt_solid_cl_waves at that time)
  size_t scl_max_row_doubles = 0;
  int scl_n_wg = 0;
  int opt_hex_kernel = 0;       // HEX8 reaction-diffusion, three unknowns: 0 = producer / consumer cluster kernel (default), 1 = pair kernels (k_rowgather_staged / k_rowgather), 2 = persistent form of the cluster kernel
  int opt_solid_cl_order = -1;  // pair order of the cluster lists (rdc_prep_cl.cpp): 1 = element-major over colour-sorted elements, 0 = node-distinct, -1 = by first use (solid: 1, reaction-diffusion: 0)
  int solid_cl_order = 1;
  int opt_solid_cl_waves = 31;  // 31 = 3 consumer + 1 producer waves (two workgroups per CU), 62 = 6 + 2 (one per CU)
  int opt_solid_kernel = 0;  // 0 = default: fused cluster kernel for HEX8 tangent requests, two-pass otherwise; 1 = coloured read-modify-write; 2 = two-pass; 3 = fused (error if unavailable)
  int opt_solid_split = 1;   // two-pass, pass 1: 1 = one thread per element row (default; measured faster), 0 = HEX8 row columns split between two threads
  int opt_solid_store = 0;   // two-pass, pass 1 diagnostics (see SolidArgs::store_mode)
  int opt_solid_gather = 0;  // two-pass, pass 2: 0 = stores staged through LDS, 1 = direct 24-byte pieces
  int32_t n_materials = 0;
  int64_t n_sides = 0;
  // timing
  bool timing = false;
  std::vector<hipEvent_t> ev;   // pairs (start, stop), one pair per timed assemble call
  size_t ev_used = 0;           // events handed out since the last rdc_timing_sum_ms / enable
  size_t max_lds = 64 * 1024;
  int n_cu = 256;
};

namespace {

int fail(rdc_ctx* c, int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  if (c) vsnprintf(c->err, sizeof(c->err), fmt, ap);
  else vsnprintf(g_create_error, sizeof(g_create_error), fmt, ap);
  va_end(ap);
  return code;
}

#define RDC_HIP(ctx, call)                                                                       \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) return fail(ctx, RDC_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

int dev_free(rdc_ctx* c, DevBuf& b) {
  if (b.p && b.owned) {
    hipError_t e = hipFree(b.p);
    if (e != hipSuccess) return fail(c, RDC_ERR_HIP, "hipFree failed: %s", hipGetErrorString(e));
  }
  b = DevBuf();
  return RDC_OK;
}

int dev_alloc(rdc_ctx* c, DevBuf& b, size_t bytes) {
  if (b.p && b.owned && b.bytes >= bytes && bytes > 0) return RDC_OK;
  int rc = dev_free(c, b);
  if (rc) return rc;
  if (bytes == 0) bytes = 8;
  hipError_t e = hipMalloc(&b.p, bytes);
  if (e != hipSuccess) {
    b = DevBuf();
    return fail(c, RDC_ERR_ALLOC, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  b.bytes = bytes;
  b.owned = true;
  return RDC_OK;
}

template <class T>
int dev_upload(rdc_ctx* c, DevBuf& b, const std::vector<T>& v) {
  int rc = dev_alloc(c, b, v.size() * sizeof(T));
  if (rc) return rc;
  if (!v.empty()) RDC_HIP(c, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
  return RDC_OK;
}

int set_device(rdc_ctx* c) {
  RDC_HIP(c, hipSetDevice(c->device));
  return RDC_OK;
}

int64_t field_width(const rdc_ctx* c, int field) {
  switch (field) {
    case RDC_FIELD_OLD_SOLUTION: return c->prep.nvar;
    case RDC_FIELD_AUX_NODAL: return 3;
    case RDC_FIELD_UNDEFORMED_XYZ: return 3;
    case RDC_FIELD_ELEM_FIBRE: return 3;
    case RDC_FIELD_PREV_SOLUTION: return c->prep.nvar;
    case RDC_FIELD_TIME_DERIV: return c->prep.nvar;
    case RDC_FIELD_RT_DOSE: return 3;
  }
  return 0;
}

int64_t field_expected(const rdc_ctx* c, int field) {
  const int64_t rows = (field == RDC_FIELD_ELEM_FIBRE) ? c->prep.n_elem : c->prep.n_node;
  return rows * field_width(c, field);
}

MeshDev mesh_view(const rdc_ctx* c) {
  MeshDev m;
  m.n_elem = c->prep.n_elem; m.n_node = c->prep.n_node; m.n_owned = c->prep.n_owned;
  m.conn = (const uint32_t*)c->conn.p;
  m.xyz = (const double*)c->xyz.p;
  m.bptr = (const int64_t*)c->bptr.p;
  m.eslot = (const uint16_t*)c->eslot.p;
  m.elem_order = (const uint32_t*)c->elem_order.p;
  m.first_mask = (const uint64_t*)c->first_mask.p;
  m.first_rhs = (const uint8_t*)c->first_rhs.p;
  m.pair_elem = (const uint32_t*)c->pair_elem.p;
  m.pair_local = (const uint8_t*)c->pair_local.p;
  m.node_pair_ptr = (const int64_t*)c->node_pair_ptr.p;
  m.wg_node_ptr = (const int32_t*)c->wg_node_ptr.p;
  return m;
}

int resolve_strategy(rdc_ctx* c, int* out) {
  int s = c->strategy;
  const bool rg = c->prep.rowgather_ok || (c->prep.rg2_ok && c->prep.nen == 4 && c->variant != RDC_VARIANT_GENERIC);
  if (s == RDC_SCATTER_AUTO) s = rg ? RDC_SCATTER_ROWGATHER : RDC_SCATTER_COLOURED;
  if (s == RDC_SCATTER_ROWGATHER && !rg)
    return fail(c, RDC_ERR_UNSUPPORTED, "row-gather scatter unavailable: a node row exceeds the LDS budget");
  *out = s;
  return RDC_OK;
}

// hands out the next (start, stop) event pair of the timing pool, growing it on demand
int next_event_pair(rdc_ctx* c, hipEvent_t* start, hipEvent_t* stop) {
  if (c->ev_used + 2 > c->ev.size()) {
    for (int x = 0; x < 2; x++) {
      hipEvent_t e = nullptr;
      RDC_HIP(c, hipEventCreate(&e));
      c->ev.push_back(e);
    }
  }
  *start = c->ev[c->ev_used];
  *stop = c->ev[c->ev_used + 1];
  c->ev_used += 2;
  return RDC_OK;
}

// parameter-sparsity specialisation: models may offer a variant with smaller structural masks that is
// exact for the given parameter values (PIHNA with the cell transport terms off)
template <class M, class P>
hipError_t launch_specialised(const LaunchArgs& a, const typename M::K& k, const P&) { return launch_rd<M>(a, k); }
template <>
hipError_t launch_specialised<Pihna, rdc_pihna_params>(const LaunchArgs& a, const Pihna::K& k, const rdc_pihna_params& p) {
  // any parameter values: the element-visit kernel with all 22 moments ("ev_general", rdc_tet4_ev.h GEN)
  if (a.nen == 4 && a.variant != RDC_VARIANT_GENERIC && a.ev_general && a.use_ev && a.ev.n_wg > 0 && a.strategy == RDC_SCATTER_ROWGATHER) return launch_tet4_ev(a, k);
  if (a.nen == 4 && a.variant != RDC_VARIANT_GENERIC && a.opt_special && PihnaNoCellTransport::applies(p)) {
    // default: one thread per element visit, moments accumulated per node block (rdc_tet4_ev.hip)
    if (a.use_ev && a.ev.n_wg > 0 && a.strategy == RDC_SCATTER_ROWGATHER) return launch_tet4_ev(a, k);
    if (a.opt_slim && a.exp_mode == 3) return launch_tet4_fast<PihnaNoCellTransportSlim>(a, k);
    if (a.opt_moments) return launch_tet4_fast<PihnaNoCellTransportMoments>(a, k);  // same sums, moment form
    return launch_tet4_fast<PihnaNoCellTransport>(a, k);
  }
  return launch_rd<Pihna>(a, k);
}

template <>
hipError_t launch_specialised<Ripf, rdc_ripf_params>(const LaunchArgs& a, const Ripf::K& k, const rdc_ripf_params& p) {
  if (a.nen == 4 && a.variant != RDC_VARIANT_GENERIC && a.opt_special && RipfReduced::applies(p)) return launch_tet4_fast<RipfReduced>(a, k);
  return launch_rd<Ripf>(a, k);
}

// the all-rates-zero HCC of run/Coupled/HCC and the decay-only ADPM of run/HCP102513: any element type and kernel
template <>
hipError_t launch_specialised<Hcc, rdc_hcc_params>(const LaunchArgs& a, const Hcc::K& k, const rdc_hcc_params& p) {
  if (a.variant != RDC_VARIANT_GENERIC && a.opt_special && HccMassOnly::applies(p)) return launch_rd<HccMassOnly>(a, k);
  return launch_rd<Hcc>(a, k);
}
template <>
hipError_t launch_specialised<Adpm, rdc_adpm_params>(const LaunchArgs& a, const Adpm::K& k, const rdc_adpm_params& p) {
  if (a.variant != RDC_VARIANT_GENERIC && a.opt_special && AdpmDecayOnly::applies(p)) return launch_rd<AdpmDecayOnly>(a, k);
  return launch_rd<Adpm>(a, k);
}

// k_tet4_evc serves the all-terms Ripf instantiation: not when the reduced one will be chosen (launch_specialised<Ripf>)
template <class P> bool evc_wanted(const P*, bool) { return false; }
template <> bool evc_wanted<rdc_ripf_params>(const rdc_ripf_params* p, bool special) { return !(special && RipfReduced::applies(*p)); }

template <class P> bool pihna_pattern_applies(const P*) { return false; }
template <> bool pihna_pattern_applies<rdc_pihna_params>(const rdc_pihna_params* p) { return PihnaNoCellTransport::applies(*p); }

// two-part assembly on the element-visit lists: (re)builds the workgroup order for the current "interior_nodes":
// clusters all of whose nodes are below it first
int ev_order_for_interior(rdc_ctx* c) {
  if (c->ev_perm_interior == c->opt_interior && c->ev_perm.p) return RDC_OK;
  const std::vector<HostPrepEv::Desc>& D = c->prep_ev.desc;
  std::vector<uint32_t> perm;
  perm.reserve(D.size());
  for (size_t w = 0; w < D.size(); w++) if ((int64_t)D[w].max_node < c->opt_interior) perm.push_back((uint32_t)w);
  c->ev_part1_wg = (int)perm.size();
  for (size_t w = 0; w < D.size(); w++) if (!((int64_t)D[w].max_node < c->opt_interior)) perm.push_back((uint32_t)w);
  int rc = dev_upload(c, c->ev_perm, perm);
  if (rc) return rc;
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  c->ev_perm_interior = c->opt_interior;
  return RDC_OK;
}

// two-part assembly: number of leading row-gather workgroups whose nodes all lie inside [0, interior_nodes)
// element-visit lists: clusters are not node ranges; the rows of [0, n) are complete after part 1 when no cluster with a
// node below n reaches up to "interior_nodes"
int64_t ev_part1_node_bound(const rdc_ctx* c) {
  int64_t n = c->opt_interior;
  for (const HostPrepEv::Desc& d : c->prep_ev.desc)
    if (!((int64_t)d.max_node < c->opt_interior)) n = std::min<int64_t>(n, (int64_t)d.min_node);
  return n < 0 ? 0 : n;
}

int part1_workgroups(const rdc_ctx* c) {
  int lo = 0, hi = (int)c->prep.wg2.size();
  while (lo < hi) {
    const int mid = (lo + hi) / 2;
    if ((int64_t)c->prep.wg2[(size_t)mid].n0 + c->prep.wg2[(size_t)mid].nnodes <= c->opt_interior) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// element-visit lists (rdc_prep_ev.cpp) of the current mesh: at the upload for five unknowns (PIHNA), on the first
// assembly for three (RIPF / HCC).  A mesh they cannot describe simply keeps the pair kernels (prep_ev.ok stays false).
int build_ev_lists(rdc_ctx* c, const uint32_t* conn) {
  int rc;
  c->ev_tried = true;
  const HostPrep& P = c->prep;
  const std::string ev_err = prep_build_ev(P, conn, (size_t)c->opt_ev_lds, c->prep_ev, c->opt_interior <= P.n_owned ? c->opt_interior : -1);
  if (!ev_err.empty()) { c->prep_ev = HostPrepEv(); return RDC_OK; }
  if ((rc = dev_upload(c, c->ev_desc, c->prep_ev.desc))) return rc;
  if ((rc = dev_upload(c, c->ev_nlist, c->prep_ev.nlist))) return rc;
  if ((rc = dev_upload(c, c->ev_vloc, c->prep_ev.vloc))) return rc;
  if ((rc = dev_upload(c, c->ev_vslot, c->prep_ev.vslot))) return rc;
  if ((rc = dev_upload(c, c->ev_ntab, c->prep_ev.ntab))) return rc;
  if ((rc = dev_upload(c, c->ev_bpart, c->prep_ev.bpart))) return rc;
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  // the big host copies are not needed again (the descriptors are: two-part order)
  std::vector<uint32_t>().swap(c->prep_ev.nlist); std::vector<uint32_t>().swap(c->prep_ev.vloc);
  std::vector<uint32_t>().swap(c->prep_ev.vslot);
  return RDC_OK;
}

// cluster lists of the producer / consumer HEX8 kernels (three unknowns: solid system and reaction-diffusion models share
// them), built on first use.  solid_cl_state: 1 = ready, -1 = not available for this mesh (c->err says why).
int ensure_cluster_lists(rdc_ctx* c, int order_of_caller) {
  int rc;
  const int want_order = c->opt_solid_cl_order < 0 ? (c->solid_cl_state == 1 ? c->solid_cl_order : order_of_caller) : c->opt_solid_cl_order;
  if (c->solid_cl_state != 0 && (c->solid_cl_waves != c->opt_solid_cl_waves || c->solid_cl_order != want_order || c->scl_interior != c->opt_interior)) c->solid_cl_state = 0;
  if (c->solid_cl_state != 0) return RDC_OK;
  const int cw = c->opt_solid_cl_waves / 10, pw = c->opt_solid_cl_waves % 10;
  HostPrepCl::Limits lim;
  lim.max_nodes = cw * 8; lim.max_pairs = cw * 64; lim.max_elems = pw * 64;
  // the LDS image of the cluster's CSR rows overlays the point buffers of the solid kernel (2 x 64 pw records of 49 doubles)
  lim.max_row_doubles = (int)(2 * pw * 64 * 49) - 3 * cw * 8 - 2;
  if (c->prep.nvar == 5) {   // k_hex8_cl_rows: the image holds one equation row of the cluster's nodes and overlays its smaller point buffers
    lim.img_per_block = 5;
    lim.max_row_doubles = 5 * 27 * lim.max_nodes + 5 * lim.max_nodes;
  }
  lim.pair_order = want_order;
  HostPrepCl cl;
  std::vector<uint32_t> conn_h((size_t)c->prep.n_elem * 8);      // the context keeps the connectivity on the device only
  RDC_HIP(c, hipMemcpyAsync(conn_h.data(), c->conn.p, conn_h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  const std::string err = prep_build_cl(c->prep, conn_h.data(), lim, cl, c->opt_interior);
  c->scl_interior = c->opt_interior;
  c->solid_cl_waves = c->opt_solid_cl_waves;
  c->solid_cl_order = want_order;
  if (!err.empty()) {
    c->solid_cl_state = -1;
    std::snprintf(c->err, sizeof(c->err), "%s", err.c_str());
    return RDC_OK;
  }
  if ((rc = dev_upload(c, c->scl_desc, cl.desc))) return rc;
  if ((rc = dev_upload(c, c->scl_ntab, cl.ntab))) return rc;
  if ((rc = dev_upload(c, c->scl_eid, cl.eid))) return rc;
  if ((rc = dev_upload(c, c->scl_pair, cl.pair))) return rc;
  if ((rc = dev_upload(c, c->scl_pslot, cl.pslot))) return rc;
  RDC_HIP(c, hipStreamSynchronize(c->stream));  // the host vectors go out of scope
  c->scl_max_row_doubles = cl.max_row_doubles;
  c->scl_n_wg = (int)cl.desc.size();
  c->scl_n_wg_interior = (int)cl.n_wg_interior;
  c->scl_part1_nodes = cl.part1_nodes;
  c->solid_cl_state = 1;
  return RDC_OK;
}

// part 0: every cluster; 1: the leading clusters of interior nodes; 2: the rest (the kernels index the lists by blockIdx.x)
ClDev cluster_view(const rdc_ctx* c, int part = 0) {
  ClDev v;
  v.cw = c->solid_cl_waves / 10; v.pw = c->solid_cl_waves % 10;
  const int split = c->scl_interior >= 0 ? c->scl_n_wg_interior : 0;
  const size_t b = part == 2 ? (size_t)split : 0;
  v.n_wg = part == 1 ? split : (part == 2 ? c->scl_n_wg - split : c->scl_n_wg);
  const size_t max_nodes = (size_t)v.cw * 8, max_pairs = (size_t)v.cw * 64, max_elems = (size_t)v.pw * 64, wpp = (size_t)c->prep.nen / 4;
  v.desc = (const HostPrepCl::Desc*)c->scl_desc.p + b;
  v.ntab = (const HostPrepCl::Node*)c->scl_ntab.p + b * max_nodes;
  v.eid = (const uint32_t*)c->scl_eid.p + b * max_elems;
  v.pair = (const uint32_t*)c->scl_pair.p + b * max_pairs;
  v.pslot = (const uint32_t*)c->scl_pslot.p + b * max_pairs * wpp;
  v.max_row_doubles = c->scl_max_row_doubles;
  return v;
}

template <class M, class P>
int assemble_rd(rdc_ctx* c, const P* p, int nvar_expected, bool need_aux) {
  if (!c) return RDC_ERR_INVALID;
  if (!p) return fail(c, RDC_ERR_INVALID, "null parameter struct");
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "assemble called before rdc_mesh_upload");
  if (c->prep.nvar != nvar_expected)
    return fail(c, RDC_ERR_INVALID, "model needs nvar=%d, mesh was uploaded with nvar=%d", nvar_expected, c->prep.nvar);
  if (!c->field[RDC_FIELD_OLD_SOLUTION].p) return fail(c, RDC_ERR_STATE, "old solution field not set");
  if (need_aux && !c->field[RDC_FIELD_AUX_NODAL].p) return fail(c, RDC_ERR_STATE, "aux nodal field not set");
  if (M::NELEM > 0 && (!c->field[RDC_FIELD_ELEM_TRACTS].p || c->field_count[RDC_FIELD_ELEM_TRACTS] != (int64_t)M::NELEM * c->prep.n_elem))
    return fail(c, RDC_ERR_STATE, "per-element field (tracts) not set");
  int rc = set_device(c);
  if (rc) return rc;
  const typename M::K k = M::derive(*p);
  LaunchArgs a;
  a.m = mesh_view(c);
  a.nen = c->prep.nen;
  a.exp_mode = exp_mode_of(M::exponent(k)) == M::FAST_EXP_MODE ? M::FAST_EXP_MODE : 0;
  rc = resolve_strategy(c, &a.strategy);
  if (rc) return rc;
  a.u = (const double*)c->field[RDC_FIELD_OLD_SOLUTION].p;
  a.aux = (const double*)c->field[RDC_FIELD_AUX_NODAL].p;
  a.elem = (const double*)c->field[RDC_FIELD_ELEM_TRACTS].p;
  if (c->prep.hx_ok) {
    a.hx_nl_ptr = (const int64_t*)c->hx_nl_ptr.p;
    a.hx_nlist = (const uint32_t*)c->hx_nlist.p;
    a.hx_ploc = (const uint16_t*)c->hx_ploc.p;
    a.hx_max_nodes = c->prep.hx_max_nodes;
  }
  a.opt_staged = c->opt_staged;
  a.packed = (double*)c->packed.p;
  a.variant = c->variant;
  a.opt_occ = c->opt_occ;
  a.opt_ablate = c->opt_ablate;
  a.opt_kernel = c->opt_kernel;
  a.opt_special = c->opt_special;
  a.opt_slim = c->opt_slim;
  a.opt_moments = c->opt_moments;
  a.opt_stagger = c->opt_stagger;
  a.opt_ldspad = c->opt_ldspad;
  a.opt_xcd = c->opt_xcd;
  a.opt_grid = c->opt_grid;
  a.opt_pf = c->opt_pf;
  a.stamps = (long long*)c->stamps.p;
  if (c->prep.rg2_ok && c->prep.nen == 4) {
    a.rg2.n_wg = (int)c->prep.wg2.size();
    a.rg2.desc = (const HostPrep::WgDesc*)c->rg2_desc.p;
    a.rg2.pair_rec = (const uint32_t*)c->rg2_pair.p;
    a.rg2.chunk = (const HostPrep::Chunk*)c->rg2_chunk.p;
    a.rg2.sdesc = (const HostPrep::StoreDesc*)c->rg2_sdesc.p;
    a.rg2.contrib = (const uint16_t*)c->rg2_contrib.p;
    a.rg2.pair_aux = (const uint16_t*)c->rg2_aux.p;
    a.rg2.node_tab = (const uint16_t*)c->rg2_ntab.p;
    if (c->prep.rg4_nl_stride > 0) {
      a.rg2.nlist = (const uint32_t*)c->rg4_nlist.p;
      a.rg2.wg_ntab = (const uint16_t*)c->rg4_wgntab.p;
      a.rg2.pair_loc = (const uint32_t*)c->rg4_ploc.p;
      a.rg2.nl_stride = c->prep.rg4_nl_stride;
      if ((M::NELEM > 0 || M::AUX_LOCAL_NODE >= 0) && !c->prep.pair_eid.empty()) {
        if (!c->rg5_eid_ready) {
          if ((rc = dev_upload(c, c->rg5_eid, c->prep.pair_eid))) return rc;
          c->rg5_eid_ready = true;
        }
        a.rg2.pair_eid = (const uint32_t*)c->rg5_eid.p;
      }
    }
    a.rg2.lds_bytes = c->prep.rg2_lds_bytes;
    a.rg2.block = c->prep.rg2_block;
  }
  // RIPF with all terms on (k_tet4_evc): the element-visit lists are built on the first assembly
  const bool evc_model = EvcEligible<M>::value && evc_wanted(p, c->opt_special != 0);
  if (evc_model && a.nen == 4 && !c->ev_tried && c->prep.rg2_ok && c->opt_kernel == 0 && a.strategy == RDC_SCATTER_ROWGATHER && a.variant != RDC_VARIANT_GENERIC) {
    std::vector<uint32_t> conn_h((size_t)c->prep.n_elem * 4);      // the context keeps the connectivity on the device only
    RDC_HIP(c, hipMemcpyAsync(conn_h.data(), c->conn.p, conn_h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    RDC_HIP(c, hipStreamSynchronize(c->stream));
    if ((rc = build_ev_lists(c, conn_h.data()))) return rc;
  }
  // element-visit kernel: default for the shipped-pattern PIHNA / TET4 ("kernel" = 0 or 7); the diagnostic knobs of the
  // pair kernels (ablate, stamps, slim, coefficient form, occupancy 1) and "kernel" = 5 select k_tet4_rg5 instead
  a.use_ev = c->prep_ev.ok && (c->opt_kernel == 0 || c->opt_kernel == 7) && c->opt_moments && !c->opt_slim && (!c->opt_ablate || c->opt_kernel == 7) &&
             (!c->stamps.p || (c->opt_kernel == 7 && c->opt_ablate == 4)) && c->opt_occ != 1 && c->opt_ldspad == 0;
  a.opt_ev_occ = c->opt_ev_occ;
  a.opt_evc_occ = c->opt_evc_occ;
  a.opt_ev_resident = c->opt_ev_resident;
  a.opt_ev_bg = c->opt_ev_bg;
  if (c->opt_ev_resident) {   // cluster counter of the resident kernel
    if ((rc = dev_alloc(c, c->ev_ticket, 256))) return rc;   // [0]: whole launches and part 1, [16]: part 2
    a.ev_ticket = (int*)c->ev_ticket.p;
  }
  a.ev_grid = c->opt_grid > 0 ? c->opt_grid : 2 * c->n_cu;
  if (c->opt_kernel == 5 || c->opt_kernel == 7) a.opt_kernel = 0;
  if (a.use_ev) {
    a.ev.n_wg = (int)c->prep_ev.desc.size();
    a.ev.desc = (const HostPrepEv::Desc*)c->ev_desc.p;
    a.ev.nlist = (const uint32_t*)c->ev_nlist.p;
    a.ev.vloc = (const uint32_t*)c->ev_vloc.p;
    a.ev.vslot = (const uint32_t*)c->ev_vslot.p;
    a.ev.ntab = (const HostPrepEv::Node*)c->ev_ntab.p;
    a.ev.bpart = (const uint8_t*)c->ev_bpart.p;
    a.ev.nls = c->prep_ev.nls;
    a.ev.max_out_doubles = c->prep_ev.max_out_doubles;
  }
  // HEX8, three (five) unknowns: producer / consumer cluster kernel; a two-part call launches the interior clusters / the rest
  // (the persistent form, "hex_kernel" = 2, assembles whole meshes only)
  bool hex_cl = false;
  if (a.nen == 8 && (M::NV == 3 || M::NV == 5) && c->opt_hex_kernel != 1 && (c->opt_part == 0 || c->opt_hex_kernel == 0) &&
      a.strategy == RDC_SCATTER_ROWGATHER && c->opt_solid_cl_waves == 31) {
    if ((rc = ensure_cluster_lists(c, 0))) return rc;
    hex_cl = c->solid_cl_state == 1;
  }
  bool pattern_ok = evc_model;
  if constexpr (std::is_same<M, Pihna>::value) pattern_ok = pihna_pattern_applies(p) || c->opt_ev_general;
  const bool ev_path = a.use_ev && (std::is_same<M, Pihna>::value || evc_model) && a.nen == 4 && a.variant != RDC_VARIANT_GENERIC &&
                       (c->opt_special || c->opt_ev_general || !std::is_same<M, Pihna>::value) && a.strategy == RDC_SCATTER_ROWGATHER && pattern_ok;
  if constexpr (std::is_same<M, Pihna>::value) a.ev_general = ev_path && c->opt_ev_general && !(c->opt_special && pihna_pattern_applies(p));
  if (!std::is_same<M, Pihna>::value) a.use_ev = ev_path && c->opt_kernel == 0;   // k_tet4_evc (rdc_tet4_fast.hip dispatches on it)
  if (c->opt_part != 0 && ev_path) {
    // two-part assembly on the element-visit lists: the clusters all of whose nodes are interior run in part 1
    if (c->opt_interior < 0) return fail(c, RDC_ERR_STATE, "\"part\" needs \"interior_nodes\"");
    if ((rc = ev_order_for_interior(c))) return rc;
    a.ev.wg_perm = (const uint32_t*)c->ev_perm.p;
    if (c->opt_part == 1) {
      c->part1_packed = false;
      c->part1_nodes = 0;
      if (c->ev_part1_wg == 0) return RDC_OK;
      c->part1_nodes = ev_part1_node_bound(c);
      a.ev.wg_begin = 0; a.ev.wg_count = c->ev_part1_wg;
      if (!c->pack_event) RDC_HIP(c, hipEventCreateWithFlags(&c->pack_event, hipEventDisableTiming));
      a.pack_part = 1; a.pack_event = c->pack_event;
      c->part1_packed = true;
    } else {
      a.ev.wg_begin = c->ev_part1_wg; a.ev.wg_count = -1;
      if (c->part1_packed) { a.pack_part = 2; a.pack_event = c->pack_event; }
      c->part1_packed = false;
    }
  } else
  if (c->opt_part != 0 && hex_cl) {
    // HEX8 cluster kernels: the cluster lists respect "interior_nodes" (interior clusters first)
    c->part1_packed = false;
    if (c->opt_part == 1) {
      c->part1_nodes = c->scl_interior >= 0 ? c->scl_part1_nodes : 0;
      if (c->scl_interior < 0 || c->scl_n_wg_interior == 0) { c->part1_nodes = 0; return RDC_OK; }
    }
    a.cl = cluster_view(c, c->scl_interior >= 0 ? c->opt_part : (c->opt_part == 2 ? 0 : 1));
    if (a.cl.n_wg == 0) return RDC_OK;
  } else
  if (c->opt_part != 0) {
    // two-part assembly (halo overlap): part 1 = the leading workgroups whose nodes are all interior, part 2 = the
    // rest.  Only the default TET4 row-gather kernel launches sub-ranges; every other path assembles everything in
    // part 2 and nothing in part 1.
    const bool sub = a.nen == 4 && a.strategy == RDC_SCATTER_ROWGATHER && a.variant != RDC_VARIANT_GENERIC && a.rg2.n_wg > 0 &&
                     a.rg2.pair_aux && a.rg2.nlist && a.rg2.block == 256 && a.opt_kernel == 0 && c->opt_interior >= 0 &&
                     ((M::NELEM == 0 && M::AUX_LOCAL_NODE < 0) || a.rg2.pair_eid);
    const int split = sub ? part1_workgroups(c) : 0;
    if (c->opt_part == 1) {
      c->part1_packed = false;
      c->part1_nodes = 0;
      if (!sub || split == 0) return RDC_OK;
      c->part1_nodes = (int64_t)c->prep.wg2[(size_t)split - 1].n0 + c->prep.wg2[(size_t)split - 1].nnodes;
      a.rg2.wg_begin = 0; a.rg2.wg_count = split;
      // part 1 packs the records of the owned nodes only and part 2 those of the ghosts, ordered by an event: the two
      // parts may run on different streams (rdc_assembly.h, stream contract of the two-part assembly)
      if (!c->pack_event) RDC_HIP(c, hipEventCreateWithFlags(&c->pack_event, hipEventDisableTiming));
      a.pack_part = 1; a.pack_event = c->pack_event;
      c->part1_packed = true;
    } else {
      a.rg2.wg_begin = split; a.rg2.wg_count = -1;
      if (sub && c->part1_packed) { a.pack_part = 2; a.pack_event = c->pack_event; }
      c->part1_packed = false;
    }
  } else {
    c->part1_packed = false;
  }
  if (hex_cl && c->opt_part == 0) {
    a.cl = cluster_view(c);
    a.cl.grid = c->opt_hex_kernel == 2 ? 2 * c->n_cu : 0;   // persistent form: the workgroups resident at once
  }
  a.val = (double*)c->val.p;
  a.rhs = (double*)c->rhs.p;
  a.stream = c->stream;
  a.colour_ptr = c->prep.colour_ptr.data();
  a.n_colours = c->prep.n_colours;
  a.n_wg = c->prep.rowgather_ok ? (int)c->prep.wg_node_ptr.size() - 1 : 0;
  a.lds_bytes = c->prep.rg_lds_bytes;
  a.ev_start = nullptr;
  hipEvent_t ev_stop = nullptr;
  if (c->timing) {
    if ((rc = next_event_pair(c, &a.ev_start, &ev_stop))) return rc;
  }
  hipError_t e = launch_specialised<M>(a, k, *p);
  if (e != hipSuccess) return fail(c, RDC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
  if (c->timing) RDC_HIP(c, hipEventRecord(ev_stop, c->stream));
  return RDC_OK;
}

// one call per part of a two-part step: "part" and the stream for this call only (a step of the multi-GPU harness is then
// two C-ABI calls instead of six: set_option, assemble, set_stream, set_option, assemble, set_stream)
template <class F>
int with_part(rdc_ctx* c, int part, void* stream, F&& call) {
  if (!c) return RDC_ERR_INVALID;
  if (part < 0 || part > 2) return fail(c, RDC_ERR_INVALID, "part must be 0, 1 or 2");
  const int part0 = c->opt_part;
  const hipStream_t s0 = c->stream;
  c->opt_part = part;
  c->stream = (hipStream_t)stream;
  const int rc = call();
  c->opt_part = part0;
  c->stream = s0;
  return rc;
}
}  // namespace

extern "C" {

int rdc_abi_version(void) { return RDC_ABI_VERSION; }

const char* rdc_last_error(const rdc_ctx* ctx) { return ctx ? ctx->err : g_create_error; }

int rdc_device_count(int* n) {
  if (!n) return RDC_ERR_INVALID;
  int ndev = 0;
  const hipError_t e = hipGetDeviceCount(&ndev);
  *n = e == hipSuccess ? ndev : 0;
  return e == hipSuccess ? RDC_OK : fail(nullptr, RDC_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
}

int rdc_ctx_create(int device_ordinal, rdc_ctx** out) {
  if (!out) return fail(nullptr, RDC_ERR_INVALID, "null output pointer");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, RDC_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device_ordinal < 0 || device_ordinal >= ndev)
    return fail(nullptr, RDC_ERR_INVALID, "device ordinal %d out of range [0,%d)", device_ordinal, ndev);
  rdc_ctx* c = new (std::nothrow) rdc_ctx();
  if (!c) return fail(nullptr, RDC_ERR_ALLOC, "out of host memory");
  c->device = device_ordinal;
  e = hipSetDevice(device_ordinal);
  if (e != hipSuccess) {
    delete c;
    return fail(nullptr, RDC_ERR_HIP, "device initialisation failed: %s", hipGetErrorString(e));
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess) {
    if (prop.sharedMemPerBlock > 0) c->max_lds = prop.sharedMemPerBlock;
    if (prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
  }
  *out = c;
  return RDC_OK;
}

int rdc_ctx_destroy(rdc_ctx* c) {
  if (!c) return RDC_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  DevBuf* all[] = {&c->conn, &c->xyz, &c->bptr, &c->eslot, &c->elem_order, &c->first_mask, &c->first_rhs,
                   &c->pair_elem, &c->pair_local, &c->node_pair_ptr, &c->wg_node_ptr, &c->val, &c->rhs, &c->packed, &c->stamps, &c->rg2_desc, &c->rg2_pair, &c->rg2_chunk,
                   &c->rg2_sdesc, &c->rg2_contrib, &c->rg2_aux, &c->rg2_ntab, &c->rg4_nlist, &c->rg4_ploc, &c->rg4_wgntab, &c->rg5_eid, &c->hx_nl_ptr, &c->hx_nlist, &c->hx_ploc,
                   &c->elem_material, &c->materials, &c->side_elem, &c->side_id, &c->side_disp,
                   &c->scl_desc, &c->scl_ntab, &c->scl_eid, &c->scl_pair, &c->scl_pslot, &c->solid_ke, &c->solid_fe, &c->sg_gptr, &c->sg_gsrc, &c->sg_brow, &c->wg_max, &c->solid_post, &c->adpm_slot,
                   &c->ev_desc, &c->ev_nlist, &c->ev_vloc, &c->ev_vslot, &c->ev_ntab, &c->ev_bpart, &c->ev_perm, &c->ev_ticket};
  for (DevBuf* b : all) dev_free(c, *b);
  for (int f = 0; f < RDC_FIELD_COUNT; f++) dev_free(c, c->field[f]);
  for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
  if (c->pack_event) (void)hipEventDestroy(c->pack_event);
  if (c->solid_part1_event) (void)hipEventDestroy(c->solid_part1_event);
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  if (c->copy_fence) (void)hipEventDestroy(c->copy_fence);
  for (hipEvent_t e : c->ticket) if (e) (void)hipEventDestroy(e);
  delete c;
  return RDC_OK;
}

int rdc_set_stream(rdc_ctx* c, void* s) {
  if (!c) return RDC_ERR_INVALID;
  c->stream = (hipStream_t)s;
  return RDC_OK;
}

int rdc_synchronize(rdc_ctx* c) {
  if (!c) return RDC_ERR_INVALID;
  int rc = set_device(c);
  if (rc) return rc;
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  return RDC_OK;
}

int rdc_set_scatter(rdc_ctx* c, int s) {
  if (!c) return RDC_ERR_INVALID;
  if (s != RDC_SCATTER_AUTO && s != RDC_SCATTER_COLOURED && s != RDC_SCATTER_ROWGATHER)
    return fail(c, RDC_ERR_INVALID, "unknown scatter strategy %d", s);
  c->strategy = s;
  return RDC_OK;
}

int rdc_set_kernel_variant(rdc_ctx* c, int v) {
  if (!c) return RDC_ERR_INVALID;
  if (v != RDC_VARIANT_AUTO && v != RDC_VARIANT_GENERIC) return fail(c, RDC_ERR_INVALID, "unknown kernel variant %d", v);
  c->variant = v;
  return RDC_OK;
}

int rdc_set_option(rdc_ctx* c, const char* key, int value) {
  if (!c || !key) return RDC_ERR_INVALID;
  if (!std::strcmp(key, "occupancy")) c->opt_occ = value;
  else if (!std::strcmp(key, "ablate")) c->opt_ablate = value;
  else if (!std::strcmp(key, "block")) {
    if (value != 128 && value != 256) return fail(c, RDC_ERR_INVALID, "block must be 128 or 256");
    c->opt_block = value;
  }
  else if (!std::strcmp(key, "specialise")) c->opt_special = value;
  else if (!std::strcmp(key, "slim")) c->opt_slim = value;
  else if (!std::strcmp(key, "lds_pad")) c->opt_ldspad = value;  // k_tet4_rg5: KB of unused LDS per workgroup (diagnostic: fewer co-resident workgroups)
  else if (!std::strcmp(key, "stagger")) c->opt_stagger = value;  // k_tet4_rg5: start delay of every CU's second workgroup, in units of 1024 cycles
  else if (!std::strcmp(key, "moments")) c->opt_moments = value;  // 1 (default): shipped-pattern PIHNA/TET4 rows in moment form (rdc_tet4_pihna_moments.h), 0: coefficient form
  else if (!std::strcmp(key, "interior_nodes")) c->opt_interior = value;  // see rdc_assembly.h (two-part assembly)
  else if (!std::strcmp(key, "part")) {
    if (value < 0 || value > 2) return fail(c, RDC_ERR_INVALID, "part must be 0, 1 or 2");
    c->opt_part = value;
  }
  else if (!std::strcmp(key, "staged")) c->opt_staged = value;  // HEX8 generic row gather: node table in LDS (default 1)
  else if (!std::strcmp(key, "xcd")) c->opt_xcd = value;
  else if (!std::strcmp(key, "schedule")) c->opt_sched = value;
  else if (!std::strcmp(key, "grid")) c->opt_grid = value;
  else if (!std::strcmp(key, "prefetch")) c->opt_pf = value;
  else if (!std::strcmp(key, "solid_gather")) c->opt_solid_gather = value ? 1 : 0;
  else if (!std::strcmp(key, "solid_split")) c->opt_solid_split = value ? 1 : 0;
  else if (!std::strcmp(key, "solid_store")) c->opt_solid_store = value;
  else if (!std::strcmp(key, "solid_kernel")) {
    if (value < 0 || value > 3) return fail(c, RDC_ERR_INVALID, "solid_kernel must be 0 (default), 1 (coloured), 2 (two-pass) or 3 (fused cluster kernel)");
    c->opt_solid_kernel = value;
  } else if (!std::strcmp(key, "hex_kernel")) {
    if (value < 0 || value > 2) return fail(c, RDC_ERR_INVALID, "hex_kernel must be 0 (cluster kernel), 1 (pair kernels) or 2 (persistent cluster kernel)");
    c->opt_hex_kernel = value;
  } else if (!std::strcmp(key, "solid_cl_order")) {
    c->opt_solid_cl_order = value < 0 ? -1 : (value ? 1 : 0);
  } else if (!std::strcmp(key, "solid_cl_waves")) {
    if (value != 31 && value != 62) return fail(c, RDC_ERR_INVALID, "solid_cl_waves must be 31 (3 consumer + 1 producer waves) or 62");
    c->opt_solid_cl_waves = value;
  }
  else if (!std::strcmp(key, "ev_background")) c->opt_ev_bg = value ? 1 : 0;   // element-visit kernel: skip the moments that are sums of zeros in the background state (1, default)
  else if (!std::strcmp(key, "ev_general")) c->opt_ev_general = value ? 1 : 0;   // general-parameter PIHNA / TET4: element-visit kernel with 22 moments (1, default) or the pair kernel (0)
  else if (!std::strcmp(key, "ev_resident")) c->opt_ev_resident = value == 2 ? 2 : (value ? 1 : 0);   // 2 = also for small launches (tests, tools/ev_timeline.py); whole-mesh launches of the element-visit kernel as three resident, pipelined workgroups per CU (k_tet4_evq; default 1)
  else if (!std::strcmp(key, "evc_occupancy")) c->opt_evc_occ = value == 3 ? 3 : 2;   // k_tet4_evc: waves per SIMD its registers are bounded for
  else if (!std::strcmp(key, "ev_occupancy")) c->opt_ev_occ = value;   // element-visit kernel: 3 (default, 168 registers) or 2
  else if (!std::strcmp(key, "ev_lds")) c->opt_ev_lds = value;   // LDS bytes per workgroup the element-visit clusters are sized for (next rdc_mesh_upload)
  else if (!std::strcmp(key, "kernel")) c->opt_kernel = value;  // 0 = default (LDS-staged node records, k_tet4_rg5), 3 = k_tet4_rg3, 4 = persistent k_tet4_rg4, 6 = k_tet4_rg6 (rg5 over work items with a tail prefetch, experimental),
                                                                 // 1 = first row-gather kernel, 2 = staged deterministic k_tet4_rg2
  else return fail(c, RDC_ERR_INVALID, "unknown option '%s'", key);
  return RDC_OK;
}

int rdc_get_scatter(const rdc_ctx* c, int* s) {
  if (!c || !s) return RDC_ERR_INVALID;
  int r = c->strategy;
  if (r == RDC_SCATTER_AUTO && c->have_mesh) r = c->prep.rowgather_ok ? RDC_SCATTER_ROWGATHER : RDC_SCATTER_COLOURED;
  *s = r;
  return RDC_OK;
}

int rdc_mesh_upload(rdc_ctx* c, int elem_type, int64_t n_elem, int64_t n_node, int64_t n_owned,
                    const uint32_t* conn, const double* xyz, int nvar) {
  if (!c) return RDC_ERR_INVALID;
  if (!conn || !xyz) return fail(c, RDC_ERR_INVALID, "null mesh arrays");
  int rc = set_device(c);
  if (rc) return rc;
  c->have_mesh = false;
  // LDS budget of a row-gather workgroup: half the per-block limit keeps two workgroups per CU
  const size_t budget = c->max_lds >= 64 * 1024 ? 50 * 1024 : c->max_lds / 2;
  std::string err = prep_build(elem_type, n_elem, n_node, n_owned, conn, nvar, budget, c->opt_block, c->prep, c->opt_sched != 0);
  if (!err.empty()) return fail(c, RDC_ERR_INVALID, "%s", err.c_str());
  const HostPrep& P = c->prep;
  std::vector<uint32_t> conn_v(conn, conn + n_elem * elem_type);
  std::vector<double> xyz_v(xyz, xyz + n_node * 3);
  if ((rc = dev_upload(c, c->conn, conn_v))) return rc;
  if ((rc = dev_upload(c, c->xyz, xyz_v))) return rc;
  if ((rc = dev_upload(c, c->bptr, P.bptr))) return rc;
  if ((rc = dev_upload(c, c->eslot, P.eslot))) return rc;
  if ((rc = dev_upload(c, c->elem_order, P.elem_order))) return rc;
  if ((rc = dev_upload(c, c->first_mask, P.first_mask))) return rc;
  if ((rc = dev_upload(c, c->first_rhs, P.first_rhs))) return rc;
  if ((rc = dev_upload(c, c->pair_elem, P.pair_elem))) return rc;
  if ((rc = dev_upload(c, c->pair_local, P.pair_local))) return rc;
  if ((rc = dev_upload(c, c->node_pair_ptr, P.node_pair_ptr))) return rc;
  if ((rc = dev_upload(c, c->wg_node_ptr, P.wg_node_ptr))) return rc;
  if (P.hx_ok) {
    if ((rc = dev_upload(c, c->hx_nl_ptr, P.hx_nl_ptr))) return rc;
    if ((rc = dev_upload(c, c->hx_nlist, P.hx_nlist))) return rc;
    if ((rc = dev_upload(c, c->hx_ploc, P.hx_ploc))) return rc;
  }
  if (P.rg2_ok && elem_type == RDC_TET4) {
    if ((rc = dev_upload(c, c->rg2_desc, P.wg2))) return rc;
    if ((rc = dev_upload(c, c->rg2_pair, P.pair_rec))) return rc;
    if ((rc = dev_upload(c, c->rg2_chunk, P.chunk))) return rc;
    if ((rc = dev_upload(c, c->rg2_sdesc, P.sdesc))) return rc;
    if ((rc = dev_upload(c, c->rg2_contrib, P.contrib))) return rc;
    if ((rc = dev_upload(c, c->rg2_aux, P.pair_aux))) return rc;
    if ((rc = dev_upload(c, c->rg2_ntab, P.node_tab))) return rc;
    if (P.rg4_nl_stride > 0) {
      if ((rc = dev_upload(c, c->rg4_nlist, P.nlist))) return rc;
      if ((rc = dev_upload(c, c->rg4_wgntab, P.wg_ntab))) return rc;
      if ((rc = dev_upload(c, c->rg4_ploc, P.pair_loc))) return rc;
    }
  }
  c->prep_ev = HostPrepEv();
  c->part1_nodes = -1;
  c->ev_perm_interior = -2;
  c->ev_tried = false;
  if (elem_type == RDC_TET4 && nvar == 5 && P.rg2_ok) {
    // element-visit lists of the PIHNA kernel; a mesh they cannot describe simply keeps the pair kernels
    // "interior_nodes" set BEFORE the upload lets the clusters respect the interior / near-ghost split (two-part assembly)
    if ((rc = build_ev_lists(c, conn))) return rc;
  }
  const size_t nnz = (size_t)nvar * nvar * P.bptr[n_owned];
  if ((rc = dev_alloc(c, c->val, nnz * sizeof(double)))) return rc;
  if ((rc = dev_alloc(c, c->rhs, (size_t)n_owned * nvar * sizeof(double)))) return rc;
  if (elem_type == RDC_TET4 && (rc = dev_alloc(c, c->packed, (size_t)n_node * 12 * sizeof(double)))) return rc;
  RDC_HIP(c, hipMemsetAsync(c->val.p, 0, c->val.bytes, c->stream));
  RDC_HIP(c, hipMemsetAsync(c->rhs.p, 0, c->rhs.bytes, c->stream));
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  // fields are tied to the mesh sizes: drop library-owned ones
  for (int f = 0; f < RDC_FIELD_COUNT; f++) { dev_free(c, c->field[f]); c->field_count[f] = 0; }
  dev_free(c, c->elem_material); dev_free(c, c->materials); dev_free(c, c->adpm_slot);
  dev_free(c, c->side_elem); dev_free(c, c->side_id); dev_free(c, c->side_disp);
  c->n_materials = 0; c->n_sides = 0;
  c->have_mesh = true;
  c->solid_gather_ready = false;
  c->solid_cl_state = 0;
  c->rg5_eid_ready = false;
  return RDC_OK;
}

int rdc_mesh_update_coords(rdc_ctx* c, const double* xyz) {
  if (!c) return RDC_ERR_INVALID;
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  if (!xyz) return fail(c, RDC_ERR_INVALID, "null coordinates");
  int rc = set_device(c);
  if (rc) return rc;
  RDC_HIP(c, hipMemcpyAsync(c->xyz.p, xyz, (size_t)c->prep.n_node * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  return RDC_OK;
}

int rdc_mesh_coords_device_ptr(rdc_ctx* c, double** d) {
  if (!c || !d) return RDC_ERR_INVALID;
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  *d = (double*)c->xyz.p;
  return RDC_OK;
}

int rdc_mesh_dims(const rdc_ctx* c, int64_t* n_elem, int64_t* n_node, int64_t* n_owned, int* elem_type,
                  int* nvar, int* n_colours) {
  if (!c || !c->have_mesh) return RDC_ERR_STATE;
  if (n_elem) *n_elem = c->prep.n_elem;
  if (n_node) *n_node = c->prep.n_node;
  if (n_owned) *n_owned = c->prep.n_owned;
  if (elem_type) *elem_type = c->prep.nen;
  if (nvar) *nvar = c->prep.nvar;
  if (n_colours) *n_colours = c->prep.n_colours;
  return RDC_OK;
}

int rdc_csr_dims(const rdc_ctx* c, int64_t* n_rows, int64_t* nnz) {
  if (!c || !c->have_mesh) return RDC_ERR_STATE;
  if (n_rows) *n_rows = c->prep.n_owned * c->prep.nvar;
  if (nnz) *nnz = (int64_t)c->prep.nvar * c->prep.nvar * c->prep.bptr[c->prep.n_owned];
  return RDC_OK;
}

int rdc_csr_pattern_download(const rdc_ctx* c, int64_t* row_ptr, int32_t* col_idx) {
  if (!c || !c->have_mesh) return RDC_ERR_STATE;
  if (!row_ptr || !col_idx) return RDC_ERR_INVALID;
  const HostPrep& P = c->prep;
  const int nv = P.nvar;
  row_ptr[0] = 0;
  for (int64_t n = 0; n < P.n_owned; n++) {
    const int64_t len = P.bptr[n + 1] - P.bptr[n];
    for (int a = 0; a < nv; a++) {
      const int64_t r = n * nv + a;
      row_ptr[r + 1] = row_ptr[r] + len * nv;
      int32_t* o = col_idx + row_ptr[r];
      for (int64_t k = 0; k < len; k++)
        for (int b = 0; b < nv; b++) *o++ = P.bcol[P.bptr[n] + k] * nv + b;
    }
  }
  return RDC_OK;
}

int rdc_mesh_colours_download(const rdc_ctx* c, int32_t* colour) {
  if (!c || !c->have_mesh) return RDC_ERR_STATE;
  if (!colour) return RDC_ERR_INVALID;
  std::memcpy(colour, c->prep.colour.data(), sizeof(int32_t) * (size_t)c->prep.n_elem);
  return RDC_OK;
}

int rdc_field_device_ptr(rdc_ctx* c, int field, int64_t count, double** d_ptr) {
  if (!c || !d_ptr) return RDC_ERR_INVALID;
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  if (field < 0 || field >= RDC_FIELD_COUNT) return fail(c, RDC_ERR_INVALID, "unknown field %d", field);
  if (count != field_expected(c, field))
    return fail(c, RDC_ERR_INVALID, "field %d needs %lld values, got %lld", field, (long long)field_expected(c, field), (long long)count);
  int rc = set_device(c);
  if (rc) return rc;
  DevBuf& b = c->field[field];
  if (!b.p || !b.owned || c->field_count[field] != count) {
    if (b.p && !b.owned) b = DevBuf();
    if ((rc = dev_alloc(c, b, (size_t)count * sizeof(double)))) return rc;
    c->field_count[field] = count;
  }
  *d_ptr = (double*)b.p;
  return RDC_OK;
}

int rdc_field_upload(rdc_ctx* c, int field, const double* host, int64_t count) {
  if (!c) return RDC_ERR_INVALID;
  if (!host) return fail(c, RDC_ERR_INVALID, "null host pointer");
  double* d = nullptr;
  int rc = rdc_field_device_ptr(c, field, count, &d);
  if (rc) return rc;
  RDC_HIP(c, hipMemcpyAsync(d, host, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  return RDC_OK;
}

int rdc_field_download(rdc_ctx* c, int field, double* host, int64_t count) {
  if (!c) return RDC_ERR_INVALID;
  if (!host) return fail(c, RDC_ERR_INVALID, "null host pointer");
  if (field < 0 || field >= RDC_FIELD_COUNT || !c->field[field].p) return fail(c, RDC_ERR_STATE, "field %d not set", field);
  if (count != c->field_count[field]) return fail(c, RDC_ERR_INVALID, "field %d holds %lld values", field, (long long)c->field_count[field]);
  int rc = set_device(c);
  if (rc) return rc;
  RDC_HIP(c, hipMemcpyAsync(host, c->field[field].p, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  return RDC_OK;
}

int rdc_field_bind_device(rdc_ctx* c, int field, double* d_ptr, int64_t count) {
  if (!c) return RDC_ERR_INVALID;
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  if (field < 0 || field >= RDC_FIELD_COUNT) return fail(c, RDC_ERR_INVALID, "unknown field %d", field);
  if (!d_ptr) return fail(c, RDC_ERR_INVALID, "null device pointer");
  if (count != field_expected(c, field))
    return fail(c, RDC_ERR_INVALID, "field %d needs %lld values, got %lld", field, (long long)field_expected(c, field), (long long)count);
  int rc = dev_free(c, c->field[field]);
  if (rc) return rc;
  c->field[field].p = d_ptr;
  c->field[field].bytes = (size_t)count * sizeof(double);
  c->field[field].owned = false;
  c->field_count[field] = count;
  return RDC_OK;
}

int rdc_solid_set_materials(rdc_ctx* c, const int32_t* elem_material, int32_t n_materials,
                            const rdc_solid_material* materials) {
  if (!c) return RDC_ERR_INVALID;
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  if (!elem_material || !materials || n_materials <= 0) return fail(c, RDC_ERR_INVALID, "bad material table");
  for (int64_t e = 0; e < c->prep.n_elem; e++)
    if (elem_material[e] < 0 || elem_material[e] >= n_materials) return fail(c, RDC_ERR_INVALID, "material index out of range at element %lld", (long long)e);
  int rc = set_device(c);
  if (rc) return rc;
  std::vector<int32_t> em(elem_material, elem_material + c->prep.n_elem);
  std::vector<rdc_solid_material> mt(materials, materials + n_materials);
  if ((rc = dev_upload(c, c->elem_material, em))) return rc;
  if ((rc = dev_upload(c, c->materials, mt))) return rc;
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  c->n_materials = n_materials;
  return RDC_OK;
}

int rdc_solid_set_sides(rdc_ctx* c, int64_t n_sides, const int64_t* side_elem, const int32_t* side_id,
                        const double* side_disp) {
  if (!c) return RDC_ERR_INVALID;
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  if (n_sides < 0 || (n_sides > 0 && (!side_elem || !side_id || !side_disp))) return fail(c, RDC_ERR_INVALID, "bad side list");
  const int nsides_elem = c->prep.nen == 4 ? 4 : 6;
  for (int64_t s = 0; s < n_sides; s++) {
    if (side_elem[s] < 0 || side_elem[s] >= c->prep.n_elem) return fail(c, RDC_ERR_INVALID, "side %lld: element out of range", (long long)s);
    if (side_id[s] < 0 || side_id[s] >= nsides_elem) return fail(c, RDC_ERR_INVALID, "side %lld: side id out of range", (long long)s);
  }
  int rc = set_device(c);
  if (rc) return rc;
  std::vector<int64_t> se(side_elem, side_elem + n_sides);
  std::vector<int32_t> si(side_id, side_id + n_sides);
  std::vector<double> sd(side_disp, side_disp + 3 * n_sides);
  if ((rc = dev_upload(c, c->side_elem, se))) return rc;
  if ((rc = dev_upload(c, c->side_id, si))) return rc;
  if ((rc = dev_upload(c, c->side_disp, sd))) return rc;
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  c->n_sides = n_sides;
  return RDC_OK;
}

int rdc_assemble_pihna(rdc_ctx* c, const rdc_pihna_params* p) { return assemble_rd<Pihna>(c, p, 5, false); }
int rdc_assemble_ripf(rdc_ctx* c, const rdc_ripf_params* p) { return assemble_rd<Ripf>(c, p, 3, true); }
int rdc_assemble_hcc(rdc_ctx* c, const rdc_hcc_params* p) { return assemble_rd<Hcc>(c, p, 3, false); }
int rdc_assemble_adpm(rdc_ctx* c, const rdc_adpm_params* p) { return assemble_rd<Adpm>(c, p, 3, false); }
int rdc_assemble_proteas(rdc_ctx* c, const rdc_proteas_params* p) { return assemble_rd<Proteas>(c, p, 5, true); }

int rdc_assemble_pihna_part(rdc_ctx* c, const rdc_pihna_params* p, int part, void* stream) {
  return with_part(c, part, stream, [&] { return assemble_rd<Pihna>(c, p, 5, false); });
}
int rdc_assemble_hcc_part(rdc_ctx* c, const rdc_hcc_params* p, int part, void* stream) {
  return with_part(c, part, stream, [&] { return assemble_rd<Hcc>(c, p, 3, false); });
}
int rdc_solid_assemble_part(rdc_ctx* c, const rdc_solid_params* p, int request_jacobian, int part, void* stream) {
  return with_part(c, part, stream, [&] { return rdc_solid_assemble(c, p, request_jacobian); });
}

int rdc_solid_assemble(rdc_ctx* c, const rdc_solid_params* p, int request_jacobian) {
  if (!c) return RDC_ERR_INVALID;
  if (!p) return fail(c, RDC_ERR_INVALID, "null parameter struct");
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "assemble called before rdc_mesh_upload");
  if (c->prep.nvar != 3) return fail(c, RDC_ERR_INVALID, "solid system needs nvar=3");
  if (!c->field[RDC_FIELD_UNDEFORMED_XYZ].p) return fail(c, RDC_ERR_STATE, "undeformed coordinates not set");
  if (!c->field[RDC_FIELD_ELEM_FIBRE].p) return fail(c, RDC_ERR_STATE, "fibre field not set");
  if (c->n_materials <= 0) return fail(c, RDC_ERR_STATE, "materials not set");
  int rc = set_device(c);
  if (rc) return rc;
  SolidArgs a;
  a.m = mesh_view(c);
  a.nen = c->prep.nen;
  a.Xu = (const double*)c->field[RDC_FIELD_UNDEFORMED_XYZ].p;
  a.fibre = (const double*)c->field[RDC_FIELD_ELEM_FIBRE].p;
  a.elem_material = (const int32_t*)c->elem_material.p;
  a.materials = (const rdc_solid_material*)c->materials.p;
  a.n_sides = c->n_sides;
  a.side_elem = (const int64_t*)c->side_elem.p;
  a.side_id = (const int32_t*)c->side_id.p;
  a.side_disp = (const double*)c->side_disp.p;
  a.params = *p;
  a.request_jacobian = request_jacobian;
  a.val = (double*)c->val.p;
  a.rhs = (double*)c->rhs.p;
  a.stream = c->stream;
  a.colour_ptr = c->prep.colour_ptr.data();
  a.n_colours = c->prep.n_colours;
  a.gather = c->opt_solid_gather;
  a.split = c->opt_solid_split;
  a.store_mode = c->opt_solid_store;
  a.nblocks = c->prep.bptr[(size_t)c->prep.n_owned];
  // kernel choice: the fused cluster kernel serves HEX8 tangent requests; everything else is two-pass (or coloured on request)
  int kernel = c->opt_solid_kernel == 1 ? 1 : 0;
  if ((c->opt_solid_kernel == 0 || c->opt_solid_kernel == 3) && c->prep.nen == 8 && request_jacobian) {
    if ((rc = ensure_cluster_lists(c, 1))) return rc;
    if (c->solid_cl_state != 1 && c->opt_solid_kernel == 3) return fail(c, RDC_ERR_UNSUPPORTED, "fused solid kernel: %s", c->err);
    if (c->solid_cl_state == 1) {
      kernel = 3;
      a.cl = cluster_view(c, c->scl_interior >= 0 ? c->opt_part : (c->opt_part == 1 ? 1 : 0));
    }
  } else if (c->opt_solid_kernel == 3) {
    return fail(c, RDC_ERR_UNSUPPORTED, "fused solid kernel: HEX8 tangent requests only");
  }
  // two-part assembly (halo overlap): the fused cluster kernel launches the clusters of interior nodes in part 1 and the rest
  // (+ the penalty sides, which add into rows of both kinds) in part 2; the two-pass and coloured forms assemble everything in part 2
  if (c->opt_part == 1) {
    c->part1_nodes = 0;
    if (kernel != 3 || c->scl_interior < 0 || a.cl.n_wg == 0) return RDC_OK;
    c->part1_nodes = c->scl_part1_nodes;
    a.n_sides = 0;
    if (!c->solid_part1_event) RDC_HIP(c, hipEventCreateWithFlags(&c->solid_part1_event, hipEventDisableTiming));
    a.done_record = c->solid_part1_event;
    c->solid_part1_pending = true;
  } else if (c->opt_part == 2 && c->solid_part1_pending) {
    a.sides_wait = c->solid_part1_event;
    c->solid_part1_pending = false;
  } else {
    c->solid_part1_pending = false;
  }
  a.kernel = kernel;
  if (a.kernel == 0) {
    if (!c->solid_gather_ready) {  // one-time: gather lists and the element-matrix buffers
      SolidGather g;
      const std::string err = solid_gather_build(c->prep, g);
      if (!err.empty()) return fail(c, RDC_ERR_UNSUPPORTED, "%s", err.c_str());
      if ((rc = dev_upload(c, c->sg_gptr, g.gptr))) return rc;
      if ((rc = dev_upload(c, c->sg_gsrc, g.gsrc))) return rc;
      if ((rc = dev_upload(c, c->sg_brow, g.brow))) return rc;
      const size_t rows = (size_t)c->prep.n_elem * c->prep.nen;
      if ((rc = dev_alloc(c, c->solid_ke, rows * c->prep.nen * 9 * sizeof(double)))) return rc;
      if ((rc = dev_alloc(c, c->solid_fe, rows * 3 * sizeof(double)))) return rc;
      RDC_HIP(c, hipStreamSynchronize(c->stream));  // the host vectors go out of scope
      c->solid_gather_ready = true;
    }
    a.ke = (double*)c->solid_ke.p;
    a.fe = (double*)c->solid_fe.p;
    a.gptr = (const uint32_t*)c->sg_gptr.p;
    a.gsrc = (const uint32_t*)c->sg_gsrc.p;
    a.brow = (const int32_t*)c->sg_brow.p;
  } else {
    a.ke = a.fe = nullptr;
    a.gptr = a.gsrc = nullptr;
    a.brow = nullptr;
  }
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  if (a.sides_wait) {   // part 2 starts behind part 1 (its penalty sides add into rows of both parts); in a step it follows the halo exchange anyway
    RDC_HIP(c, hipStreamWaitEvent(c->stream, a.sides_wait, 0));
    a.sides_wait = nullptr;
  }
  if (c->timing) {
    if ((rc = next_event_pair(c, &ev_start, &ev_stop))) return rc;
    RDC_HIP(c, hipEventRecord(ev_start, c->stream));
  }
  hipError_t e = launch_solid(a);
  if (e != hipSuccess) return fail(c, RDC_ERR_HIP, "solid kernel launch failed: %s", hipGetErrorString(e));
  if (c->timing) RDC_HIP(c, hipEventRecord(ev_stop, c->stream));
  return RDC_OK;
}

int rdc_csr_values_device_ptr(rdc_ctx* c, double** d_val, double** d_rhs) {
  if (!c) return RDC_ERR_INVALID;
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  if (d_val) *d_val = (double*)c->val.p;
  if (d_rhs) *d_rhs = (double*)c->rhs.p;
  return RDC_OK;
}

int rdc_csr_download(rdc_ctx* c, double* val, double* rhs) {
  if (!c) return RDC_ERR_INVALID;
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  int rc = set_device(c);
  if (rc) return rc;
  const size_t nnz = (size_t)c->prep.nvar * c->prep.nvar * c->prep.bptr[c->prep.n_owned];
  if (val) RDC_HIP(c, hipMemcpyAsync(val, c->val.p, nnz * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (rhs) RDC_HIP(c, hipMemcpyAsync(rhs, c->rhs.p, (size_t)c->prep.n_owned * c->prep.nvar * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  return RDC_OK;
}

int rdc_csr_download_rows(rdc_ctx* c, int64_t node_begin, int64_t node_end, double* val, double* rhs, int async) {
  if (!c) return RDC_ERR_INVALID;
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  if (node_begin < 0 || node_end < node_begin || node_end > c->prep.n_owned) return fail(c, RDC_ERR_INVALID, "bad node range");
  int rc = set_device(c);
  if (rc) return rc;
  const int64_t nv = c->prep.nvar;
  const int64_t v0 = nv * nv * c->prep.bptr[(size_t)node_begin], v1 = nv * nv * c->prep.bptr[(size_t)node_end];
  if (val && v1 > v0)
    RDC_HIP(c, hipMemcpyAsync(val + v0, (const double*)c->val.p + v0, (size_t)(v1 - v0) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (rhs && node_end > node_begin)
    RDC_HIP(c, hipMemcpyAsync(rhs + node_begin * nv, (const double*)c->rhs.p + node_begin * nv,
                              (size_t)((node_end - node_begin) * nv) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (!async) RDC_HIP(c, hipStreamSynchronize(c->stream));
  return RDC_OK;
}

int rdc_csr_download_rows_async(rdc_ctx* c, int64_t node_begin, int64_t node_end, double* val, double* rhs, int* ticket) {
  if (!c || !ticket) return RDC_ERR_INVALID;
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  if (node_begin < 0 || node_end < node_begin || node_end > c->prep.n_owned) return fail(c, RDC_ERR_INVALID, "bad node range");
  int rc = set_device(c);
  if (rc) return rc;
  if (!c->copy_stream) RDC_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  if (!c->copy_fence) RDC_HIP(c, hipEventCreateWithFlags(&c->copy_fence, hipEventDisableTiming));
  const int t = c->next_ticket;
  c->next_ticket = (t + 1) % rdc_ctx::N_TICKETS;
  if (!c->ticket[t]) RDC_HIP(c, hipEventCreateWithFlags(&c->ticket[t], hipEventDisableTiming | hipEventBlockingSync));
  else RDC_HIP(c, hipEventSynchronize(c->ticket[t]));   // the slot's previous copy (16 calls ago) has certainly been consumed
  // the rows must be complete in what has been enqueued on the context's stream so far; work enqueued later (part 2) is not waited for
  RDC_HIP(c, hipEventRecord(c->copy_fence, c->stream));
  RDC_HIP(c, hipStreamWaitEvent(c->copy_stream, c->copy_fence, 0));
  const int64_t nv = c->prep.nvar;
  const int64_t v0 = nv * nv * c->prep.bptr[(size_t)node_begin], v1 = nv * nv * c->prep.bptr[(size_t)node_end];
  if (val && v1 > v0)
    RDC_HIP(c, hipMemcpyAsync(val + v0, (const double*)c->val.p + v0, (size_t)(v1 - v0) * sizeof(double), hipMemcpyDeviceToHost, c->copy_stream));
  if (rhs && node_end > node_begin)
    RDC_HIP(c, hipMemcpyAsync(rhs + node_begin * nv, (const double*)c->rhs.p + node_begin * nv,
                              (size_t)((node_end - node_begin) * nv) * sizeof(double), hipMemcpyDeviceToHost, c->copy_stream));
  RDC_HIP(c, hipEventRecord(c->ticket[t], c->copy_stream));
  *ticket = t;
  return RDC_OK;
}

int rdc_ticket_wait(rdc_ctx* c, int ticket) {
  if (!c || ticket < 0 || ticket >= rdc_ctx::N_TICKETS || !c->ticket[ticket]) return c ? fail(c, RDC_ERR_INVALID, "no such ticket") : RDC_ERR_INVALID;
  int rc = set_device(c);
  if (rc) return rc;
  RDC_HIP(c, hipEventSynchronize(c->ticket[ticket]));
  return RDC_OK;
}

int rdc_host_pin(rdc_ctx* c, void* p, size_t bytes) {
  if (!c || !p) return RDC_ERR_INVALID;
  int rc = set_device(c);
  if (rc) return rc;
  RDC_HIP(c, hipHostRegister(p, bytes, hipHostRegisterDefault));
  return RDC_OK;
}

int rdc_host_unpin(rdc_ctx* c, void* p) {
  if (!c || !p) return RDC_ERR_INVALID;
  RDC_HIP(c, hipHostUnregister(p));
  return RDC_OK;
}

int rdc_part1_nodes(const rdc_ctx* c, int64_t* n_nodes) {
  if (!c || !n_nodes) return RDC_ERR_INVALID;
  if (!c->have_mesh) return RDC_ERR_STATE;
  *n_nodes = 0;
  // after a part-1 call: what THAT call completed (the kernel path depends on the model, the parameter values and the
  // tuning options, and the paths split the rows differently)
  if (c->part1_nodes >= 0) { *n_nodes = c->part1_nodes; return RDC_OK; }
  // before any part-1 call: the prediction for the default path of the shipped-pattern PIHNA (element-visit clusters)
  if (c->opt_interior >= 0 && c->prep_ev.ok && (c->opt_kernel == 0 || c->opt_kernel == 7)) {
    *n_nodes = ev_part1_node_bound(c);
    return RDC_OK;
  }
  if (c->opt_interior < 0 || !c->prep.rg2_ok || c->prep.nen != 4 || c->prep.wg2.empty()) return RDC_OK;
  const int split = part1_workgroups(c);
  if (split > 0) *n_nodes = (int64_t)c->prep.wg2[(size_t)split - 1].n0 + c->prep.wg2[(size_t)split - 1].nnodes;
  return RDC_OK;
}

int rdc_clamp_nonnegative(rdc_ctx* c, int field) {
  if (!c) return RDC_ERR_INVALID;
  if (field < 0 || field >= RDC_FIELD_COUNT || !c->field[field].p) return fail(c, RDC_ERR_STATE, "field %d not set", field);
  int rc = set_device(c);
  if (rc) return rc;
  const int64_t n = c->field_count[field];
  if (n > 0) {
    int64_t grid = (n + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_clamp_nonnegative, dim3((unsigned)grid), dim3(256), 0, c->stream, (double*)c->field[field].p, n);
    RDC_HIP(c, hipGetLastError());
  }
  return RDC_OK;
}

int rdc_pihna_volume_integrals(rdc_ctx* c, const rdc_pihna_ranges* r, int64_t n_elem, double* out4) {
  if (!c) return RDC_ERR_INVALID;
  if (!r || !out4) return fail(c, RDC_ERR_INVALID, "null argument");
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  if (c->prep.nvar != 5) return fail(c, RDC_ERR_INVALID, "PIHNA volume integrals need nvar=5");
  if (!c->field[RDC_FIELD_OLD_SOLUTION].p) return fail(c, RDC_ERR_STATE, "solution field not set");
  if (n_elem < 0) n_elem = c->prep.n_elem;
  if (n_elem > c->prep.n_elem) return fail(c, RDC_ERR_INVALID, "n_elem exceeds the mesh");
  int rc = set_device(c);
  if (rc) return rc;
  int64_t grid = (n_elem + 255) / 256;
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  if ((rc = dev_alloc(c, c->wg_max, (size_t)grid * 4 * sizeof(double)))) return rc;
  const MeshDev m = mesh_view(c);
  if (c->prep.nen == 4)
    hipLaunchKernelGGL((k_pihna_volumes<4>), dim3((unsigned)grid), dim3(256), 0, c->stream, m, n_elem,
                       (const double*)c->field[RDC_FIELD_OLD_SOLUTION].p, *r, (double*)c->wg_max.p);
  else
    hipLaunchKernelGGL((k_pihna_volumes<8>), dim3((unsigned)grid), dim3(256), 0, c->stream, m, n_elem,
                       (const double*)c->field[RDC_FIELD_OLD_SOLUTION].p, *r, (double*)c->wg_max.p);
  RDC_HIP(c, hipGetLastError());
  std::vector<double> h((size_t)grid * 4);
  RDC_HIP(c, hipMemcpyAsync(h.data(), c->wg_max.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  for (int x = 0; x < 4; x++) {
    double s = 0.0;
    for (int64_t g = 0; g < grid; g++) s += h[(size_t)g * 4 + x];
    out4[x] = s;
  }
  return RDC_OK;
}

int rdc_ripf_volume_integrals(rdc_ctx* c, const rdc_ripf_ranges* r, int64_t n_elem, double* out2) {
  if (!c) return RDC_ERR_INVALID;
  if (!r || !out2) return fail(c, RDC_ERR_INVALID, "null argument");
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  if (c->prep.nvar != 3) return fail(c, RDC_ERR_INVALID, "RIPF volume integrals need nvar=3");
  if (!c->field[RDC_FIELD_OLD_SOLUTION].p) return fail(c, RDC_ERR_STATE, "solution field not set");
  if (n_elem < 0) n_elem = c->prep.n_elem;
  if (n_elem > c->prep.n_elem) return fail(c, RDC_ERR_INVALID, "n_elem exceeds the mesh");
  int rc = set_device(c);
  if (rc) return rc;
  int64_t grid = (n_elem + 255) / 256;
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  if ((rc = dev_alloc(c, c->wg_max, (size_t)grid * 2 * sizeof(double)))) return rc;
  const MeshDev m = mesh_view(c);
  if (c->prep.nen == 4)
    hipLaunchKernelGGL((k_ripf_volumes<4>), dim3((unsigned)grid), dim3(256), 0, c->stream, m, n_elem,
                       (const double*)c->field[RDC_FIELD_OLD_SOLUTION].p, *r, (double*)c->wg_max.p);
  else
    hipLaunchKernelGGL((k_ripf_volumes<8>), dim3((unsigned)grid), dim3(256), 0, c->stream, m, n_elem,
                       (const double*)c->field[RDC_FIELD_OLD_SOLUTION].p, *r, (double*)c->wg_max.p);
  RDC_HIP(c, hipGetLastError());
  std::vector<double> h((size_t)grid * 2);
  RDC_HIP(c, hipMemcpyAsync(h.data(), c->wg_max.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  for (int x = 0; x < 2; x++) {
    double s = 0.0;
    for (int64_t g = 0; g < grid; g++) s += h[(size_t)g * 2 + x];
    out2[x] = s;
  }
  return RDC_OK;
}

int rdc_adpm_parcellation_integrals(rdc_ctx* c, const rdc_adpm_ranges* r, const int32_t* elem_subdomain,
                                    const int32_t* ids, int32_t n_ids, int64_t n_elem, double* out,
                                    int64_t* last_elem) {
  if (!c) return RDC_ERR_INVALID;
  if (!r || !elem_subdomain || !ids || !out) return fail(c, RDC_ERR_INVALID, "null argument");
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  if (c->prep.nvar != 3) return fail(c, RDC_ERR_INVALID, "ADPM parcellation integrals need nvar=3");
  if (!c->field[RDC_FIELD_OLD_SOLUTION].p) return fail(c, RDC_ERR_STATE, "solution field not set");
  if (n_ids <= 0 || n_ids > 2048) return fail(c, RDC_ERR_INVALID, "n_ids must be in [1, 2048]");
  for (int32_t i = 1; i < n_ids; i++)
    if (ids[i] <= ids[i - 1]) return fail(c, RDC_ERR_INVALID, "parcellation ids must be strictly ascending");
  if (n_elem < 0) n_elem = c->prep.n_elem;
  if (n_elem > c->prep.n_elem) return fail(c, RDC_ERR_INVALID, "n_elem exceeds the mesh");
  int rc = set_device(c);
  if (rc) return rc;
  // region slot of every element; the last element of each region carries bit 30
  std::vector<int32_t> slot((size_t)(n_elem > 0 ? n_elem : 1), -1);
  std::vector<int64_t> last((size_t)n_ids, -1);
  for (int64_t e = 0; e < n_elem; e++) {
    const int32_t* it = std::lower_bound(ids, ids + n_ids, elem_subdomain[e]);
    if (it == ids + n_ids || *it != elem_subdomain[e]) continue;
    slot[(size_t)e] = (int32_t)(it - ids);
    last[(size_t)(it - ids)] = e;
  }
  for (int32_t i = 0; i < n_ids; i++)
    if (last[(size_t)i] >= 0) slot[(size_t)last[(size_t)i]] |= 0x40000000;
  int64_t grid = (n_elem + 255) / 256;
  if (grid > 256) grid = 256;
  if (grid < 1) grid = 1;
  const size_t n_part = (size_t)grid * 2 * n_ids;
  if ((rc = dev_alloc(c, c->wg_max, (n_part + 2 * (size_t)n_ids) * sizeof(double)))) return rc;
  if ((rc = dev_upload(c, c->adpm_slot, slot))) return rc;
  double* part = (double*)c->wg_max.p;
  double* conc = part + n_part;
  RDC_HIP(c, hipMemsetAsync(conc, 0, 2 * (size_t)n_ids * sizeof(double), c->stream));
  const MeshDev m = mesh_view(c);
  const size_t lds = 2 * (size_t)n_ids * sizeof(double);
  if (c->prep.nen == 4)
    hipLaunchKernelGGL((k_adpm_parcellation<4>), dim3((unsigned)grid), dim3(256), lds, c->stream, m, n_elem,
                       (const double*)c->field[RDC_FIELD_OLD_SOLUTION].p, *r, (const int32_t*)c->adpm_slot.p, (int)n_ids, part, conc);
  else
    hipLaunchKernelGGL((k_adpm_parcellation<8>), dim3((unsigned)grid), dim3(256), lds, c->stream, m, n_elem,
                       (const double*)c->field[RDC_FIELD_OLD_SOLUTION].p, *r, (const int32_t*)c->adpm_slot.p, (int)n_ids, part, conc);
  RDC_HIP(c, hipGetLastError());
  std::vector<double> h(n_part + 2 * (size_t)n_ids);
  RDC_HIP(c, hipMemcpyAsync(h.data(), c->wg_max.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  for (int32_t i = 0; i < n_ids; i++) {
    double sa = 0.0, st = 0.0;
    for (int64_t g = 0; g < grid; g++) {
      sa += h[(size_t)g * 2 * n_ids + 2 * i];
      st += h[(size_t)g * 2 * n_ids + 2 * i + 1];
    }
    out[4 * i] = h[n_part + 2 * i];
    out[4 * i + 1] = h[n_part + 2 * i + 1];
    out[4 * i + 2] = sa;
    out[4 * i + 3] = st;
    if (last_elem) last_elem[i] = last[(size_t)i];
  }
  return RDC_OK;
}

int rdc_solid_post_process(rdc_ctx* c, const rdc_solid_params* p, double* pressure, double* von_mises,
                           double* fibre_current) {
  if (!c) return RDC_ERR_INVALID;
  if (!p) return fail(c, RDC_ERR_INVALID, "null parameter struct");
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  if (c->prep.nvar != 3) return fail(c, RDC_ERR_INVALID, "solid system needs nvar=3");
  if (!c->field[RDC_FIELD_UNDEFORMED_XYZ].p) return fail(c, RDC_ERR_STATE, "undeformed coordinates not set");
  if (!c->field[RDC_FIELD_ELEM_FIBRE].p) return fail(c, RDC_ERR_STATE, "fibre field not set");
  if (c->n_materials <= 0) return fail(c, RDC_ERR_STATE, "materials not set");
  int rc = set_device(c);
  if (rc) return rc;
  const size_t ne = (size_t)c->prep.n_elem;
  if ((rc = dev_alloc(c, c->solid_post, ne * 5 * sizeof(double)))) return rc;
  SolidArgs a{};
  a.m = mesh_view(c);
  a.nen = c->prep.nen;
  a.Xu = (const double*)c->field[RDC_FIELD_UNDEFORMED_XYZ].p;
  a.fibre = (const double*)c->field[RDC_FIELD_ELEM_FIBRE].p;
  a.elem_material = (const int32_t*)c->elem_material.p;
  a.materials = (const rdc_solid_material*)c->materials.p;
  a.params = *p;
  a.stream = c->stream;
  hipError_t e = launch_solid_post(a, (double*)c->solid_post.p);
  if (e != hipSuccess) return fail(c, RDC_ERR_HIP, "solid post-process launch failed: %s", hipGetErrorString(e));
  std::vector<double> h(ne * 5);
  RDC_HIP(c, hipMemcpyAsync(h.data(), c->solid_post.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  for (size_t e2 = 0; e2 < ne; e2++) {
    if (pressure) pressure[e2] = h[5 * e2];
    if (von_mises) von_mises[e2] = h[5 * e2 + 1];
    if (fibre_current) { fibre_current[3 * e2] = h[5 * e2 + 2]; fibre_current[3 * e2 + 1] = h[5 * e2 + 3]; fibre_current[3 * e2 + 2] = h[5 * e2 + 4]; }
  }
  return RDC_OK;
}

int rdc_ripf_check_solution(rdc_ctx* c, const rdc_ripf_check_params* p, double* rt_total_max) {
  if (!c) return RDC_ERR_INVALID;
  if (!p) return fail(c, RDC_ERR_INVALID, "null parameter struct");
  if (!c->have_mesh) return fail(c, RDC_ERR_STATE, "no mesh uploaded");
  if (c->prep.nvar != 3) return fail(c, RDC_ERR_INVALID, "RIPF check_solution needs nvar=3");
  if (!(p->time_step > 0.0)) return fail(c, RDC_ERR_INVALID, "time_step must be positive");
  if (p->RT_broad_fractions < 0 || p->RT_focus_fractions < 0) return fail(c, RDC_ERR_INVALID, "negative fraction count");
  const int need[] = {RDC_FIELD_OLD_SOLUTION, RDC_FIELD_PREV_SOLUTION, RDC_FIELD_RT_DOSE};
  const int64_t n = c->prep.n_node;
  for (int f : need)
    if (!c->field[f].p || c->field_count[f] != 3 * n) return fail(c, RDC_ERR_STATE, "field %d not set", f);
  int rc = set_device(c);
  if (rc) return rc;
  for (int f : {RDC_FIELD_TIME_DERIV, RDC_FIELD_AUX_NODAL}) {
    if (c->field[f].p && c->field_count[f] == 3 * n) continue;
    if (c->field[f].p && !c->field[f].owned) return fail(c, RDC_ERR_STATE, "bound field %d has the wrong size", f);
    if ((rc = dev_alloc(c, c->field[f], (size_t)(3 * n) * sizeof(double)))) return rc;
    c->field_count[f] = 3 * n;
  }
  int64_t grid = (n + 255) / 256;
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  if ((rc = dev_alloc(c, c->wg_max, (size_t)grid * sizeof(double)))) return rc;
  hipLaunchKernelGGL(k_ripf_check, dim3((unsigned)grid), dim3(256), 0, c->stream, n, 1.0 / p->time_step, p->HU_min, p->HU_max,
                     (double)p->RT_broad_fractions, (double)p->RT_focus_fractions, (int)p->day,
                     (double*)c->field[RDC_FIELD_OLD_SOLUTION].p, (double*)c->field[RDC_FIELD_PREV_SOLUTION].p,
                     (double*)c->field[RDC_FIELD_TIME_DERIV].p, (double*)c->field[RDC_FIELD_RT_DOSE].p,
                     (double*)c->field[RDC_FIELD_AUX_NODAL].p, (double*)c->wg_max.p);
  RDC_HIP(c, hipGetLastError());
  std::vector<double> h((size_t)grid);
  RDC_HIP(c, hipMemcpyAsync(h.data(), c->wg_max.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  double mx = -1.0;
  for (double v : h) mx = v > mx ? v : mx;
  if (rt_total_max) *rt_total_max = mx;
  return RDC_OK;
}

int rdc_debug_stamps(rdc_ctx* c, long long* host_out, int64_t capacity, int64_t* n_written) {
  if (!c) return RDC_ERR_INVALID;
  if (!c->have_mesh || !c->prep.rg2_ok) return fail(c, RDC_ERR_STATE, "no row-gather work lists");
  int rc = set_device(c);
  if (rc) return rc;
  // "kernel" = 7 with "ablate" = 4: the stamped build of the element-visit kernel, [cluster][wave][12] (rdc_tet4_ev.hip, tools/ev_timeline.py)
  const bool ev_tl = c->opt_kernel == 7 && c->opt_ablate == 4 && c->prep_ev.ok;
  const int64_t n = ev_tl ? (int64_t)c->prep_ev.desc.size() * 4 * 12 : (int64_t)c->prep.wg2.size() * 4 * 9;
  if (!host_out) {  // arm: the next PIHNA (shipped-parameter) assembly runs the stamped diagnostic kernel
    if ((rc = dev_alloc(c, c->stamps, (size_t)n * sizeof(long long)))) return rc;
    RDC_HIP(c, hipMemsetAsync(c->stamps.p, 0, (size_t)n * sizeof(long long), c->stream));
    if (n_written) *n_written = n;
    return RDC_OK;
  }
  if (!c->stamps.p) return fail(c, RDC_ERR_STATE, "stamps not armed");
  const int64_t m = n < capacity ? n : capacity;
  RDC_HIP(c, hipMemcpyAsync(host_out, c->stamps.p, (size_t)m * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
  RDC_HIP(c, hipStreamSynchronize(c->stream));
  if (n_written) *n_written = m;
  dev_free(c, c->stamps);
  return RDC_OK;
}

int rdc_timing_enable(rdc_ctx* c, int on) {
  if (!c) return RDC_ERR_INVALID;
  c->timing = on != 0;
  c->ev_used = 0;
  return RDC_OK;
}

int rdc_timing_last_ms(rdc_ctx* c, float* ms) {
  if (!c || !ms) return RDC_ERR_INVALID;
  if (c->ev_used < 2) return fail(c, RDC_ERR_STATE, "no timed assemble call recorded");
  int rc = set_device(c);
  if (rc) return rc;
  RDC_HIP(c, hipEventSynchronize(c->ev[c->ev_used - 1]));
  RDC_HIP(c, hipEventElapsedTime(ms, c->ev[c->ev_used - 2], c->ev[c->ev_used - 1]));
  return RDC_OK;
}

int rdc_timing_sum_ms(rdc_ctx* c, float* total_ms, int* n_calls) {
  if (!c || !total_ms || !n_calls) return RDC_ERR_INVALID;
  int rc = set_device(c);
  if (rc) return rc;
  float total = 0.0f;
  for (size_t x = 0; x + 1 < c->ev_used; x += 2) {
    float ms = 0.0f;
    RDC_HIP(c, hipEventSynchronize(c->ev[x + 1]));
    RDC_HIP(c, hipEventElapsedTime(&ms, c->ev[x], c->ev[x + 1]));
    total += ms;
  }
  *total_ms = total;
  *n_calls = (int)(c->ev_used / 2);
  c->ev_used = 0;
  return RDC_OK;
}

int rdc_timing_samples_ms(rdc_ctx* c, float* out, int capacity, int* n_calls) {
  if (!c || !out || !n_calls || capacity < 0) return RDC_ERR_INVALID;
  int rc = set_device(c);
  if (rc) return rc;
  int n = 0;
  for (size_t x = 0; x + 1 < c->ev_used; x += 2, n++) {
    float ms = 0.0f;
    RDC_HIP(c, hipEventSynchronize(c->ev[x + 1]));
    RDC_HIP(c, hipEventElapsedTime(&ms, c->ev[x], c->ev[x + 1]));
    if (n < capacity) out[n] = ms;
  }
  *n_calls = n;
  c->ev_used = 0;
  return RDC_OK;
}

}  // extern "C"
