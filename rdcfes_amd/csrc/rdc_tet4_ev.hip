// rdc_tet4_ev.hip — element-visit / moment-accumulation kernel of the PIHNA TET4 assembly (shipped parameter pattern).
// See rdc_tet4_ev.h for the formulation and rdc_prep_ev.cpp for the work lists.
//
// One workgroup = one CLUSTER of <= 16 owned nodes (not necessarily consecutive ids) and the <= 256 elements touching it.
//   phase 0  zero the moment slice, load the lists, LDS-DMA the node records of the cluster's closure      (as k_tet4_rg5)
//   phase 1  one thread per element visit: pihna_visit() -> ds_add_f64 of the moments of the rows it owns
//   phase 2  one thread per node block: 16 moments (+ the symmetric ones of its mirror block) -> 25 matrix entries (pihna_expand), written to an LDS image of the
//            CSR rows of the cluster's nodes; rhs entries straight to memory
//   phase 3  the image leaves with 16-byte non-temporal stores, one contiguous CSR segment per node
// Every CSR value is written exactly once; no global atomics, no colours.  Sums are order-dependent in the last bits.
#include "rdc_internal.h"
#include <type_traits>
#include "rdc_tet4_ev.h"

namespace rdc {

template <class M> struct RecEv {
  static constexpr int RAW = 3 + M::NV + M::NAUX;
  static constexpr int N = (RAW + 1) & ~1;
};

// ABL (diagnostic builds): 1 = plain LDS stores instead of atomics, 2 = no LDS accumulation traffic, 3 = no compute phase (results WRONG);
// 4 = the real kernel with s_memtime stamps per wave and phase (results right; tools/ev_timeline.py)
template <int ABL_>
struct EvSink {
  static constexpr int ABL = ABL_ == 4 ? 0 : ABL_;
  double* p[4][4];   // LDS address of moment 0 of block (node i, node j)
  double* pr[4];     // LDS address of rhs entry 0 of node i
  double sum = 0.0;
  __device__ __forceinline__ void mom(int m, int i, int j, double v) {
    if (ABL == 0) __hip_atomic_fetch_add(p[i][j] + m * ev::NBP, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (ABL == 1) p[i][j][m * ev::NBP] = v;
    else sum += v;
  }
  __device__ __forceinline__ void rhs(int a, int i, double v) {
    if (ABL == 0) __hip_atomic_fetch_add(pr[i] + a * ev::MAXN, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (ABL == 1) pr[i][a * ev::MAXN] = v;
    else sum += v;
  }
};


// GEN: every term on (22 moments, rdc_tet4_ev.h); otherwise the shipped parameter pattern (16)
template <int EXP_MODE, int MINW, int ABL = 0, bool GEN = false>
__global__ void __launch_bounds__(256, MINW)
k_tet4_ev(const HostPrepEv::Desc* __restrict__ desc, const uint32_t* __restrict__ nlist, const uint32_t* __restrict__ vloc,
          const uint32_t* __restrict__ vslot, const HostPrepEv::Node* __restrict__ ntab, const uint8_t* __restrict__ bpart,
          const uint32_t* __restrict__ wg_perm, const PihnaK k, const double* __restrict__ rec, double* __restrict__ val,
          double* __restrict__ rhs, const int nls, const int wg_begin, const int xcd_n, const int stagger,
          long long* __restrict__ stamps, const int bg_skip) {
  constexpr int BLOCK = 256, NP = 4;   // PIHNA node record: 8 doubles = 4 pieces of 16 bytes
  constexpr bool TL = ABL == 4;
  long long ts[11];
#define RDC_TS(x) if (TL) ts[x] = __builtin_amdgcn_s_memtime()
  RDC_TS(0);
  constexpr int NM = GEN ? ev::NMG : ev::NM, NBP = ev::NBP, MAXN = ev::MAXN;
  extern __shared__ __attribute__((aligned(16))) double lds[];   // [M: NM x NBP | R: 5 x MAXN | records: NP x nls x 16 B]
  __shared__ HostPrepEv::Node snode[MAXN];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  // XCD-aware order: consecutive workgroup ids go to the 8 XCDs in turn (each with its own 4 MB L2), consecutive CLUSTERS are
  // neighbours in the mesh (they were grown along a front) and share most of their closure nodes -- so XCD x takes the
  // x-th eighth of the launch's clusters and walks through it in order: a cluster's node records and its neighbours' are then
  // fetched by one L2 instead of by all eight (xcd_n = clusters of this launch, 0 = off)
  // Diagnostic knob ("stagger" = n): the workgroups of the first wave of the launch (three per CU start together) begin 0, n or 2 n
  // x 1024 cycles late, by the parity of their LDS allocation (the three co-resident workgroups of a CU get different
  // addresses), so that their load / compute / store phases do not run in lockstep
  if (stagger > 0 && blockIdx.x < 1024u) {
    const unsigned lds_base = __builtin_amdgcn_s_getreg((7 << 11) | 6);   // HW_REG_LDS_ALLOC.LDS_BASE (bits 7:0): 0 for the CU's first workgroup
    const uint32_t slot = lds_base == 0 ? 0u : (lds_base < 144u ? 1u : 2u);
    for (uint32_t i = 0; i < slot * (uint32_t)stagger; i++) __builtin_amdgcn_s_sleep(16);   // 16 x 64 cycles each
  }
  int wl = (int)blockIdx.x;
  if (xcd_n > 0) {
    const int q = xcd_n >> 3, rem = xcd_n & 7, x = wl & 7;
    wl = x * q + (x < rem ? x : rem) + (wl >> 3);
  }
  int w = wl + wg_begin;
  if (wg_perm) w = (int)wg_perm[w];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double* const R = lds + NM * NBP;
  double* const recs = R + 5 * MAXN;
  // phase 0: zero [M | R], list loads, LDS-DMA of the node records.  A store moves its address and data registers to the
  // LDS at 2 cycles per source dword and wave instruction (MI355X_MICROARCH.md, LDS): ds_write_addtid_b32 has no address
  // register (address = M0 + offset + 4 * lane), so zeroing the 32 KB slice costs 2 cycles per 256 bytes instead of 13 per
  // 1024 with 16-byte stores.  Wave wv clears its quarter of the slice; M0 is restored (the LDS-DMA below sets it too).
  {
    static_assert((NM * NBP * 8) % (4 * 1024) == 0, "zeroing: 4 waves x (NM / 2) groups of four addtid stores of 256 bytes");
    constexpr int PER_WAVE = NM * NBP * 8 / 4;   // bytes
    const uint32_t zbase = (uint32_t)(uintptr_t)lds + (uint32_t)__builtin_amdgcn_readfirstlane(wv) * (uint32_t)PER_WAVE;
    const uint32_t zero = 0u;
    uint32_t m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0" : "=&s"(m0_saved) : "s"(zbase) : "memory");
#pragma unroll
    for (int o = 0; o < PER_WAVE; o += 1024)
      asm volatile("ds_write_addtid_b32 %0 offset:%1\n\tds_write_addtid_b32 %0 offset:%1+256\n\tds_write_addtid_b32 %0 offset:%1+512\n\tds_write_addtid_b32 %0 offset:%1+768"
                   :: "v"(zero), "n"(o) : "memory");
    asm volatile("s_mov_b32 m0, %0" :: "s"(m0_saved) : "memory");
    if (tid < 5 * MAXN) lds[NM * NBP + tid] = 0.0;
  }
  const int rounds = nls >> 6;
  uint32_t nid = 0;
  if (wv < rounds) nid = nlist[(size_t)w * nls + wv * 64 + lane];
  const uint32_t pl = vloc[(size_t)w * BLOCK + tid];
  const uint2 sl = reinterpret_cast<const uint2*>(vslot)[(size_t)w * BLOCK + tid];
  const HostPrepEv::Desc d = desc[w];
  const int mirror = (int)bpart[(size_t)w * NBP + tid];   // the block whose symmetric moments are added to this one's (tid: none)
  if (tid < MAXN) snode[tid] = ntab[(size_t)w * MAXN + tid];
  if (wv < rounds) {
    const char* src = reinterpret_cast<const char*>(rec) + (size_t)nid * (NP * 16);
#pragma unroll
    for (int p = 0; p < NP; p++)
      __builtin_amdgcn_global_load_lds((glb_ptr)(src + p * 16), (lds_ptr)(recs + (p * nls + wv * 64) * 2), 16, 0, 0);
  }
  RDC_TS(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RDC_TS(2);
  __syncthreads();
  RDC_TS(3);
  // phase 1: element visits
  if (pl != 0xFFFFFFFFu) {
    double X[4][3], U[4][5];
    int li[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      li[j] = (pl >> (8 * j)) & 0xFF;
      double rr[2 * NP];
#pragma unroll
      for (int p = 0; p < NP; p++) {
        const double2 v2 = reinterpret_cast<const double2*>(recs)[p * nls + li[j]];
        rr[2 * p] = v2.x; rr[2 * p + 1] = v2.y;
      }
      X[j][0] = rr[0]; X[j][1] = rr[1]; X[j][2] = rr[2];
#pragma unroll
      for (int v = 0; v < 5; v++) U[j][v] = rr[3 + v];
    }
    const int nown = (int)d.nown;
    // the owned vertices come first: r = number of list positions below nown
    const int r = (li[0] < nown) + (li[1] < nown) + (li[2] < nown) + (li[3] < nown);
    EvSink<ABL> sink;
    const uint32_t sw[4] = {sl.x & 0xFFFFu, sl.x >> 16, sl.y & 0xFFFFu, sl.y >> 16};   // four 4-bit column slots per row
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int a = (i < r) ? li[i] : 0;   // owned vertices come first in the node list: list position == cluster index
      sink.pr[i] = R + a;
#pragma unroll
      for (int j = 0; j < 4; j++) sink.p[i][j] = lds + a + MAXN * (int)((sw[i] >> (4 * j)) & 0xF);   // block (a, slot): slot * 16 + a
    }
    // background state (n = c = h = a = 0, v > 0) at every vertex of every visit of this wave: a scalar flag (rdc_tet4_ev.h, bg)
    const bool bg = bg_skip && __builtin_amdgcn_ballot_w64(!ev::pihna_background(U)) == 0;
    if (ABL != 3) ev::pihna_visit<EXP_MODE, EvSink<ABL>, true, GEN>(k, X, U, r, sink, bg);   // ABL 3: no compute phase at all (data movement only)
    if (ABL == 2 && sink.sum == 1.2345e300) rhs[0] = sink.sum;  // keeps the arithmetic alive
  }
  if (TL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS atomics have been executed
  RDC_TS(4);
  __syncthreads();
  RDC_TS(5);
  // phase 2: node block tid = slot * 16 + node: moments -> entries
  double e[NM];
  const int bn = tid & (MAXN - 1), bs = tid >> 4;
  const bool has = bn < (int)d.nown && bs < (int)snode[bn < (int)d.nown ? bn : 0].len;
  if (has) {
#pragma unroll
    for (int m = 0; m < NM; m++) e[m] = lds[m * NBP + tid];
    if (mirror != tid) {   // rdc_tet4_ev.h, MIRROR
#pragma unroll
      for (int m = 0; m < NM; m++)
        if (GEN ? ev::symmetric_moment_gen(m) : ev::symmetric_moment(m)) e[m] += lds[m * NBP + mirror];
    }
  }
  if (tid < (int)d.nown * 5) {   // rhs: R[a][node] -> rhs[node * 5 + a]
    const int n = tid / 5, a = tid - n * 5;
    rhs[(size_t)snode[n].node * 5 + a] = R[a * MAXN + n];
  }
  RDC_TS(6);
  __syncthreads();   // every moment has been read: the image may overwrite the slice
  if (has) {
    double o[25];
    ev::pihna_expand(k, e, o);
    const int n = bn, s = bs;
    const int len5 = 5 * (int)snode[n].len;
    double* dst = lds + snode[n].obase + 5 * s;
#pragma unroll
    for (int a = 0; a < 5; a++)
#pragma unroll
      for (int b = 0; b < 5; b++) dst[a * len5 + b] = o[a * 5 + b];
  }
  RDC_TS(7);
  __syncthreads();
  RDC_TS(8);
  // phase 3: one contiguous CSR segment per node; the image has the 16-byte phase of its segment in memory
  for (int n = wv; n < (int)d.nown; n += 4) {
    const HostPrepEv::Node nd = snode[n];
    const int cnt = 25 * (int)nd.len;
    double* out = val + (size_t)25 * nd.bptr;            // out[x] <-> img[x]
    const double* img = lds + nd.obase;
    const int sh = (int)(nd.obase & 1);                  // == (25 * bptr) & 1 by construction
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    const int npair = (cnt - sh) >> 1;
    const v2d_t* src = reinterpret_cast<const v2d_t*>(img + sh);
    v2d_t* dstg = reinterpret_cast<v2d_t*>(out + sh);
    for (int x = lane; x < npair; x += 64) __builtin_nontemporal_store(src[x], dstg + x);
    if (sh && lane == 0) __builtin_nontemporal_store(img[0], out);
    if (((cnt - sh) & 1) && lane == 1) __builtin_nontemporal_store(img[cnt - 1], out + cnt - 1);
  }
  if (TL) {
    RDC_TS(9);
    if (stagger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wave's stores have been acknowledged ("stagger" != 0: not waited for)
    RDC_TS(10);
    if (lane == 0 && stamps) {
      long long* o = stamps + ((int64_t)blockIdx.x * 4 + wv) * 12;
#pragma unroll
      for (int x = 0; x < 11; x++) o[x] = ts[x];
      // HW_REG_HW_ID (4): wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13;  HW_REG_XCC_ID (20): xcc 3:0
      o[11] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
  }
#undef RDC_TS
}


// ---- pipelined resident form ("ev_resident" = 1, experimental) -------------------------------------------------------------------
// The stamped build of k_tet4_ev (tools/ev_timeline.py, K(119)) puts 23 % of a workgroup's life into phase 0 -- list loads, the
// dependent record DMA, the barrier behind the slowest wave's loads -- during which its LDS and registers do nothing; the
// stores are not what it waits for at its end (1 %).  Here three workgroups per CU stay resident (walking over the clusters
// blockIdx + i * gridDim; a first form that only loaded the next node list ahead, k_tet4_evl, measured 2.18 vs 2.04 ms and was removed) and fetch
// cluster i + 1 WHILE cluster i is expanded and copied out: behind the barrier that ends the visits of cluster i every wave
// issues the LDS-DMA of the records and lists of cluster i + 1 (node ids DMA'd one cluster earlier) and of the node ids of
// cluster i + 2.  They land during the moment reads, the expansion and the copy-out; the one s_waitcnt vmcnt(0) at the top of a
// cluster then only waits for the copy-out stores just issued.  Everything is fetched by LDS-DMA in asm the compiler does not
// see (no result registers, no compiler-placed vmcnt waits), and nothing between the fetch and the top of the next cluster may
// touch scratch (a scratch reload waits for every earlier vector-memory operation of its wave).
// LDS: the record and list landing areas may not be covered by the CSR image, so the image is built and sent in two halves
// (nodes 0-7 by waves 0-1, nodes 8-15 by waves 2-3: at most 8 x 16 x 25 doubles, inside the 32 KB moment slice); 46 KB per workgroup.
namespace {
// LDS-DMA of 16 bytes per lane (lane l lands at lds_base + 16 l), issued where the compiler cannot see it: it guards the next LDS
// write after a VISIBLE LDS-DMA with s_waitcnt vmcnt(0) (possible alias), which would stall the wave behind every fetch
__device__ __forceinline__ void evl_dma16(const void* g, const void* lds_base) {
  const uint32_t b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(b) : "memory", "m0");
}
}
namespace {
struct EvqLists {   // landing area of a cluster's lists (bytes): visit positions, column slots, then mirror blocks | node table | descriptor; node ids x 2
  static constexpr int PL = 0, SL = 1024, MISC = SL + 2048, MIRROR = MISC, SNODE = MISC + 256, DESC = MISC + 512, NL = MISC + 544;
  static constexpr int bytes(int nls) { return NL + 2 * nls * 4; }
};
}
// GEN: the 22-moment instantiation (any parameter values), at the register budget of two workgroups per CU as in k_tet4_ev
template <int EXP_MODE, bool TL, bool GEN = false>
__global__ void __launch_bounds__(256, GEN ? 2 : 3)
k_tet4_evq(const HostPrepEv::Desc* __restrict__ desc, const uint32_t* __restrict__ nlist, const uint32_t* __restrict__ vloc,
           const uint32_t* __restrict__ vslot, const HostPrepEv::Node* __restrict__ ntab, const uint8_t* __restrict__ bpart,
           const PihnaK k, const double* __restrict__ rec, double* __restrict__ val, double* __restrict__ rhs, const int nls,
           const int wg_begin, const int wg_count, long long* __restrict__ stamps, const int bg_skip, int* __restrict__ ticket,
           const uint32_t* __restrict__ wg_perm) {
  constexpr int BLOCK = 256, NP = 4;
  constexpr int NM = GEN ? ev::NMG : ev::NM, NBP = ev::NBP, MAXN = ev::MAXN;
  extern __shared__ __attribute__((aligned(16))) double lds[];   // [M: NM x NBP, later the image halves | R: 5 x MAXN | records: NP x nls x 16 B | lists]
  __shared__ HostPrepEv::Node snode[MAXN];
  __shared__ uint8_t smirror[NBP];
  __shared__ int s_nown;
  __shared__ int s_tk[2];
  double* const R = lds + NM * NBP;
  double* const recs = R + 5 * MAXN;
  char* const lists = reinterpret_cast<char*>(recs + (size_t)NP * nls * 2);
  const int rounds = nls >> 6;
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // Clusters are handed out dynamically: the first three of a workgroup are blockIdx + {0, 1, 2} gridDim, every later one comes from a global counter
  // (`ticket`, zero at launch), requested THREE clusters ahead -- the node ids of cluster i + 2 are fetched while cluster i is
  // expanded -- by lane 0 of wave 0 and passed on through LDS at the barrier at the top of the next cluster.  c0 = this cluster,
  // c1, c2 = the next two (indices into the launch's range; >= wg_count: none).
  constexpr int NONE = 0x7fffffff;
  int c0 = (int)blockIdx.x, c1 = c0 + (int)gridDim.x, c2 = c0 + 2 * (int)gridDim.x;   // the first three of a workgroup are static (no ticket round trip at its start)
  // position in the launch's range -> cluster (two-part assembly: the range is a piece of a permuted cluster order).  Uniform
  // addresses: scalar loads, issued where the position becomes known and used a phase later
  auto cluster_of = [&](const int c) -> int { return c < wg_count ? (wg_perm ? (int)wg_perm[wg_begin + c] : wg_begin + c) : -1; };
  int w0c = cluster_of(c0), w1c = cluster_of(c1), w2c = cluster_of(c2);
  long long ts[8];
#define RDC_TS(x) if (TL) ts[x] = __builtin_amdgcn_s_memtime()
  // everything cluster `w` needs (node ids in `nid`), and the node ids of cluster `w2` into id buffer `nb2`, by LDS-DMA
  auto fetch = [&](const int w, const uint32_t nid, const int w2, const int nb2, const int lane) {
    if (wv < rounds) {
      const char* src = reinterpret_cast<const char*>(rec) + (size_t)nid * (NP * 16);
#pragma unroll
      for (int p = 0; p < NP; p++) evl_dma16(src + p * 16, recs + (p * nls + wv * 64) * 2);
    }
    if (wv == 0) evl_dma16(reinterpret_cast<const char*>(vloc + (size_t)w * BLOCK) + lane * 16, lists + EvqLists::PL);
    else if (wv < 3) evl_dma16(reinterpret_cast<const char*>(vslot + (size_t)w * BLOCK * 2) + (wv - 1) * 1024 + lane * 16, lists + EvqLists::SL + (wv - 1) * 1024);
    else {
      if (lane < 34) {
        const char* src = lane < 16 ? reinterpret_cast<const char*>(bpart + (size_t)w * NBP) + lane * 16
                        : lane < 32 ? reinterpret_cast<const char*>(ntab + (size_t)w * MAXN) + (lane - 16) * 16
                                    : reinterpret_cast<const char*>(desc + w) + (lane - 32) * 16;
        evl_dma16(src, lists + EvqLists::MISC);
      }
      if (w2 >= 0 && lane < (nls >> 2)) evl_dma16(reinterpret_cast<const char*>(nlist + (size_t)w2 * nls) + lane * 16, lists + EvqLists::NL + nb2 * nls * 4);
    }
  };
  {
    uint32_t nid0 = 0;
    if (wv < rounds) nid0 = nlist[(size_t)w0c * nls + threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nid0)::"memory");
    fetch(w0c, nid0, w1c, 1, (int)(threadIdx.x & 63));
  }
  int tk = NONE;          // lane 0 of wave 0: the ticket requested while the previous cluster was expanded
  bool pending = false;
#pragma unroll 1
  for (int it = 0;; it++) {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));   // recomputed every iteration: never hoisted out of the loop and spilled
    const int w = w0c;
    RDC_TS(0);
    {   // zero [M | R] (ds_write_addtid_b32: k_tet4_ev); the image of the previous cluster has been read (barrier at its end)
      constexpr int PER_WAVE = NM * NBP * 8 / 4;   // bytes
      const uint32_t zbase = (uint32_t)(uintptr_t)lds + (uint32_t)wv * (uint32_t)PER_WAVE;
      const uint32_t zero = 0u;
      uint32_t m0_saved;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0" : "=&s"(m0_saved) : "s"(zbase) : "memory");
#pragma unroll
      for (int o = 0; o < PER_WAVE; o += 1024)
        asm volatile("ds_write_addtid_b32 %0 offset:%1\n\tds_write_addtid_b32 %0 offset:%1+256\n\tds_write_addtid_b32 %0 offset:%1+512\n\tds_write_addtid_b32 %0 offset:%1+768"
                     :: "v"(zero), "n"(o) : "memory");
      asm volatile("s_mov_b32 m0, %0" :: "s"(m0_saved) : "memory");
      if (wv * 64 + lane < 5 * MAXN) lds[NM * NBP + wv * 64 + lane] = 0.0;
    }
    // the one wait of the cluster: what was fetched for it while the previous one was expanded -- and that one's stores
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(tk)::"memory");
    if (pending && wv == 0 && lane == 0) s_tk[0] = tk == NONE ? NONE : 3 * (int)gridDim.x + tk;
    RDC_TS(1);
    lds_barrier();
    RDC_TS(2);
    if (pending) { c2 = __builtin_amdgcn_readfirstlane(s_tk[0]); w2c = cluster_of(c2); }
    if (c0 >= wg_count) break;   // uniform
    {
      const int tid = wv * 64 + lane;
      const uint32_t pl = *reinterpret_cast<const uint32_t*>(lists + EvqLists::PL + tid * 4);
      const uint2 sl = *reinterpret_cast<const uint2*>(lists + EvqLists::SL + tid * 8);
      const int nown = (int)reinterpret_cast<const HostPrepEv::Desc*>(lists + EvqLists::DESC)->nown;
      // what the expansion needs from the lists is parked where the next fetch does not land
      {
        const int blk = (((wv & 1) << 3) | (lane >> 3)) * MAXN + (((wv >> 1) << 3) | (lane & 7));
        smirror[blk] = *reinterpret_cast<const uint8_t*>(lists + EvqLists::MIRROR + blk);
        if (tid < MAXN) snode[tid] = *reinterpret_cast<const HostPrepEv::Node*>(lists + EvqLists::SNODE + tid * 16);
        if (tid == 0) s_nown = nown;
      }
      // ---- element visits (phase 1 of k_tet4_ev)
      if (pl != 0xFFFFFFFFu) {
        double X[4][3], U[4][5];
        int li[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          li[j] = (pl >> (8 * j)) & 0xFF;
          double rr[2 * NP];
#pragma unroll
          for (int p = 0; p < NP; p++) {
            const double2 v2 = reinterpret_cast<const double2*>(recs)[p * nls + li[j]];
            rr[2 * p] = v2.x; rr[2 * p + 1] = v2.y;
          }
          X[j][0] = rr[0]; X[j][1] = rr[1]; X[j][2] = rr[2];
#pragma unroll
          for (int v = 0; v < 5; v++) U[j][v] = rr[3 + v];
        }
        const int r = (li[0] < nown) + (li[1] < nown) + (li[2] < nown) + (li[3] < nown);
        EvSink<0> sink;
        const uint32_t sw[4] = {sl.x & 0xFFFFu, sl.x >> 16, sl.y & 0xFFFFu, sl.y >> 16};
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int a = (i < r) ? li[i] : 0;
          sink.pr[i] = R + a;
#pragma unroll
          for (int j = 0; j < 4; j++) sink.p[i][j] = lds + a + MAXN * (int)((sw[i] >> (4 * j)) & 0xF);
        }
        const bool bg = bg_skip && __builtin_amdgcn_ballot_w64(!ev::pihna_background(U)) == 0;   // k_tet4_ev
        ev::pihna_visit<EXP_MODE, EvSink<0>, true, GEN>(k, X, U, r, sink, bg);
      }
    }
    RDC_TS(3);
    lds_barrier();   // every visit has read its lists and records and added its moments
    RDC_TS(4);
    // ---- from here to the top of the next cluster: NO scratch (the lane index is taken from the hardware again)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    if (c1 < wg_count) {
      uint32_t nid = 0;
      if (wv < rounds) nid = *reinterpret_cast<const uint32_t*>(lists + EvqLists::NL + (((it + 1) & 1) * nls + wv * 64 + lane) * 4);
      fetch(w1c, nid, w2c, it & 1, lane);
    }
    // the cluster after those two: one ticket per workgroup (nothing left once c2 is past the end: the counter only grows)
    tk = NONE;
    if (c2 < wg_count && wv == 0 && lane == 0) {
      // in asm: the compiler's atomicAdd is a wave-aggregated atomic whose result it broadcasts (and waits for) at once; this one is
      // waited for by the s_waitcnt vmcnt(0) at the top of the next cluster
      const int one = 1;
      asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(tk) : "v"(ticket), "v"(one) : "memory");
    }
    pending = true;
    // ---- node block (bn, bs): moments -> entries; waves 0-1 hold the blocks of nodes 0-7, waves 2-3 those of nodes 8-15
    const int nown = s_nown;
    const int bn = ((wv >> 1) << 3) | (lane & 7), bs = ((wv & 1) << 3) | (lane >> 3), blk = bs * MAXN + bn;
    double e[NM];
    const bool has = bn < nown && bs < (int)snode[bn < nown ? bn : 0].len;
#pragma unroll
    for (int m = 0; m < NM; m++) e[m] = 0.0;
    if (has) {
      const int mirror = (int)smirror[blk];
#pragma unroll
      for (int m = 0; m < NM; m++) e[m] = lds[m * NBP + blk];
      if (mirror != blk) {
#pragma unroll
        for (int m = 0; m < NM; m++)
          if (GEN ? ev::symmetric_moment_gen(m) : ev::symmetric_moment(m)) e[m] += lds[m * NBP + mirror];
      }
    }
    {
      const int tid = wv * 64 + lane;
      if (tid < nown * 5) {
        const int n = tid / 5, a = tid - n * 5;
        rhs[(size_t)snode[n].node * 5 + a] = R[a * MAXN + n];
      }
    }
    double o[25];
    ev::pihna_expand(k, e, o);
    lds_barrier();   // every moment has been read: the image halves may overwrite the slice
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
      const int n0 = h * 8;
      if (n0 >= nown) break;   // uniform
      const uint32_t base = h == 0 ? 0u : (snode[8].obase & ~1u);   // even: the 16-byte phase of the segments is kept
      if ((wv >> 1) == h && has) {
        const int len5 = 5 * (int)snode[bn].len;
        double* dst = lds + (snode[bn].obase - base) + 5 * bs;
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
          for (int b = 0; b < 5; b++) dst[a * len5 + b] = o[a * 5 + b];
      }
      lds_barrier();
      const int n1 = nown < n0 + 8 ? nown : n0 + 8;
      for (int n = n0 + wv; n < n1; n += 4) {
        const HostPrepEv::Node nd = snode[n];
        const int cnt = 25 * (int)nd.len;
        double* out = val + (size_t)25 * nd.bptr;
        const double* img = lds + (nd.obase - base);
        const int sh = (int)(nd.obase & 1);
        typedef double v2d_t __attribute__((ext_vector_type(2)));
        const int npair = (cnt - sh) >> 1;
        const v2d_t* src = reinterpret_cast<const v2d_t*>(img + sh);
        v2d_t* dstg = reinterpret_cast<v2d_t*>(out + sh);
        for (int x = lane; x < npair; x += 64) __builtin_nontemporal_store(src[x], dstg + x);
        if (sh && lane == 0) __builtin_nontemporal_store(img[0], out);
        if (((cnt - sh) & 1) && lane == 1) __builtin_nontemporal_store(img[cnt - 1], out + cnt - 1);
      }
      lds_barrier();   // the half has been read (into the stores' registers)
    }
    RDC_TS(5);
    if (TL && lane == 0 && stamps) {
      long long* op = stamps + ((int64_t)c0 * 4 + wv) * 12;
#pragma unroll
      for (int x = 0; x < 6; x++) op[x] = ts[x];
      op[11] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
    c0 = c1; c1 = c2; w0c = w1c; w1c = w2c;   // c2, w2c: from the ticket, behind the barrier at the top
  }
#undef RDC_TS
}

// (A first resident form, k_tet4_evp -- two workgroups per CU, a loader wave, two LDS buffers: 2.45 vs 2.11 ms in round 2 -- was removed at the end of
// round 3: profiles/r02i_ab_evp.txt)

hipError_t launch_tet4_ev(const LaunchArgs& a, const PihnaK& k) {
  const EvDev& E = a.ev;
  // node records (same pack kernel and two-part rules as the pair kernels)
  hipError_t e = pack_nodes_pihna(a);
  if (e != hipSuccess) return e;
  if (a.ev_start) (void)hipEventRecord(a.ev_start, a.stream);
  const int wg_count = E.wg_count < 0 ? E.n_wg - E.wg_begin : E.wg_count;
  if (wg_count <= 0) return hipSuccess;
  const size_t acc = (size_t)(a.ev_general ? ev::NMG : ev::NM) * ev::NBP + 5 * ev::MAXN + (size_t)4 * E.nls * 2;
  const size_t lds_doubles = acc > E.max_out_doubles ? acc : E.max_out_doubles;
  const size_t lds_bytes = lds_doubles * sizeof(double);
  // default: pipelined resident workgroups, clusters handed out by a counter (the two parts of a two-part assembly, which may run
  // concurrently on two streams, have a counter each: ev_ticket[0] and ev_ticket[16]).  Not when a diagnostic knob of k_tet4_ev is set
  // ... and only for launches of at least 56 clusters per resident workgroup (43,000 on 256 CUs): a workgroup's start (ticket -> node
  // ids -> fetch: three round trips) and the tail of the launch are not hidden and cost ~20 us whatever the size -- K(55) = 10,700
  // clusters runs 0.184 vs 0.167 ms, K(75) 0.388 vs 0.370, K(85) = 40,000 clusters 0.577 vs 0.575, K(94) 0.713 vs 0.737, K(119)
  // 1.38 vs 1.50 ("grid" > 0 or "ev_resident" = 2 force it: tests)
  const int evq_grid = a.opt_grid > 0 ? a.opt_grid : (a.ev_grid > 0 ? (a.ev_general ? a.ev_grid : a.ev_grid / 2 * 3) : 768);   // three per CU (22 moments: two)
  if (a.opt_ev_resident && (!a.opt_ablate || (a.opt_ablate == 4 && !a.ev_general)) && a.opt_ev_occ == 3 && !a.opt_xcd && a.opt_stagger == 0 &&
      (a.opt_grid > 0 || a.opt_ev_resident == 2 || (int64_t)wg_count >= 56 * (int64_t)evq_grid)) {
    int grid = evq_grid;
    if (grid > wg_count) grid = wg_count;
    const size_t bytes = ((size_t)(a.ev_general ? ev::NMG : ev::NM) * ev::NBP + 5 * ev::MAXN + (size_t)4 * E.nls * 2) * sizeof(double) + EvqLists::bytes(E.nls);
    if (!a.ev_ticket) return hipErrorInvalidValue;   // the cluster counter: zeroed by the record pack kernel in front of this launch
#define RDC_EVQ(MODE, TLV, GENV)                                                                                                              \
  hipLaunchKernelGGL((k_tet4_evq<MODE, TLV, GENV>), dim3(grid), dim3(256), bytes, a.stream, E.desc, E.nlist, E.vloc, E.vslot, E.ntab, E.bpart, k, \
                     a.packed, a.val, a.rhs, E.nls, E.wg_begin, wg_count, a.stamps, a.opt_ev_bg, a.ev_ticket + (a.pack_part == 2 ? 16 : 0), E.wg_perm)
    if (a.ev_general) { if (a.exp_mode == 3) RDC_EVQ(3, false, true); else RDC_EVQ(0, false, true); }
    else if (a.opt_ablate == 4) { if (a.exp_mode == 3) RDC_EVQ(3, true, false); else RDC_EVQ(0, true, false); }
    else { if (a.exp_mode == 3) RDC_EVQ(3, false, false); else RDC_EVQ(0, false, false); }
#undef RDC_EVQ
    return hipGetLastError();
  }
#define RDC_EV(MODE, MINW)                                                                                          \
  hipLaunchKernelGGL((k_tet4_ev<MODE, MINW>), dim3(wg_count), dim3(256), lds_bytes, a.stream, E.desc, E.nlist, E.vloc, \
                     E.vslot, E.ntab, E.bpart, E.wg_perm, k, a.packed, a.val, a.rhs, E.nls, E.wg_begin, a.opt_xcd ? wg_count : 0, a.opt_stagger, a.stamps, a.opt_ev_bg)
  if (a.exp_mode == 3 && a.opt_ablate >= 1 && a.opt_ablate <= 4 && !a.ev_general) {   // diagnostic builds (1-3: timing only)
#define RDC_EVA(X)                                                                                                    \
  hipLaunchKernelGGL((k_tet4_ev<3, 3, X>), dim3(wg_count), dim3(256), lds_bytes, a.stream, E.desc, E.nlist, E.vloc, \
                     E.vslot, E.ntab, E.bpart, E.wg_perm, k, a.packed, a.val, a.rhs, E.nls, E.wg_begin, a.opt_xcd ? wg_count : 0, a.opt_stagger, a.stamps, a.opt_ev_bg)
    if (a.opt_ablate == 1) RDC_EVA(1); else if (a.opt_ablate == 2) RDC_EVA(2); else if (a.opt_ablate == 3) RDC_EVA(3); else RDC_EVA(4);
#undef RDC_EVA
    return hipGetLastError();
  }
  if (a.ev_general) {
    // every term on: 22 moments, 54 KB of LDS.  Register budget of TWO workgroups per CU (244 registers, no scratch): at three the visit
    // spills 79 registers (304 B of scratch per lane) and runs 8.8 instead of 3.0 ms on K(119) (profiles/r03_ev_ab_log.md, r03q)
#define RDC_EVG(MODE)                                                                                                       \
  hipLaunchKernelGGL((k_tet4_ev<MODE, 2, 0, true>), dim3(wg_count), dim3(256), lds_bytes, a.stream, E.desc, E.nlist, E.vloc, \
                     E.vslot, E.ntab, E.bpart, E.wg_perm, k, a.packed, a.val, a.rhs, E.nls, E.wg_begin, a.opt_xcd ? wg_count : 0, a.opt_stagger, a.stamps, a.opt_ev_bg)
    if (a.exp_mode == 3) RDC_EVG(3); else RDC_EVG(0);
#undef RDC_EVG
    return hipGetLastError();
  }
  if (a.exp_mode == 3) { if (a.opt_ev_occ == 2) RDC_EV(3, 2); else RDC_EV(3, 3); }
  else { if (a.opt_ev_occ == 2) RDC_EV(0, 2); else RDC_EV(0, 3); }
#undef RDC_EV
  return hipGetLastError();
}

}  // namespace rdc
