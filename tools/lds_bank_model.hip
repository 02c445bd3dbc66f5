// Microbenchmark: which lanes of a ds_add_f64 / ds_write_b64 wave instruction conflict on gfx950?
// Each pattern gives lane l of every wave the LDS double index f(l); all patterns touch 64 distinct
// addresses (no same-address serialisation), they differ in which lanes share a bank.
// hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/lds_bank_model.hip -o /tmp/lds_bank_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
constexpr int N = 8192;  // doubles (64 KB)
template <int MODE>
__global__ void __launch_bounds__(256) k(const int* __restrict__ idx, double* out, int iters, long long* cyc) {
  __shared__ double lds[N];
  for (int x = threadIdx.x; x < N; x += 256) lds[x] = 0.0;
  __syncthreads();
  const int my = idx[threadIdx.x & 63] + (threadIdx.x >> 6) * 2048;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (MODE == 0) __hip_atomic_fetch_add(&lds[my + r * 0], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 1) { lds[my] = (double)(it + r); __builtin_amdgcn_sched_barrier(0); }
    }
  }
  long long t1 = clock64();
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
int main() {
  const int nblk = 512;
  int* d_idx; double* d_out; long long* d_cyc;
  hipMalloc(&d_idx, 64 * sizeof(int)); hipMalloc(&d_out, nblk * 256 * sizeof(double)); hipMalloc(&d_cyc, nblk * sizeof(long long));
  struct Pat { const char* name; std::function<int(int)> f; };
  std::vector<int> perm(64); for (int i = 0; i < 64; i++) perm[i] = i;
  srand(3); std::random_shuffle(perm.begin(), perm.end());
  std::vector<int> rnd(64); for (int i = 0; i < 64; i++) rnd[i] = rand() % 2048;
  std::vector<Pat> pats = {
    {"identity l", [](int l) { return l; }},
    {"random permutation of 0..63", [&](int l) { return perm[l]; }},
    {"random in 0..2047", [&](int l) { return rnd[l]; }},
    {"stride 2 doubles", [](int l) { return 2 * l; }},
    {"stride 4 doubles", [](int l) { return 4 * l; }},
    {"stride 8 doubles", [](int l) { return 8 * l; }},
    {"stride 16 doubles", [](int l) { return 16 * l; }},
    {"stride 32 doubles (32 distinct? )", [](int l) { return 32 * l; }},
    {"lanes l,l+32 same bank: l%32 + (l/32)*64", [](int l) { return l % 32 + (l / 32) * 64; }},
    {"lanes l,l+32 offset 32: l%32 + (l/32)*32 (=identity)", [](int l) { return l % 32 + (l / 32) * 32; }},
    {"lanes l,l+16 same bank: l%16 + (l/16)*64", [](int l) { return l % 16 + (l / 16) * 64; }},
    {"lanes l,l+16 offset 32: l%16 + (l/16)*32", [](int l) { return l % 16 + (l / 16) * 32; }},
    {"lanes l,l+8 offset 64: l%8 + (l/8)*64", [](int l) { return l % 8 + (l / 8) * 64; }},
    {"lanes l,l+8 offset 32: l%8 + (l/8)*32", [](int l) { return l % 8 + (l / 8) * 32; }},
    {"adjacent pair offset 64: l/2 + (l%2)*64", [](int l) { return l / 2 + (l % 2) * 64; }},
    {"adjacent pair offset 32: l/2 + (l%2)*32", [](int l) { return l / 2 + (l % 2) * 32; }},
    {"adjacent pair offset 16: l/2 + (l%2)*16 ", [](int l) { return (l / 2) % 16 + ((l / 2) / 16) * 32 + (l % 2) * 16; }},
    {"4 adjacent offset 64: l/4 + (l%4)*64", [](int l) { return l / 4 + (l % 4) * 64; }},
    {"4 adjacent offset 32: l/4 + (l%4)*32", [](int l) { return l / 4 + (l % 4) * 32; }},
    {"8 adjacent offset 64", [](int l) { return l / 8 + (l % 8) * 64; }},
    {"within-32 pairs (l, l^16) offset 64", [](int l) { return (l & 15) + ((l >> 5) << 4) + ((l >> 4) & 1) * 64; }},
    {"row-slice like: 3 nodes x ~21 lanes, slot*5", [](int l) { return (l / 22) * 375 + ((l % 22) * 7 % 15) * 5; }},
  };
  for (auto& p : pats) {
    std::vector<int> h(64);
    for (int l = 0; l < 64; l++) h[l] = p.f(l) % 2048;
    std::vector<int> s = h; std::sort(s.begin(), s.end());
    const bool distinct = std::adjacent_find(s.begin(), s.end()) == s.end();
    hipMemcpy(d_idx, h.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    double res[2];
    for (int mode = 0; mode < 2; mode++) {
      const int iters = 200;
      for (int rep = 0; rep < 2; rep++) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(256), 0, 0, d_idx, d_out, iters, d_cyc);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(256), 0, 0, d_idx, d_out, iters, d_cyc);
        hipDeviceSynchronize();
      }
      std::vector<long long> c(nblk); hipMemcpy(c.data(), d_cyc, nblk * sizeof(long long), hipMemcpyDeviceToHost);
      double avg = 0; for (auto x : c) avg += x; avg /= nblk;
      res[mode] = avg / (iters * 8.0) / 8.0;  // 8 waves share the CU's LDS
    }
    printf("%-56s %s atomic %6.2f  store %6.2f  CU-cycles/wave-instr\n", p.name, distinct ? "distinct" : "DUPLICATE", res[0], res[1]);
  }
  return 0;
}
