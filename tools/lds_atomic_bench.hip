// Microbenchmark: LDS throughput of FP64 atomic adds vs plain 64-bit stores/loads on gfx950, for the
// address patterns of the row-gather kernel.  hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
constexpr int N = 4096;  // doubles of LDS used (32 KB)
template <int MODE>
__global__ void __launch_bounds__(256) k(const int* __restrict__ idx, double* out, int iters, long long* cyc) {
  __shared__ double lds[N];
  for (int x = threadIdx.x; x < N; x += 256) lds[x] = 0.0;
  __syncthreads();
  int my[8];
  for (int r = 0; r < 8; r++) my[r] = idx[(blockIdx.x % 16) * 2048 + r * 256 + threadIdx.x];
  double acc = 0.0;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (MODE == 0) __hip_atomic_fetch_add(&lds[my[r]], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 1) lds[my[r]] = (double)it;
      if (MODE == 2) acc += lds[my[r]];
    }
  }
  long long t1 = clock64();
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x] + acc;
}
int main() {
  const int nblk = 256 * 2;
  int* d_idx; double* d_out; long long* d_cyc;
  hipMalloc(&d_idx, 16 * 2048 * sizeof(int)); hipMalloc(&d_out, nblk * 256 * sizeof(double)); hipMalloc(&d_cyc, nblk * sizeof(long long));
  const char* pat_name[] = {"consecutive", "random distinct-ish (10-dword granules)", "6 lanes same address (groups)", "24 lanes same address", "stride 21 doubles"};
  for (int pat = 0; pat < 5; pat++) {
    std::vector<int> h(16 * 2048);
    srand(1);
    for (int b = 0; b < 16; b++)
      for (int r = 0; r < 8; r++)
        for (int t = 0; t < 256; t++) {
          int v;
          if (pat == 0) v = (r * 256 + t) % N;
          else if (pat == 1) v = (5 * (rand() % 800) + rand() % 5) % N;
          else if (pat == 2) v = (5 * ((t / 6) * 37 % 800) + r % 5) % N;
          else if (pat == 3) v = (5 * ((t / 24) * 37 % 800) + r % 5) % N;
          else v = (t * 21 + r) % N;
          h[b * 2048 + r * 256 + t] = v;
        }
    hipMemcpy(d_idx, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; mode++) {
      const int iters = 200;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(256), 0, 0, d_idx, d_out, iters, d_cyc);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(256), 0, 0, d_idx, d_out, iters, d_cyc);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nblk), dim3(256), 0, 0, d_idx, d_out, iters, d_cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> c(nblk); hipMemcpy(c.data(), d_cyc, nblk * sizeof(long long), hipMemcpyDeviceToHost);
      double avg = 0; for (auto x : c) avg += x; avg /= nblk;
      // 2 WGs per CU resident (512 blocks / 256 CUs), 4 waves each: per-CU LDS sees 8 waves
      const double instr_per_wave = iters * 8.0;
      printf("%-42s %-8s wave-cycles/instr %7.1f   CU-cycles/wave-instr %6.2f  (kernel %.3f ms)\n", pat_name[pat],
             mode == 0 ? "atomic" : (mode == 1 ? "store" : "load"), avg / instr_per_wave, avg / instr_per_wave / 8.0, ms);
    }
  }
  return 0;
}
