// Microbenchmark: what does a ds_add_f64 / ds_write_b64 wave instruction cost on gfx950 when only SOME lanes are active?
// (the element-visit kernel issues its row-i atomics with the lanes whose visit has more than i rows)
// All active lanes hit distinct banks (index = lane).  hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/lds_mask_model.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long mask, double* out, int iters, long long* cyc) {
  __shared__ double lds[8192];
  for (int x = threadIdx.x; x < 8192; x += 256) lds[x] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int my = lane + (threadIdx.x >> 6) * 2048;
  const bool active = (mask >> lane) & 1ull;
  long long t0 = clock64();
  if (active) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        if (MODE == 0) __hip_atomic_fetch_add(&lds[my], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 1) { lds[my] = (double)(it + r); __builtin_amdgcn_sched_barrier(0); }
      }
    }
  }
  long long t1 = clock64();
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
int main() {
  const int nblk = 512;
  double* d_out; long long* d_cyc;
  hipMalloc(&d_out, nblk * 256 * sizeof(double)); hipMalloc(&d_cyc, nblk * sizeof(long long));
  struct Pat { const char* name; unsigned long long mask; };
  std::vector<Pat> pats = {
    {"all 64 lanes", ~0ull}, {"lanes 0-47 (3 groups)", 0x0000FFFFFFFFFFFFull}, {"lanes 0-31 (2 groups)", 0x00000000FFFFFFFFull},
    {"lanes 0-15 (1 group)", 0xFFFFull}, {"lanes 0-7", 0xFFull}, {"lane 0", 1ull},
    {"one lane per group (0,16,32,48)", 0x0001000100010001ull}, {"4 lanes per group", 0x000F000F000F000Full},
    {"8 lanes per group", 0x00FF00FF00FF00FFull}, {"every other lane", 0x5555555555555555ull},
    {"groups 0 and 2", 0x0000FFFF0000FFFFull},
  };
  for (auto& p : pats) {
    double res[2];
    for (int mode = 0; mode < 2; mode++) {
      const int iters = 200;
      for (int rep = 0; rep < 2; rep++) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(256), 0, 0, p.mask, d_out, iters, d_cyc);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(256), 0, 0, p.mask, d_out, iters, d_cyc);
        hipDeviceSynchronize();
      }
      std::vector<long long> c(nblk); hipMemcpy(c.data(), d_cyc, nblk * sizeof(long long), hipMemcpyDeviceToHost);
      double avg = 0; for (auto x : c) avg += x; avg /= nblk;
      res[mode] = avg / (iters * 8.0) / 8.0;  // 8 waves (2 workgroups of 4) share the CU's LDS
    }
    printf("%-36s atomic %6.2f  store %6.2f  CU-cycles/wave-instr\n", p.name, res[0], res[1]);
  }
  return 0;
}
