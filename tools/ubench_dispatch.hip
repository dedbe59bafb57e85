// ubench_dispatch.hip -- how well does the hardware dispatcher keep the chip's workgroup slots filled?
// Workgroups that do nothing but wait for a given time (s_memrealtime, 100 MHz), launched as the scan matcher launches
// its pairs: 10,000 workgroups, far more than fit at once.  Kernel time against sum(durations) / slots says how much of
// the slot-time the dispatcher leaves empty, as a function of the workgroup's shape (threads, LDS, registers), of the
// spread of the durations and of the launch form (one workgroup per item / persistent workgroups pulling items).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_dispatch.hip -o tools/ubench_dispatch && tools/ubench_dispatch
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

template <int THREADS, int VGPRS, bool PERSISTENT>
__global__ __launch_bounds__(THREADS) void wait_kernel(const unsigned *ticks, int n, unsigned *counter) {
  extern __shared__ unsigned char smem[];
  if (VGPRS > 64) asm volatile("v_mov_b32 v%c0, 0" ::"i"(VGPRS - 1) : "memory");  // forces the allocation
  __shared__ int s_item;
  for (;;) {
    int item = blockIdx.x;
    if (PERSISTENT) {
      if (threadIdx.x == 0) s_item = (int)atomicAdd(counter, 1u);
      __syncthreads();
      item = s_item;
      __syncthreads();
      if (item >= n) return;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t1 = t0 + ticks[item];
    smem[threadIdx.x] = (unsigned char)item;
    while (__builtin_amdgcn_s_memrealtime() < t1) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (!PERSISTENT) return;
  }
}

template <int THREADS, int VGPRS, bool PERSISTENT>
double run(const std::vector<unsigned> &ticks, size_t lds, int wgs_per_cu, const char *name) {
  const int n = (int)ticks.size();
  unsigned *d_ticks, *d_counter;
  hipMalloc(&d_ticks, 4 * n);
  hipMalloc(&d_counter, 4);
  hipMemcpy(d_ticks, ticks.data(), 4 * n, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void *>(wait_kernel<THREADS, VGPRS, PERSISTENT>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    hipMemset(d_counter, 0, 4);
    hipEventRecord(e0, 0);
    const int grid = PERSISTENT ? 256 * wgs_per_cu : n;
    hipLaunchKernelGGL((wait_kernel<THREADS, VGPRS, PERSISTENT>), dim3(grid), dim3(THREADS), lds, 0, d_ticks, n, d_counter);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  double sum = 0;
  for (unsigned t : ticks) sum += t;
  const double ideal_ms = sum / 100e6 * 1e3 / (256.0 * wgs_per_cu);
  printf("%-46s kernel %7.3f ms  ideal %7.3f ms  slots filled %5.1f %%\n", name, best, ideal_ms, 100.0 * ideal_ms / best);
  hipFree(d_ticks);
  hipFree(d_counter);
  return best;
}

int main() {
  const int n = 10000;
  std::mt19937 rng(1);
  std::vector<unsigned> uniform(n, 30000);  // 300 us
  std::vector<unsigned> spread(n);
  std::lognormal_distribution<double> ln(std::log(25000.0), 0.6);  // median 250 us, heavy tail (as the matcher's pairs)
  for (auto &t : spread) t = (unsigned)std::min(ln(rng), 400000.0);
  const size_t lds75 = 75 * 1024, lds36 = 36 * 1024;
  printf("10,000 workgroups that wait; 512 threads + 75 KB LDS + 128 VGPRs = 2 per CU is the matcher's shape\n");
  run<512, 128, false>(uniform, lds75, 2, "512 thr, 75 KB, 128 vgpr, uniform 300 us");
  run<512, 128, false>(spread, lds75, 2, "512 thr, 75 KB, 128 vgpr, spread");
  run<512, 128, true>(spread, lds75, 2, "  ... persistent (512 workgroups pull items)");
  run<512, 64, false>(spread, lds75, 2, "512 thr, 75 KB, 64 vgpr, spread");
  run<512, 128, false>(spread, 1024, 2, "512 thr, 1 KB, 128 vgpr, spread");
  run<256, 128, false>(spread, lds36, 4, "256 thr, 36 KB, 128 vgpr, spread (4 per CU)");
  run<256, 128, true>(spread, lds36, 4, "  ... persistent (1024 workgroups pull items)");
  run<256, 128, false>(uniform, lds36, 4, "256 thr, 36 KB, 128 vgpr, uniform");
  run<64, 128, false>(spread, 9 * 1024, 16, "64 thr, 9 KB, 128 vgpr, spread (16 per CU)");
  return 0;
}
