// ubench_stage.hip -- what does it cost a workgroup to stage a 35 KB table from its target's grid slot into LDS?
// 10,000 workgroups of 512 threads and 75 KB of LDS (the scan matcher's shape), each copying `bytes` from
// base + slot * stride (+ offset) to LDS, slot = (workgroup / 10) % slots, workgroups of one slot on one XCD as in
// csm_bnb_kernel.  Reports the kernel time for the matcher's layout (slots 6.26 MB apart in a 6.3 GB allocation) and
// for a compact array of tables, for hipMalloc'ed memory.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stage.hip -o tools/ubench_stage && tools/ubench_stage
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512) void stage_kernel(const unsigned char *base, size_t stride, size_t offset, int bytes, int slots,
                                                    int per_xcd, int n, unsigned *out, int rounds) {
  extern __shared__ __align__(16) unsigned char smem[];
  const unsigned bid = blockIdx.x;
  const int item = (int)((bid & 7u) * (unsigned)per_xcd + (bid >> 3));
  if ((int)(bid >> 3) >= per_xcd || item >= n) return;
  const int slot = (item / 10) % slots;
  const uint4 *gp = reinterpret_cast<const uint4 *>(base + (size_t)slot * stride + offset);
  uint4 *sp = reinterpret_cast<uint4 *>(smem);
  const int n16 = bytes / 16;
  unsigned acc = 0;
  for (int r = 0; r < rounds; r++) {
    for (int i = threadIdx.x; i < n16; i += 4 * 512) {
      uint4 v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = gp[min(i + j * 512, n16 - 1)];
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (i + j * 512 < n16) sp[i + j * 512] = v[j];
    }
    __syncthreads();
    acc += smem[(threadIdx.x * 37 + r) % bytes];
    __syncthreads();
  }
  if (acc == 0xffffffffu) out[0] = acc;
  if (threadIdx.x == 0) atomicAdd(out + 1, 1u);
}

static float run(const unsigned char *base, size_t stride, size_t offset, int bytes, int slots, int n, unsigned *d_out,
                 size_t lds, int rounds) {
  const int per_xcd = (n + 7) / 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int it = 0; it < 4; it++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(stage_kernel, dim3(per_xcd * 8), dim3(512), lds, 0, base, stride, offset, bytes, slots, per_xcd, n, d_out, rounds);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("launch error: %s\n", hipGetErrorString(err));
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it > 0 && ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t stride = 6256896, offset = 3875328 + 122496;
  const int bytes = 35712, slots = 1000, n = 10000;
  const size_t lds = 75 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void *>(stage_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  unsigned char *big = nullptr, *compact = nullptr;
  unsigned *d_out;
  if (hipMalloc(&big, stride * slots) != hipSuccess || hipMalloc(&compact, (size_t)bytes * slots) != hipSuccess) return 1;
  hipMalloc(&d_out, 8);
  hipMemset(d_out, 0, 8);
  hipMemset(big, 1, stride * slots);
  hipMemset(compact, 1, (size_t)bytes * slots);
  hipDeviceSynchronize();
  for (int rounds : {1, 4}) {
    printf("rounds %d: slots 6.26 MB apart %.3f ms | compact tables %.3f ms | one slot for all %.3f ms | no LDS need (1 KB) strided %.3f ms\n", rounds,
           run(big, stride, offset, bytes, slots, n, d_out, lds, rounds), run(compact, bytes, 0, bytes, slots, n, d_out, lds, rounds),
           run(big, 0, offset, bytes, slots, n, d_out, lds, rounds), run(big, stride, offset, bytes, slots, n, d_out, 1024 + 35712, rounds));
  }
  unsigned h[2];
  hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost);
  printf("workgroups that ran: %u (expected %d)\n", h[1], 2 * 4 * 4 * n);
  // the same strided reads with the whole 6.3 GB touched in between (the matcher's grid build writes all of it)
  return 0;
}
