// Microbenchmark: what a GATHER costs on gfx950 -- wave-level buffer loads from an L2-resident table, by address pattern
// and by width.  The candidates kernel of the matcher (csm_bnb_cand_kernel) is bound by its vector-memory loads; timing
// builds (profiles/r05_cand_strip_loads.txt) showed its strip loads cost the same whether their lanes touch 14 or 6
// lines and whether a lane loads 16 bytes or 12 -- this program measures the cost model directly.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o /tmp/ubench_gather ; run: /tmp/ubench_gather [MB per slice]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int ITERS = 256;  // rounds of U loads per wave
constexpr int U = 8;        // loads in flight per wave

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// MODE 0: lanes contiguous (lane l at base + l * W), base random per load
//      1: every lane random
//      2: quads contiguous (4 lanes x W bytes in a row), quads random
//      3: eight lanes in one 128-byte line at 16-byte slots (lane = row of an 8 x 16-byte tile), lines random
//      4: "wall": ~5 lanes per line in arbitrary slots, 13 lines per load, lines within 4 KB of each other
//      5: sixteen lanes in one 128-byte line, each reading W bytes at 8-byte slots (two lanes may overlap), lines random
template <int MODE, int W>
__global__ __launch_bounds__(256) void k(const uint8_t *table, uint32_t slice_bytes, uint32_t *out, uint32_t seed) {
  const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 4u + (threadIdx.x >> 6));
  const uint8_t *base = table + (size_t)(blockIdx.x & 7u) * slice_bytes;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, (int)slice_bytes, 0x00020000);
  const uint32_t mask = slice_bytes - 1u;  // (power of two)
  uint32_t acc = 0u;
  for (int it = 0; it < ITERS; it++) {
    uint32_t a[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
      const uint32_t key = seed + (wave * ITERS + (uint32_t)it) * U + (uint32_t)j;
      uint32_t off;
      if (MODE == 0) off = ((mix(key) << 10) + lane * (uint32_t)W);
      else if (MODE == 1) off = mix(key * 64u + lane) * 16u;
      else if (MODE == 2) off = (mix(key * 16u + (lane >> 2)) << 6) + (lane & 3u) * (uint32_t)W;
      else if (MODE == 3) off = (mix(key * 8u + (lane >> 3)) << 7) + (lane & 7u) * 16u;
      else if (MODE == 4) off = ((mix(key) << 12) + (lane / 5u) * 256u + (mix(key * 64u + lane) & 7u) * 16u);
      else off = (mix(key * 4u + (lane >> 4)) << 7) + (lane & 15u) * 8u;
      a[j] = off & mask & ~15u;
      if (MODE == 0 || MODE == 2) a[j] = off & mask & ~3u;
      if (MODE == 5) a[j] = off & mask & ~7u;
    }
#pragma unroll
    for (int j = 0; j < U; j++) {
      if (W == 4) acc ^= __builtin_amdgcn_raw_buffer_load_b32(rs, (int)a[j], 0, 0);
      else if (W == 8) { const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)a[j], 0, 0); acc ^= v.x ^ v.y; }
      else if (W == 12) { const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)a[j], 0, 0); acc ^= v.x ^ v.y ^ v.z; }
      else { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a[j], 0, 0); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int MODE, int W>
int run(const char *name, const uint8_t *d_table, uint32_t slice_bytes, uint32_t *d_out, int blocks, int cus) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k<MODE, W>), dim3(blocks), dim3(256), 0, 0, d_table, slice_bytes, d_out, 1u + w);
  CHECK(hipDeviceSynchronize());
  const int reps = 5;
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k<MODE, W>), dim3(blocks), dim3(256), 0, 0, d_table, slice_bytes, d_out, 77u + r);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double loads = (double)reps * blocks * 4.0 * ITERS * U;  // wave-level load instructions
  const double per_cu_per_us = loads / cus / (ms * 1e3);
  printf("%-58s W=%2d  %8.3f ms  %7.2f wave-loads/us/CU  %6.1f clk/wave-load/CU (2.4 GHz)  %6.2f TB/s useful\n", name, W, ms / reps,
         per_cu_per_us, 2400.0 / per_cu_per_us, loads * 64.0 * W / (ms * 1e-3) / 1e12);
  return 0;
}

int main(int argc, char **argv) {
  const uint32_t slice_mb = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;  // per (blockIdx & 7): ~ per XCD
  const uint32_t slice_bytes = slice_mb << 20;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint8_t *d_table;
  uint32_t *d_out;
  const int blocks = cus * 8;
  CHECK(hipMalloc(&d_table, (size_t)slice_bytes * 8));
  CHECK(hipMemset(d_table, 1, (size_t)slice_bytes * 8));
  CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
  printf("# %s, %d CUs, table slice %u MB per block class (8 classes), %d blocks x 4 waves, %d loads in flight per wave\n", prop.name, cus,
         slice_mb, blocks, U);
#define ROW(M, NAME) \
  if (run<M, 4>(NAME, d_table, slice_bytes, d_out, blocks, cus)) return 1; \
  if (run<M, 8>(NAME, d_table, slice_bytes, d_out, blocks, cus)) return 1; \
  if (run<M, 12>(NAME, d_table, slice_bytes, d_out, blocks, cus)) return 1; \
  if (run<M, 16>(NAME, d_table, slice_bytes, d_out, blocks, cus)) return 1;
  ROW(0, "0 lanes contiguous")
  ROW(1, "1 every lane random (16-byte aligned)")
  ROW(2, "2 quads contiguous, quads random")
  ROW(3, "3 eight lanes per 128-B line at 16-B slots, lines random")
  ROW(4, "4 wall: 13 lines per load, ~5 lanes per line, any slot")
  ROW(5, "5 sixteen lanes per 128-B line at 8-B slots, lines random")
  return 0;
}
