// Microbenchmark: issue rate of candidate "add a byte of w into an accumulator" sequences on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
  uint32_t a[16];
  uint32_t w0 = seed * (threadIdx.x + 1), w1 = w0 ^ 0x9e3779b9u, w2 = w0 * 3u, w3 = w1 * 5u;
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = i;
  for (int it = 0; it < ITERS; it++) {
    if (MODE == 0) {  // plain v_add_u32: 16 per iter
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(w0));
    } else if (MODE == 1) {  // SDWA byte add: 16 per iter
#pragma unroll
      for (int i = 0; i < 4; i++) {
        asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(a[4 * i + 0]) : "v"(w0));
        asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[4 * i + 1]) : "v"(w1));
        asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "+v"(a[4 * i + 2]) : "v"(w2));
        asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(a[4 * i + 3]) : "v"(w3));
      }
    } else if (MODE == 2) {  // v_bfe_u32 + v_add_u32: 8 lookups per iter (16 instr)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        uint32_t t;
        asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(t) : "v"(w0));
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(t));
      }
    } else if (MODE == 3) {  // v_sad_u8: 16 per iter (sums 4 bytes each)
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_sad_u8 %0, %1, 0, %0" : "+v"(a[i]) : "v"(w0));
    } else if (MODE == 4) {  // v_perm_b32: 16 per iter
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(w0), "v"(w1));
    } else if (MODE == 5) {  // v_pk_add_u16: 16 per iter
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a[i]) : "v"(w0));
    } else if (MODE == 6) {  // v_dot4_u32_u8: 16 per iter
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w0), "v"(w1));
    } else if (MODE == 7) {  // v_alignbyte_b32: 16 per iter
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_alignbyte_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(w0), "v"(w1));
    } else if (MODE == 8) {  // v_add3_u32: 16 per iter
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w0), "v"(w1));
    } else if (MODE == 9) {  // v_and_b32 with literal-free mask + add: like MODE 2 but v_and
#pragma unroll
      for (int i = 0; i < 8; i++) {
        uint32_t t;
        asm volatile("v_and_b32 %0, 0xff, %1" : "=v"(t) : "v"(w0));
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(t));
      }
    } else if (MODE == 10) {  // v_mad_u32_u24: 16 per iter
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w0), "v"(w1));
    } else if (MODE == 11) {  // v_lshl_add_u32
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_lshl_add_u32 %0, %1, 1, %0" : "+v"(a[i]) : "v"(w0));
    } else if (MODE == 13) {  // v_pk_mad_u16 (VOP3P): 16 per iter
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w0), "v"(w1));
    } else if (MODE == 14) {  // v_mul_u32_u24 (VOP2) + v_add_u32: 8 multiply-adds per iter
#pragma unroll
      for (int i = 0; i < 8; i++) {
        uint32_t t;
        asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(t) : "v"(w0), "v"(w1));
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(t));
      }
    } else if (MODE == 15) {  // v_alignbit_b32: 16 per iter
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(w0), "v"(w1));
    } else if (MODE == 16) {  // v_pk_mul_lo_u16: 16 per iter
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_pk_mul_lo_u16 %0, %1, %0" : "+v"(a[i]) : "v"(w0));
    } else if (MODE == 12) {  // v_add_u32_sdwa with WORD sel
#pragma unroll
      for (int i = 0; i < 8; i++) {
        asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "+v"(a[2 * i + 0]) : "v"(w0));
        asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a[2 * i + 1]) : "v"(w1));
      }
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> int run(const char *name, uint32_t *d_out, int blocks) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  // wave-instructions per SIMD: blocks*4 waves / (256 CU * 4 SIMD) * ITERS * 16
  double winst = (double)blocks * 4 / 1024.0 * ITERS * 16;
  printf("%-28s %8.3f ms  %6.2f ns per wave-instr per SIMD  (~%.2f clk @2.4GHz)\n", name, ms, ms * 1e6 / winst, ms * 1e6 / winst * 2.4);
  return 0;
}

int main() {
  uint32_t *d; CHECK(hipMalloc(&d, 256 * 8192 * 4));
  for (int blocks : {256 * 1, 256 * 4, 256 * 8}) {
    printf("---- %d blocks of 256 threads (%d waves/SIMD)\n", blocks, blocks / 256);
    run<0>("v_add_u32", d, blocks);
    run<1>("v_add_u32_sdwa BYTE", d, blocks);
    run<12>("v_add_u32_sdwa WORD", d, blocks);
    run<2>("v_bfe_u32 + v_add_u32 (x8)", d, blocks);
    run<9>("v_and_b32 + v_add_u32 (x8)", d, blocks);
    run<3>("v_sad_u8", d, blocks);
    run<4>("v_perm_b32", d, blocks);
    run<5>("v_pk_add_u16", d, blocks);
    run<6>("v_dot4_u32_u8", d, blocks);
    run<7>("v_alignbyte_b32", d, blocks);
    run<8>("v_add3_u32", d, blocks);
    run<10>("v_mad_u32_u24", d, blocks);
    run<11>("v_lshl_add_u32", d, blocks);
    run<13>("v_pk_mad_u16", d, blocks);
    run<16>("v_pk_mul_lo_u16", d, blocks);
    run<14>("v_mul_u32_u24 + v_add_u32 (x8)", d, blocks);
    run<15>("v_alignbit_b32", d, blocks);
  }
  return 0;
}
