// Calibrates rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this library uses:
// a known number of bytes read with 4-byte and with 16-byte per-lane coalesced loads from a 1 GiB
// buffer (larger than the 256 MiB Infinity Cache), and written with 16-byte stores.
// Build: hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o tools/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read_dword(const uint32_t *p, size_t n, uint32_t *out) {
  uint32_t a = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a ^= p[i];
  if (a == 0x12345678u) out[0] = a;
}
__global__ void read_dwordx4(const uint4 *p, size_t n, uint32_t *out) {
  uint32_t a = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = p[i]; a ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (a == 0x12345678u) out[0] = a;
}
__global__ void write_dwordx4(uint4 *p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_uint4(1, 2, 3, 4);
}
int main() {
  const size_t bytes = 1ull << 30;
  void *buf; uint32_t *out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(read_dword, dim3(2048), dim3(256), 0, 0, (const uint32_t *)buf, bytes / 4, out);
  hipLaunchKernelGGL(read_dwordx4, dim3(2048), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, out);
  hipLaunchKernelGGL(write_dwordx4, dim3(2048), dim3(256), 0, 0, (uint4 *)buf, bytes / 16);
  hipDeviceSynchronize();
  printf("each kernel touches %zu bytes\n", bytes);
  return 0;
}
