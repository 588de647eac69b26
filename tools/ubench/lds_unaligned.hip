// Do LDS reads at 2-byte-aligned addresses work on gfx950 (unaligned DS access mode), and at what rate?
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void k_check(uint32_t* out, int shift_bytes) {
  __shared__ uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (uint16_t)(i * 7 + 1);
  __syncthreads();
  const uint32_t addr = (uint32_t)(uintptr_t)lds + 8 * threadIdx.x + shift_bytes;
  uint32_t v32; uint64_t v64; uint32_t r2a, r2b;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v32) : "v"(addr) : "memory");
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v64) : "v"(addr) : "memory");
  uint64_t r2;
  asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:3\n\ts_waitcnt lgkmcnt(0)" : "=v"(r2) : "v"(addr) : "memory");
  r2a = (uint32_t)r2; r2b = (uint32_t)(r2 >> 32);
  out[threadIdx.x * 5 + 0] = v32;
  out[threadIdx.x * 5 + 1] = (uint32_t)v64;
  out[threadIdx.x * 5 + 2] = (uint32_t)(v64 >> 32);
  out[threadIdx.x * 5 + 3] = r2a;
  out[threadIdx.x * 5 + 4] = r2b;
}

template <int MODE>   // 0: b32, 1: read2_b32, 2: b64
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, int shift_bytes, int iters) {
  __shared__ uint32_t lds[12288];
  for (int i = threadIdx.x; i < 12288; i += blockDim.x) lds[i] = i;
  __syncthreads();
  // per-lane row pointer as in the search kernel: pitch 97 dwords, lane = candidate
  uint32_t addr = (uint32_t)(uintptr_t)lds + 4 * ((threadIdx.x & 63) + 97 * (threadIdx.x >> 6)) + shift_bytes;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) { uint32_t v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(u * 388)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc += v; }
      if (MODE == 1) { uint64_t v; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "i"(u * 97 % 200), "i"(u * 97 % 200 + 1)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc += (uint32_t)v + (uint32_t)(v >> 32); }
      if (MODE == 2) { uint64_t v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(u * 388)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc += (uint32_t)v + (uint32_t)(v >> 32); }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  CK(hipSetDevice(0));
  uint32_t* d; CK(hipMalloc(&d, 64 * 5 * 4 + 64));
  std::vector<uint32_t> h(64 * 5);
  auto s16 = [](int i) { return (uint32_t)(uint16_t)(i * 7 + 1); };
  for (int shift : {0, 2, 4, 1}) {   // 4: the search kernels' case -- 64-bit reads at 4-byte-aligned addresses
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d, shift);
    CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    int bad32 = 0, bad64 = 0, bad2 = 0;
    for (int t = 0; t < 64; ++t) {
      if (shift & 1) continue;
      const int s = 4 * t + shift / 2;   // sample index
      const uint32_t e0 = s16(s) | s16(s + 1) << 16, e1 = s16(s + 2) | s16(s + 3) << 16, e3 = s16(s + 6) | s16(s + 7) << 16;
      bad32 += h[t * 5] != e0; bad64 += h[t * 5 + 1] != e0 || h[t * 5 + 2] != e1; bad2 += h[t * 5 + 3] != e0 || h[t * 5 + 4] != e3;
    }
    printf("shift %d bytes: ds_read_b32 wrong %d/64, ds_read_b64 wrong %d/64, ds_read2_b32 wrong %d/64   (lane0 b32 = %08x)\n", shift, bad32, bad64, bad2, h[0]);
  }
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2048, blocks = p.multiProcessorCount * 2;
  for (int mode = 0; mode < 3; ++mode)
    for (int shift : {0, 2, 4}) {
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, d, shift, iters);
        if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, d, shift, iters);
        if (mode == 2) hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(256), 0, 0, d, shift, iters);
      };
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      printf("%s shift %d: %.4f ms  (%.1f cycles per wave-load at 2.2 GHz, 8 waves/CU, dependent loads)\n",
             mode == 0 ? "ds_read_b32 " : mode == 1 ? "ds_read2_b32" : "ds_read_b64 ", shift, ms, ms * 1e-3 * 2.2e9 / (2.0 * iters * 8));
    }
  return 0;
}
