// LDS read rate of the access patterns the 16-bit search kernel can use for its window rows (6 dwords per lane and row):
//   lane stride 2 dwords, ds_read_b64 (8-byte aligned)                  -- two candidates per lane (round 1)
//   lane stride 3 dwords, ds_read2_b32 offsets (k, k+1)                 -- three candidates per lane, adjacent dwords
//   lane stride 3 dwords, ds_read2_b32 offsets (k, k+3)                 -- the same six dwords as pairs three dwords apart
//   lane stride 3 dwords, ds_read_b32                                    -- single dwords
// 8 waves per CU, 8 independent loads between waits.  Build: hipcc --offload-arch=gfx950 -O3 -o lds_stride3 lds_stride3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, int iters) {
  __shared__ uint32_t lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int stride = MODE == 0 ? 2 : 3;
  uint32_t addr = (uint32_t)(uintptr_t)lds + 4 * (stride * lane + 162 * wave);
  uint32_t uaddr = (uint32_t)(uintptr_t)lds + 16 * wave;   // wave-uniform address (the current-block reads of the search kernels)
  asm volatile("" : "+v"(uaddr));
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint64_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[u]) : "v"(addr), "i"((u >> 1) * 648 + (u & 1) * 8));
      if (MODE == 1) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v[u]) : "v"(addr), "i"((u & 3) * 2), "i"((u & 3) * 2 + 1));
      if (MODE == 2) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v[u]) : "v"(addr), "i"(u & 3), "i"((u & 3) + 3));
      if (MODE == 3) { uint32_t w; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(w) : "v"(addr), "i"(u * 4)); v[u] = w; }
      if (MODE == 4) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 w; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w) : "v"(uaddr), "i"(u * 128)); v[u] = w.x + ((uint64_t)w.w << 32); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += (uint32_t)v[u] + (uint32_t)(v[u] >> 32);
    addr ^= 4 * 162 * 4;   // another row group next time (keeps the address live, same bank pattern)
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  CK(hipSetDevice(0));
  uint32_t* d; CK(hipMalloc(&d, 64));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4096, blocks = p.multiProcessorCount * 2;
  const char* names[5] = {"stride 2, ds_read_b64           ", "stride 3, ds_read2_b32 (k, k+1) ", "stride 3, ds_read2_b32 (k, k+3) ", "stride 3, ds_read_b32           ",
                          "uniform address, ds_read_b128   "};
  for (int mode = 0; mode < 5; ++mode) {
    auto launch = [&]() {
      if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
      if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
      if (mode == 2) hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(256), 0, 0, d, iters);
      if (mode == 3) hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(256), 0, 0, d, iters);
      if (mode == 4) hipLaunchKernelGGL(k_rate<4>, dim3(blocks), dim3(256), 0, 0, d, iters);
    };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    // per CU: 8 waves x iters x 8 loads; bytes per wave-load: 512 (b64 / read2), 256 (b32)
    const double loads = 8.0 * iters * 8, ns = ms * 1e6;
    printf("%s %.4f ms  %.2f wave-loads per ns per CU, %.0f bytes per ns per CU\n", names[mode], ms, loads / ns, loads * (mode == 3 ? 256 : mode == 4 ? 16 : 512) / ns);
  }
  return 0;
}
