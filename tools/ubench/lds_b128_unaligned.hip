// Does ds_read_b128 / ds_read_b96 at a 4-byte-aligned (not 16-byte-aligned) LDS address work on gfx950, and at what rate beside the two
// ds_read2_b32 it would replace in me_search_kernel (lane stride 1 dword, row pitch 49 dwords: the window layout of the 8-bit kernel)?
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_b128_unaligned lds_b128_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

__global__ void k_check(uint32_t* out) {
  __shared__ uint32_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0x01000000u + i;
  __syncthreads();
  const uint32_t addr = (uint32_t)(uintptr_t)lds + 4 * (threadIdx.x + 49 * 3);   // dword threadIdx.x + 147: every alignment mod 16 bytes occurs
  u32x4 q; u32x3 t;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr) : "memory");
  asm volatile("ds_read_b96 %0, %1 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(addr) : "memory");
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 7 + i] = q[i];
  for (int i = 0; i < 3; ++i) out[threadIdx.x * 7 + 4 + i] = t[i];
}

template <int MODE>   // 0: two ds_read2_b32 (dwords k,k+1 and k+2,k+3), 1: one ds_read_b128, 2: one ds_read_b128 at 16-byte-aligned addresses (lane stride 4 dwords), 3: four ds_read2_b32 (what two b128 of the A and B streams replace)
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, int iters) {
  __shared__ uint32_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t addr = (uint32_t)(uintptr_t)lds + 4 * ((MODE == 2 ? 4 * lane : lane) + 49 * wv);
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {
        uint64_t a, b;
        asm volatile("ds_read2_b32 %0, %2 offset0:%3 offset1:%4\n\tds_read2_b32 %1, %2 offset0:%5 offset1:%6" : "=&v"(a), "=&v"(b) : "v"(addr), "i"(u * 49 % 200), "i"(u * 49 % 200 + 1), "i"(u * 49 % 200 + 2), "i"(u * 49 % 200 + 3));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += (uint32_t)a + (uint32_t)(a >> 32) + (uint32_t)b + (uint32_t)(b >> 32);
      } else if (MODE == 3) {
        uint64_t a, b, c, d;
        asm volatile("ds_read2_b32 %0, %4 offset0:%5 offset1:%6\n\tds_read2_b32 %1, %4 offset0:%7 offset1:%8\n\tds_read2_b32 %2, %4 offset0:%6 offset1:%7\n\tds_read2_b32 %3, %4 offset0:%8 offset1:%9"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr), "i"(u * 49 % 200), "i"(u * 49 % 200 + 1), "i"(u * 49 % 200 + 2), "i"(u * 49 % 200 + 3), "i"(u * 49 % 200 + 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += (uint32_t)a + (uint32_t)(b >> 32) + (uint32_t)c + (uint32_t)(d >> 32);
      } else {
        u32x4 q;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(addr), "i"(4 * (u * 49 % 200) * (MODE == 2 ? 4 : 1)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += q[0] + q[1] + q[2] + q[3];
      }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  CK(hipSetDevice(0));
  uint32_t* d; CK(hipMalloc(&d, 64 * 7 * 4 + 64));
  std::vector<uint32_t> h(64 * 7);
  hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
  int bad128 = 0, bad96 = 0;
  for (int t = 0; t < 64; ++t) {
    for (int i = 0; i < 4; ++i) bad128 += h[t * 7 + i] != 0x01000000u + t + 147 + i;
    for (int i = 0; i < 3; ++i) bad96 += h[t * 7 + 4 + i] != 0x01000000u + t + 148 + i;
  }
  printf("4-byte-aligned addresses (all four alignments mod 16): ds_read_b128 wrong dwords %d/256, ds_read_b96 wrong dwords %d/192\n", bad128, bad96);
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2048, blocks = p.multiProcessorCount * 2;
  const char* names[4] = {"2 x ds_read2_b32 (k,k+1)(k+2,k+3), lane stride 1 dword", "1 x ds_read_b128, lane stride 1 dword (4-byte aligned)", "1 x ds_read_b128, lane stride 4 dwords (16-byte aligned)",
                          "4 x ds_read2_b32: (k,k+1)(k+2,k+3)(k+1,k+2)(k+3,k+4)"};
  for (int mode = 0; mode < 4; ++mode) {
    auto launch = [&]() {
      if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
      if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
      if (mode == 2) hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(256), 0, 0, d, iters);
      if (mode == 3) hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(256), 0, 0, d, iters);
    };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    // per CU: 2 workgroups x 4 waves x iters x 8 groups; a group delivers 16 bytes per lane (mode 3: 32)
    const double groups = 2.0 * 4 * iters * 8;
    printf("%-62s %.4f ms  %.3f groups per ns per CU, %.0f bytes per ns per CU\n", names[mode], ms, groups / (ms * 1e6), groups * 64 * (mode == 3 ? 32 : 16) / (ms * 1e6));
  }
  return 0;
}
