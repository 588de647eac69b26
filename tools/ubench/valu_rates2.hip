// second round: which integer ops issue faster than the ~4.4-cycle VOP3 class on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 4096, UNROLL = 8;
#define KERNEL_32(NAME, ASM)                                                             \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {              \
  uint32_t a[UNROLL]; uint32_t b = seed ^ threadIdx.x, c = seed * 3u + threadIdx.x;      \
  _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) a[u] = seed + u + threadIdx.x;      \
  for (int it = 0; it < ITERS; ++it) {                                                   \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u)                                   \
      asm volatile(ASM : "+v"(a[u]) : "v"(b), "v"(c));                                   \
  }                                                                                      \
  uint32_t r = 0; _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) r ^= a[u];          \
  if (r == 0x12345678u) out[0] = r;                                                      \
}
KERNEL_32(k_add_u32,       "v_add_u32 %0, %0, %1")
KERNEL_32(k_add_u32_b,     "v_add_u32 %0, %1, %0")
KERNEL_32(k_add_u32_3,     "v_add_u32 %0, %1, %2")
KERNEL_32(k_sub_u32,       "v_sub_u32 %0, %0, %1")
KERNEL_32(k_min_u32,       "v_min_u32 %0, %0, %1")
KERNEL_32(k_max_u32,       "v_max_u32 %0, %0, %1")
KERNEL_32(k_and_b32,       "v_and_b32 %0, %0, %1")
KERNEL_32(k_or_b32,        "v_or_b32 %0, %0, %1")
KERNEL_32(k_xor_b32,       "v_xor_b32 %0, %0, %1")
KERNEL_32(k_lshlrev,       "v_lshlrev_b32 %0, 3, %0")
KERNEL_32(k_mov,           "v_mov_b32 %0, %1")
KERNEL_32(k_min_i32,       "v_min_i32 %0, %0, %1")
KERNEL_32(k_min_u16,       "v_min_u16 %0, %0, %1")
KERNEL_32(k_add_u16,       "v_add_u16 %0, %0, %1")
KERNEL_32(k_min_f32,       "v_min_f32 %0, %0, %1")
KERNEL_32(k_add_f32,       "v_add_f32 %0, %0, %1")
KERNEL_32(k_fma_f32,       "v_fma_f32 %0, %1, %2, %0")
KERNEL_32(k_pk_add_u16,    "v_pk_add_u16 %0, %0, %1")
KERNEL_32(k_add3,          "v_add3_u32 %0, %0, %1, %2")
KERNEL_32(k_min3,          "v_min3_u32 %0, %0, %1, %2")
KERNEL_32(k_add_co,        "v_add_co_u32 %0, vcc, %0, %1")
KERNEL_32(k_cndmask_s,     "v_cndmask_b32 %0, %0, %1, s[10:11]")
KERNEL_32(k_mad_u32_u16,   "v_mad_u32_u16 %0, %1, %2, %0")
KERNEL_32(k_mad_u32_u16s,  "v_mad_u32_u16 %0, %1, s12, %0")
KERNEL_32(k_lshl_add,      "v_lshl_add_u32 %0, %1, 10, %0")
KERNEL_32(k_lshl_or,       "v_lshl_or_b32 %0, %1, 10, %0")
KERNEL_32(k_and_or,        "v_and_or_b32 %0, %1, %2, %0")
KERNEL_32(k_bfi,           "v_bfi_b32 %0, %1, %2, %0")
KERNEL_32(k_permlane32,    "v_permlane32_swap_b32 %0, %1")
KERNEL_32(k_permlane16,    "v_permlane16_swap_b32 %0, %1")
KERNEL_32(k_min_dpp_quad,  "v_min_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL_32(k_min_dpp_bank,  "v_min_u32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3")
KERNEL_32(k_readlane,      "v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0")
#define KERNEL_64(NAME, ASM)                                                             \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {              \
  uint64_t a[UNROLL]; uint64_t b = (uint64_t)(seed ^ threadIdx.x) * 0x9E3779B97F4A7C15ull;\
  uint32_t c = seed * 3u + threadIdx.x;                                                  \
  _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) a[u] = seed + u + threadIdx.x;      \
  for (int it = 0; it < ITERS; ++it) {                                                   \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u)                                   \
      asm volatile(ASM : "+v"(a[u]) : "v"(b), "v"(c));                                   \
  }                                                                                      \
  uint64_t r = 0; _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) r ^= a[u];          \
  if (r == 0x12345678u) out[0] = (uint32_t)r;                                            \
}
KERNEL_64(k_lshl_add_u64,  "v_lshl_add_u64 %0, %1, 0, %0")
KERNEL_64(k_qsad,          "v_qsad_pk_u16_u8 %0, %1, %2, %0")
KERNEL_64(k_qsad_d,        "v_qsad_pk_u16_u8 %0, %0, %2, %1")
KERNEL_64(k_pk_mul_lo,     "v_lshl_add_u64 %0, %1, 0, %0")
struct Bench { const char* name; void (*fn)(uint32_t*, uint32_t); };
static double run(void (*fn)(uint32_t*, uint32_t), int blocks, uint32_t* d_out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d_out, 12345u); CK(hipDeviceSynchronize());
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d_out, 12345u + rep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  return best;
}
// in-kernel clock: s_memtime (shader cycles) vs s_memrealtime (100 MHz) around a VALU loop
__global__ void __launch_bounds__(256) k_clock(unsigned long long* out, uint32_t seed) {
  uint32_t a[8]; for (int u = 0; u < 8; ++u) a[u] = seed + u + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < 200000; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(seed), "v"(seed));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t r = 0; for (int u = 0; u < 8; ++u) r ^= a[u];
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = (r1 - r0) + (r == 0x1234567 ? 1 : 0); }
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  uint32_t* d_out; CK(hipMalloc(&d_out, 64));
  {
    unsigned long long* d; CK(hipMalloc(&d, 16 * 2048)); std::vector<unsigned long long> h(2 * 1024);
    hipLaunchKernelGGL(k_clock, dim3(1024), dim3(256), 0, 0, d, 7u); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, 16 * 1024, hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < 1024; ++i) s += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;
    printf("in-kernel clock under an all-CU v_add3_u32 loop (4 waves/CU x4 blocks): %.1f MHz ; cycles per add3 per wave: %.2f\n", s / 1024, (double)h[0] / (200000.0 * 8));
  }
  std::vector<Bench> benches = {
    {"v_add_u32 d,d,b", k_add_u32}, {"v_add_u32 d,b,d", k_add_u32_b}, {"v_add_u32 d,b,c (no dep)", k_add_u32_3}, {"v_sub_u32", k_sub_u32},
    {"v_min_u32", k_min_u32}, {"v_max_u32", k_max_u32}, {"v_and_b32", k_and_b32}, {"v_or_b32", k_or_b32}, {"v_xor_b32", k_xor_b32},
    {"v_lshlrev_b32", k_lshlrev}, {"v_mov_b32", k_mov}, {"v_min_i32", k_min_i32}, {"v_min_u16", k_min_u16}, {"v_add_u16", k_add_u16},
    {"v_min_f32", k_min_f32}, {"v_add_f32", k_add_f32}, {"v_fma_f32", k_fma_f32}, {"v_pk_add_u16", k_pk_add_u16}, {"v_add3_u32", k_add3},
    {"v_min3_u32", k_min3}, {"v_add_co_u32", k_add_co}, {"v_cndmask_b32 (sgpr mask)", k_cndmask_s}, {"v_mad_u32_u16 vgpr", k_mad_u32_u16},
    {"v_mad_u32_u16 literal 1024", k_mad_u32_u16s}, {"v_lshl_add_u32", k_lshl_add}, {"v_lshl_or_b32", k_lshl_or}, {"v_and_or_b32", k_and_or},
    {"v_bfi_b32", k_bfi}, {"v_permlane32_swap", k_permlane32}, {"v_permlane16_swap", k_permlane16}, {"v_min_u32_dpp quad_perm", k_min_dpp_quad},
    {"v_min_u32_dpp row_ror:8 bank_mask:3", k_min_dpp_bank}, {"v_readlane+v_add (2 ops)", k_readlane},
    {"v_lshl_add_u64", k_lshl_add_u64}, {"v_qsad_pk_u16_u8 acc chain", k_qsad}, {"v_qsad_pk_u16_u8 src0 chain", k_qsad_d},
  };
  printf("%-42s %10s %10s %10s  (wave-instr per ns per CU)\n", "instr", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
  for (auto& b : benches) {
    printf("%-42s", b.name);
    for (int wps : {1, 2, 4}) {
      double ms = run(b.fn, p.multiProcessorCount * wps, d_out);
      printf(" %10.4f", (double)ITERS * UNROLL * 4.0 * wps / (ms * 1e6));
    }
    printf("\n");
  }
  return 0;
}
