// third round: ops the fractional-refinement kernel is made of (8-tap filters, shifts/clamps, Hadamard) on gfx950
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates3 valu_rates3.hip
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
KERNEL_32(k_add_u32,     "v_add_u32 %0, %0, %1")
KERNEL_32(k_dot4c,       "v_dot4c_i32_i8 %0, %1, %2")
KERNEL_32(k_dot2c,       "v_dot2c_i32_i16 %0, %1, %2")
KERNEL_32(k_dot4,        "v_dot4_i32_i8 %0, %1, %2, %0")
KERNEL_32(k_dot2,        "v_dot2_i32_i16 %0, %1, %2, %0")
KERNEL_32(k_dot4u,       "v_dot4_u32_u8 %0, %1, %2, %0")
KERNEL_32(k_mad_i24,     "v_mad_i32_i24 %0, %1, %2, %0")
KERNEL_32(k_mul_i24,     "v_mul_i32_i24 %0, %0, %1")
KERNEL_32(k_mul_lo,      "v_mul_lo_u32 %0, %0, %1")
KERNEL_32(k_mad_i16,     "v_mad_i32_i16 %0, %1, %2, %0")
KERNEL_32(k_fmac_f32,    "v_fmac_f32 %0, %1, %2")
KERNEL_32(k_med3_i32,    "v_med3_i32 %0, %0, %1, %2")
KERNEL_32(k_ashr,        "v_ashrrev_i32 %0, 3, %0")
KERNEL_32(k_max_i32,     "v_max_i32 %0, %0, %1")
KERNEL_32(k_sad_u32,     "v_sad_u32 %0, %1, %2, %0")
KERNEL_32(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL_32(k_add_f32_abs, "v_add_f32 %0, %0, |%1|")
KERNEL_32(k_sub_f32,     "v_sub_f32 %0, %0, %1")
KERNEL_32(k_pk_add_i16,  "v_pk_add_i16 %0, %0, %1")
KERNEL_32(k_pk_mad_i16,  "v_pk_mad_i16 %0, %1, %2, %0")
KERNEL_32(k_perm,        "v_perm_b32 %0, %0, %1, %2")
KERNEL_32(k_and_or,      "v_and_or_b32 %0, %0, %1, %2")
KERNEL_32(k_lshl_or,     "v_lshl_or_b32 %0, %0, 16, %1")
KERNEL_32(k_pack_f16,    "v_pack_b32_f16 %0, %0, %1")
KERNEL_32(k_sub_dpp,     "v_sub_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL_32(k_add_sdwa,    "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
KERNEL_32(k_bfe_i32,     "v_bfe_i32 %0, %0, 16, 16")

#define KERNEL_64(NAME, ASM)                                                             \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {              \
  double a[UNROLL]; double b = (double)(seed ^ threadIdx.x), c = (double)(seed * 3u + threadIdx.x); \
  _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) a[u] = (double)(seed + u + threadIdx.x);      \
  for (int it = 0; it < ITERS; ++it) {                                                   \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u)                                   \
      asm volatile(ASM : "+v"(a[u]) : "v"(b), "v"(c));                                   \
  }                                                                                      \
  double r = 0; _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) r += a[u];            \
  if (r == 0.12345678) out[0] = 1;                                                       \
}
KERNEL_64(k_pk_fma_f32,  "v_pk_fma_f32 %0, %1, %2, %0")
KERNEL_64(k_pk_fma_f32b, "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]")
KERNEL_64(k_pk_add_f32,  "v_pk_add_f32 %0, %0, %1")
KERNEL_64(k_pk_add_f32n, "v_pk_add_f32 %0, %0, %1 op_sel_hi:[0,1] neg_hi:[0,1]")
KERNEL_64(k_pk_mul_f32,  "v_pk_mul_f32 %0, %0, %1")
KERNEL_64(k_pk_mov_b32,  "v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]")
KERNEL_32(k_max_f32,     "v_max_f32 %0, %0, %1")
KERNEL_32(k_min_f32,     "v_min_f32 %0, %0, %1")
KERNEL_32(k_med3_f32,    "v_med3_f32 %0, %0, %1, %2")
KERNEL_32(k_floor_f32,   "v_floor_f32 %0, %0")
KERNEL_32(k_cvt_ub0,     "v_cvt_f32_ubyte1 %0, %0")
KERNEL_32(k_fma_f32,     "v_fma_f32 %0, %1, %2, %0")
KERNEL_32(k_mul_f32,     "v_mul_f32 %0, %0, %1")

struct B { const char* name; void (*fn)(uint32_t*, uint32_t); };
int main() {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
  uint32_t* d_out; CK(hipMalloc(&d_out, 64));
  const int blocks = p.multiProcessorCount * 2;   // 2 x 256 threads per CU = 2 waves per SIMD
  std::vector<B> bs = {{"v_add_u32", k_add_u32}, {"v_dot4c_i32_i8", k_dot4c}, {"v_dot2c_i32_i16", k_dot2c}, {"v_dot4_i32_i8", k_dot4},
    {"v_dot2_i32_i16", k_dot2}, {"v_dot4_u32_u8", k_dot4u}, {"v_mad_i32_i24", k_mad_i24}, {"v_mul_i32_i24", k_mul_i24}, {"v_mul_lo_u32", k_mul_lo},
    {"v_mad_i32_i16", k_mad_i16}, {"v_fmac_f32", k_fmac_f32}, {"v_med3_i32", k_med3_i32}, {"v_ashrrev_i32", k_ashr}, {"v_max_i32", k_max_i32},
    {"v_sad_u32", k_sad_u32}, {"v_cvt_f32_i32", k_cvt_f32_i32}, {"v_add_f32 |abs|", k_add_f32_abs}, {"v_sub_f32", k_sub_f32},
    {"v_pk_add_i16", k_pk_add_i16}, {"v_pk_mad_i16", k_pk_mad_i16}, {"v_perm_b32", k_perm}, {"v_and_or_b32", k_and_or}, {"v_lshl_or_b32", k_lshl_or},
    {"v_pack_b32_f16", k_pack_f16}, {"v_sub_u32_dpp", k_sub_dpp}, {"v_add_u32_sdwa", k_add_sdwa}, {"v_bfe_i32", k_bfe_i32},
    {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_fma_f32 bcast", k_pk_fma_f32b}, {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_add_f32 neg_hi", k_pk_add_f32n},
    {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_mov_b32", k_pk_mov_b32}, {"v_max_f32", k_max_f32}, {"v_min_f32", k_min_f32}, {"v_med3_f32", k_med3_f32},
    {"v_floor_f32", k_floor_f32}, {"v_cvt_f32_ubyte1", k_cvt_ub0}, {"v_fma_f32", k_fma_f32}, {"v_mul_f32", k_mul_f32}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, dev));
  printf("%-20s %10s %14s\n", "op", "ms", "cyc/wave-instr (at 2.2 GHz, 2 waves/SIMD)");
  for (auto& b : bs) {
    hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(256), 0, 0, d_out, 12345u + r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    // per SIMD: 2 waves x ITERS*UNROLL instrs
    const double cyc = ms * 1e-3 * 2.2e9 / (2.0 * ITERS * UNROLL);
    printf("%-20s %10.4f %14.2f\n", b.name, ms, cyc);
  }
  return 0;
}
