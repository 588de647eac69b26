// Micro-benchmark: issue rate of the packed-byte SAD family and of the VALU ops the
// reduction tree needs, on gfx950.  Also checks instruction semantics (byte order of the
// sliding window in v_qsad_pk_u16_u8, u16 wrap, the "masked" variants).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;

// each kernel: UNROLL independent dependency chains, ITERS iterations.
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

KERNEL_32(k_add_u32,      "v_add_u32 %0, %0, %1")
KERNEL_32(k_add3_u32,     "v_add3_u32 %0, %0, %1, %2")
KERNEL_32(k_min_u32,      "v_min_u32 %0, %0, %1")
KERNEL_32(k_min3_u32,     "v_min3_u32 %0, %0, %1, %2")
KERNEL_32(k_sad_u8,       "v_sad_u8 %0, %1, %2, %0")
KERNEL_32(k_sad_u16,      "v_sad_u16 %0, %1, %2, %0")
KERNEL_32(k_sad_hi_u8,    "v_sad_hi_u8 %0, %1, %2, %0")
KERNEL_32(k_msad_u8,      "v_msad_u8 %0, %1, %2, %0")
KERNEL_32(k_mad_u32_u16,  "v_mad_u32_u16 %0, %1, %2, %0")
KERNEL_32(k_mad_u32_u16h, "v_mad_u32_u16 %0, %1, %2, %0 op_sel:[1,0,0,0]")
KERNEL_32(k_mad_u32_u24,  "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL_32(k_pk_add_u16,   "v_pk_add_u16 %0, %0, %1")
KERNEL_32(k_pk_sub_u16,   "v_pk_sub_u16 %0, %0, %1")
KERNEL_32(k_pk_min_u16,   "v_pk_min_u16 %0, %0, %1")
KERNEL_32(k_lshl_add_u32, "v_lshl_add_u32 %0, %1, 8, %0")
KERNEL_32(k_alignbyte,    "v_alignbyte_b32 %0, %0, %1, %2")
KERNEL_32(k_perm_b32,     "v_perm_b32 %0, %0, %1, %2")
KERNEL_32(k_cndmask,      "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_32(k_mov_dpp,      "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL_32(k_min_dpp,      "v_min_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL_32(k_bfe_u32,      "v_bfe_u32 %0, %0, 16, 16")

// 64-bit destination / accumulator forms
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
KERNEL_64(k_qsad_pk,      "v_qsad_pk_u16_u8 %0, %1, %2, %0")
KERNEL_64(k_mqsad_pk,     "v_mqsad_pk_u16_u8 %0, %1, %2, %0")
KERNEL_64(k_pk_mov_b32,   "v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]")
KERNEL_64(k_mov_b64,      "v_mov_b64 %0, %1")

// qsad with an SGPR as the 32-bit (current-block) operand
__global__ void __launch_bounds__(256) k_qsad_pk_sgpr(uint32_t* out, uint32_t seed) {
  uint64_t a[UNROLL]; uint64_t b = (uint64_t)(seed ^ threadIdx.x) * 0x9E3779B97F4A7C15ull;
  uint32_t c = seed * 3u;
  #pragma unroll
  for (int u = 0; u < UNROLL; ++u) a[u] = seed + u + threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
    #pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b), "s"(c));
  }
  uint64_t r = 0;
  #pragma unroll
  for (int u = 0; u < UNROLL; ++u) r ^= a[u];
  if (r == 0x12345678u) out[0] = (uint32_t)r;
}

// 128-bit accumulator form
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_mqsad_u32(uint32_t* out, uint32_t seed) {
  u32x4 a[4]; uint64_t b = (uint64_t)(seed ^ threadIdx.x) * 0x9E3779B97F4A7C15ull;
  uint32_t c = seed * 3u + threadIdx.x;
  #pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = (u32x4){seed + u, 1, 2, 3};
  for (int it = 0; it < ITERS; ++it) {
    #pragma unroll
    for (int r2 = 0; r2 < 2; ++r2)
    #pragma unroll
    for (int u = 0; u < 4; ++u)
      asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b), "v"(c));
  }
  uint32_t r = 0;
  #pragma unroll
  for (int u = 0; u < 4; ++u) r ^= a[u].x ^ a[u].y ^ a[u].z ^ a[u].w;
  if (r == 0x12345678u) out[0] = r;
}

// mixed stream: 1 qsad : 3 add3 (does qsad overlap with plain VALU from the same wave / other waves?)
__global__ void __launch_bounds__(256) k_mix_qsad_add3(uint32_t* out, uint32_t seed) {
  uint64_t a[4]; uint32_t d[12]; uint64_t b = (uint64_t)(seed ^ threadIdx.x) * 0x9E3779B97F4A7C15ull;
  uint32_t c = seed * 3u + threadIdx.x;
  #pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = seed + u + threadIdx.x;
  #pragma unroll
  for (int u = 0; u < 12; ++u) d[u] = seed + u;
  for (int it = 0; it < ITERS / 2; ++it) {
    #pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b), "v"(c));
      asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(d[3*u+0]) : "v"((uint32_t)b), "v"(c));
      asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(d[3*u+1]) : "v"((uint32_t)b), "v"(c));
      asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(d[3*u+2]) : "v"((uint32_t)b), "v"(c));
    }
  }
  uint64_t r = 0;
  #pragma unroll
  for (int u = 0; u < 4; ++u) r ^= a[u];
  #pragma unroll
  for (int u = 0; u < 12; ++u) r ^= d[u];
  if (r == 0x12345678u) out[0] = (uint32_t)r;
}

// LDS read rates in the access shapes the ME kernel uses: lane stride 4 B (b32), 8 B (b64), 16 B (b128)
template <int W> __global__ void __launch_bounds__(256) k_lds_read(uint32_t* out, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[12288];
  for (int i = threadIdx.x; i < 12288; i += 256) lds[i] = i * seed;
  __syncthreads();
  uint32_t acc = 0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int it = 0; it < ITERS; ++it) {
    int base = (wv * 2048 + ((it * 52) & 1023));
    if (W == 1) {
      #pragma unroll
      for (int k = 0; k < 8; ++k) acc += lds[base + lane + k * 48];
    } else if (W == 2) {
      #pragma unroll
      for (int k = 0; k < 8; ++k) { uint2 v = *(const uint2*)&lds[(base & ~1) + lane * 2 + k * 48]; acc += v.x ^ v.y; }
    } else {
      #pragma unroll
      for (int k = 0; k < 8; ++k) { uint4 v = *(const uint4*)&lds[(base & ~3) + lane * 4 + k * 48]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

struct Bench { const char* name; void (*fn)(uint32_t*, uint32_t); double ops_per_thread; };

static double run(void (*fn)(uint32_t*, uint32_t), int blocks, uint32_t* d_out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
  CK(hipDeviceSynchronize());
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d_out, 12345u + rep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

// ---- semantics ----
__global__ void k_sem(uint64_t* out, uint64_t src0, uint32_t src1, uint64_t acc) {
  uint64_t q = acc, mq = acc; uint32_t s = (uint32_t)acc, ms = (uint32_t)acc, s16 = (uint32_t)acc;
  asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(q) : "v"(src0), "v"(src1));
  asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(mq) : "v"(src0), "v"(src1));
  asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(s) : "v"((uint32_t)src0), "v"(src1));
  asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(ms) : "v"((uint32_t)src0), "v"(src1));
  asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(s16) : "v"((uint32_t)src0), "v"(src1));
  u32x4 m4 = (u32x4){(uint32_t)acc, (uint32_t)acc, (uint32_t)acc, (uint32_t)acc};
  asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(m4) : "v"(src0), "v"(src1));
  uint32_t madh = 7;
  asm volatile("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(madh) : "v"((uint32_t)src0), "v"(2048u));
  out[0] = q; out[1] = mq; out[2] = s; out[3] = ms; out[4] = s16;
  out[5] = ((uint64_t)m4.y << 32) | m4.x; out[6] = ((uint64_t)m4.w << 32) | m4.z; out[7] = madh;
}

static uint32_t sad4(uint64_t s0, int sh, uint32_t s1, bool masked, bool mask_on_s1 = true) {
  uint32_t r = 0;
  for (int i = 0; i < 4; ++i) {
    int a = (s0 >> (8 * (sh + i))) & 0xff, b = (s1 >> (8 * i)) & 0xff;
    if (masked && (mask_on_s1 ? b == 0 : a == 0)) continue;
    r += abs(a - b);
  }
  return r;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s  CUs %d  clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  uint32_t* d_out; CK(hipMalloc(&d_out, 64));

  // semantics
  {
    uint64_t* d; CK(hipMalloc(&d, 64)); uint64_t h[8];
    struct { uint64_t s0; uint32_t s1; uint64_t acc; } cases[] = {
      {0x0807060504030201ull, 0x04030201u, 0},
      {0xf0e0d0c0b0a09080ull, 0x00ff00ffu, 0x0001000200030004ull},
      {0x00000000000000ffull, 0x00000000u, 0},
      {0xffffffffffffffffull, 0x00000000u, 0xff00ff00ff00ff00ull},   // u16 accumulate overflow: wrap or saturate?
      {0x1122334455667788ull, 0x80004000u, 5},
    };
    for (auto& c : cases) {
      hipLaunchKernelGGL(k_sem, dim3(1), dim3(1), 0, 0, d, c.s0, c.s1, c.acc);
      CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
      printf("SEM s0=%016llx s1=%08x acc=%016llx\n", (unsigned long long)c.s0, c.s1, (unsigned long long)c.acc);
      printf("  qsad_pk   = %016llx   expect(unmasked,wrap) =", (unsigned long long)h[0]);
      for (int j = 3; j >= 0; --j) printf(" %04x", (unsigned)((sad4(c.s0, j, c.s1, false) + ((c.acc >> (16 * j)) & 0xffff)) & 0xffff));
      printf("\n  mqsad_pk  = %016llx   expect(mask s1==0)    =", (unsigned long long)h[1]);
      for (int j = 3; j >= 0; --j) printf(" %04x", (unsigned)((sad4(c.s0, j, c.s1, true) + ((c.acc >> (16 * j)) & 0xffff)) & 0xffff));
      printf("\n  sad_u8    = %08llx  expect %08x\n", (unsigned long long)h[2], sad4(c.s0, 0, c.s1, false) + (uint32_t)c.acc);
      printf("  msad_u8   = %08llx  expect(mask s1==0) %08x  (mask s0==0) %08x\n", (unsigned long long)h[3],
             sad4(c.s0, 0, c.s1, true) + (uint32_t)c.acc, sad4(c.s0, 0, c.s1, true, false) + (uint32_t)c.acc);
      printf("  sad_u16   = %08llx\n", (unsigned long long)h[4]);
      printf("  mqsad_u32 = %016llx %016llx\n", (unsigned long long)h[6], (unsigned long long)h[5]);
      printf("  mad_u32_u16 op_sel hi(s0.lo32)*2048+7 = %llu expect %u\n", (unsigned long long)h[7], (((uint32_t)c.s0 >> 16) & 0xffff) * 2048u + 7u);
    }
    CK(hipFree(d));
  }

  std::vector<Bench> benches = {
    {"v_add_u32", k_add_u32, (double)ITERS * UNROLL}, {"v_add3_u32", k_add3_u32, (double)ITERS * UNROLL},
    {"v_min_u32", k_min_u32, (double)ITERS * UNROLL}, {"v_min3_u32", k_min3_u32, (double)ITERS * UNROLL},
    {"v_sad_u8", k_sad_u8, (double)ITERS * UNROLL}, {"v_sad_u16", k_sad_u16, (double)ITERS * UNROLL},
    {"v_sad_hi_u8", k_sad_hi_u8, (double)ITERS * UNROLL}, {"v_msad_u8", k_msad_u8, (double)ITERS * UNROLL},
    {"v_qsad_pk_u16_u8", k_qsad_pk, (double)ITERS * UNROLL}, {"v_qsad_pk_u16_u8(sgpr)", k_qsad_pk_sgpr, (double)ITERS * UNROLL},
    {"v_mqsad_pk_u16_u8", k_mqsad_pk, (double)ITERS * UNROLL}, {"v_mqsad_u32_u8", k_mqsad_u32, (double)ITERS * 8},
    {"v_mad_u32_u16", k_mad_u32_u16, (double)ITERS * UNROLL}, {"v_mad_u32_u16(op_sel hi)", k_mad_u32_u16h, (double)ITERS * UNROLL},
    {"v_mad_u32_u24", k_mad_u32_u24, (double)ITERS * UNROLL},
    {"v_pk_add_u16", k_pk_add_u16, (double)ITERS * UNROLL}, {"v_pk_sub_u16", k_pk_sub_u16, (double)ITERS * UNROLL},
    {"v_pk_min_u16", k_pk_min_u16, (double)ITERS * UNROLL}, {"v_lshl_add_u32", k_lshl_add_u32, (double)ITERS * UNROLL},
    {"v_alignbyte_b32", k_alignbyte, (double)ITERS * UNROLL}, {"v_perm_b32", k_perm_b32, (double)ITERS * UNROLL},
    {"v_cndmask_b32", k_cndmask, (double)ITERS * UNROLL}, {"v_mov_b32_dpp", k_mov_dpp, (double)ITERS * UNROLL},
    {"v_min_u32_dpp", k_min_dpp, (double)ITERS * UNROLL}, {"v_bfe_u32", k_bfe_u32, (double)ITERS * UNROLL},
    {"v_pk_mov_b32", k_pk_mov_b32, (double)ITERS * UNROLL}, {"v_mov_b64", k_mov_b64, (double)ITERS * UNROLL},
    {"mix 1 qsad : 3 add3 (ops counted: 4)", k_mix_qsad_add3, (double)(ITERS / 2) * 16},
    {"ds_read_b32 x8 (lane stride 4B)", k_lds_read<1>, (double)ITERS * 8},
    {"ds_read_b64 x8 (lane stride 8B)", k_lds_read<2>, (double)ITERS * 8},
    {"ds_read_b128 x8 (lane stride 16B)", k_lds_read<4>, (double)ITERS * 8},
  };
  const int cus = p.multiProcessorCount;
  printf("%-40s %12s %12s %12s   (wave-instr per ns per CU; x/2.4 = per clk at 2.4GHz)\n", "instr", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
  for (auto& b : benches) {
    printf("%-40s", b.name);
    for (int wps : {1, 2, 4}) {
      int blocks = cus * wps;                       // 256 threads = 4 waves = 1 wave per SIMD per block
      double ms = run(b.fn, blocks, d_out);
      double wave_instr = b.ops_per_thread * 4.0 * wps;   // per CU
      double per_ns = wave_instr / (ms * 1e6);
      printf(" %12.4f", per_ns);
    }
    printf("\n");
  }
  return 0;
}
