// What the LITERAL reading of BASELINE.json's north star costs on gfx950: "one wavefront per candidate MV reducing 4x4 -> 64x64 PU SADs with
// wavefront shuffles and v_sad_u8" -- the reference's layout (cl/sad.cl: a work-item per 4x4 block, all of them on ONE candidate).
// DESIGN.md 4.1 chose lane = 4 adjacent candidates instead (every sum lane-local, only the arg-min crosses lanes) on a count of
// operations; this program measures the other side of that count instead of estimating it.
//
// Layout here: one workgroup (4 waves) per CTU, the window in LDS as in the engine, each wave takes every fourth candidate; lane = one
// 8x8 block of the CTU (its 64 current samples live in 16 VGPRs for the whole search).  Per candidate a lane reads 8 window rows of 8
// samples (unaligned: three dwords + two v_alignbyte per row), forms its four 4x4 SADs with 16 v_sad_u8, and the wave builds the sums of
// the squares and halves at 8, 16, 32 and 64 (2Nx2N, 2NxN, Nx2N: the 425 slots of the AMP-OFF table -- the eight AMP parts per CU are
// left out, which flatters this layout) with the cheapest cross-lane move the hardware has for each distance: quad_perm / row_ror DPP for
// lane xor 1, 2, 8, ds_swizzle for xor 4, v_permlane16/32_swap for xor 16 / 32.  Running minima: one 32-bit key per (lane, sum) --
// (sum << 10) + ((mv cost << 10) | candidate & 1023), the engine's own trick, flushed to (cost, candidate) pairs every 1024 candidates.
// Every lane keeps all 14 minima of its position in the tree (the duplicates inside a CU are the price of uniform code).
// It is a LOWER bound of that design's work: no FEN sums, no AMP, no result decode, candidates in raster order without clipping.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o northstar_layout northstar_layout.hip ; run: ./northstar_layout [width height]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int SR = 64, W = 2 * SR + 1, PDW = 49, ROWS = W + 63;   // 129 x 129 candidates, window rows of 196 bytes (49 dwords, odd: conflict-free)
constexpr int NSUM = 14;                                            // per lane: 8x8, 8x4 x2, 4x8 x2; then (square, 2NxN half, Nx2N half) at 16, 32, 64

__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t dpp_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ uint32_t dpp_xor8(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); }  // row_ror:8
__device__ __forceinline__ uint32_t swz_xor4(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (4 << 10) | 0x1f); }           // bit-mask mode, xor 4
__device__ __forceinline__ uint32_t xor16(uint32_t v) {
  uint32_t a = v, b = v;
  asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // odd rows of a <-> even rows of b
  return (threadIdx.x & 16) ? a : b;
}
__device__ __forceinline__ uint32_t xor32(uint32_t v) {
  uint32_t a = v, b = v;
  asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // upper half of a <-> lower half of b
  return (threadIdx.x & 32) ? a : b;
}

__global__ void __launch_bounds__(256) northstar_kernel(const uint8_t* __restrict__ cur, const uint8_t* __restrict__ ref, int pitch, int ctus_x,
                                                         uint32_t lambda_q16, uint32_t* __restrict__ out_cost, uint32_t* __restrict__ out_cand) {
  __shared__ uint32_t win[ROWS * PDW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ctu = blockIdx.x, cx0 = (ctu % ctus_x) * 64, cy0 = (ctu / ctus_x) * 64;
  // window: ref rows cy0 .. cy0 + ROWS - 1, bytes cx0 .. cx0 + 195 of a plane whose origin already sits (SR, SR) inside the padding
  for (int i = tid; i < ROWS * PDW; i += 256) {
    const int r = i / PDW, k = i - r * PDW;
    win[i] = *(const uint32_t*)(ref + (long)(cy0 + r) * pitch + cx0 + 4 * k);
  }
  // this lane's 8x8 current block: 8 rows x 2 dwords
  const int bx = lane & 7, by = lane >> 3;
  uint32_t c_lo[8], c_hi[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint2 v = *(const uint2*)(cur + (long)(cy0 + SR + 8 * by + r) * pitch + cx0 + SR + 8 * bx);
    c_lo[r] = v.x; c_hi[r] = v.y;
  }
  __syncthreads();
  uint32_t best[NSUM];            // running 32-bit keys of the current 1024-candidate chunk
  uint32_t best_cost[NSUM], best_cand[NSUM];
#pragma unroll
  for (int s = 0; s < NSUM; ++s) { best[s] = 0xffffffffu; best_cost[s] = 0xffffffffu; best_cand[s] = 0; }
  auto flush = [&](int chunk) {
#pragma unroll
    for (int s = 0; s < NSUM; ++s) {
      const uint32_t cost = best[s] >> 10, cand = (uint32_t)chunk * 1024u + (best[s] & 1023u);
      if (best[s] != 0xffffffffu && cost < best_cost[s]) { best_cost[s] = cost; best_cand[s] = cand; }
      best[s] = 0xffffffffu;
    }
  };
  int chunk = 0;
  for (int c = wave; c < W * W; c += 4) {   // wave-uniform
    if ((c >> 10) != chunk) { flush(chunk); chunk = c >> 10; }
    const int my = c / W, mx = c - my * W;                    // candidate (mx, my) relative to the window's top-left
    // MV cost (HM: lambda * (bits(x) + bits(y)) >> 16), uniform over the wave
    const int vx = (mx - SR) << 2, vy = (my - SR) << 2;
    const uint32_t tx = vx <= 0 ? ((uint32_t)(-vx) << 1) + 1u : (uint32_t)vx << 1, ty = vy <= 0 ? ((uint32_t)(-vy) << 1) + 1u : (uint32_t)vy << 1;
    const uint32_t bits = 2u * (31u - (uint32_t)__builtin_clz(tx)) + 1u + 2u * (31u - (uint32_t)__builtin_clz(ty)) + 1u;
    const uint32_t cc = (((lambda_q16 * bits) >> 16) << 10) | ((uint32_t)c & 1023u);
    // the lane's 8 x 8 reference samples at (my + 8 by .. + 7, mx + 8 bx .. + 7)
    const int col = mx + 8 * bx;
    const uint32_t* row = win + (my + 8 * by) * PDW + (col >> 2);
    const uint32_t sh = (uint32_t)col & 3u;
    uint32_t tl = 0, tr = 0, bl = 0, br = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint32_t d0 = row[r * PDW], d1 = row[r * PDW + 1], d2 = row[r * PDW + 2];
      const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh), hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
      if (r < 4) { tl = __builtin_amdgcn_sad_u8(lo, c_lo[r], tl); tr = __builtin_amdgcn_sad_u8(hi, c_hi[r], tr); }
      else { bl = __builtin_amdgcn_sad_u8(lo, c_lo[r], bl); br = __builtin_amdgcn_sad_u8(hi, c_hi[r], br); }
    }
    uint32_t sum[NSUM];
    sum[1] = tl + tr; sum[2] = bl + br; sum[3] = tl + bl; sum[4] = tr + br; sum[0] = sum[1] + sum[2];   // 8x4 top/bottom, 4x8 left/right, 8x8
    const uint32_t h16 = sum[0] + dpp_xor1(sum[0]);          // 16x8 half (top for even by, bottom for odd)
    const uint32_t v16 = sum[0] + dpp_xor8(sum[0]);          // 8x16 half
    const uint32_t s16 = h16 + dpp_xor8(h16);
    sum[5] = s16; sum[6] = h16; sum[7] = v16;
    const uint32_t h32 = s16 + dpp_xor2(s16);                // 32x16 half
    const uint32_t v32 = s16 + xor16(s16);                   // 16x32 half
    const uint32_t s32 = h32 + xor16(h32);
    sum[8] = s32; sum[9] = h32; sum[10] = v32;
    const uint32_t h64 = s32 + swz_xor4(s32);                // 64x32 half
    const uint32_t v64 = s32 + xor32(s32);                   // 32x64 half
    const uint32_t s64 = h64 + xor32(h64);
    sum[11] = s64; sum[12] = h64; sum[13] = v64;
#pragma unroll
    for (int s = 0; s < NSUM; ++s) best[s] = min(best[s], (sum[s] << 10) + cc);
  }
  flush(chunk);
  // merge the four waves' minima through LDS (cost, candidate) -- raster order on ties
  __syncthreads();
  unsigned long long* m = (unsigned long long*)win;   // [64 lanes][NSUM]
  for (int i = tid; i < 64 * NSUM; i += 256) m[i] = ~0ull;
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NSUM; ++s) atomicMin(&m[lane * NSUM + s], (unsigned long long)best_cost[s] << 32 | best_cand[s]);
  __syncthreads();
  for (int i = tid; i < 64 * NSUM; i += 256) {
    out_cost[(long)ctu * 64 * NSUM + i] = (uint32_t)(m[i] >> 32);
    out_cand[(long)ctu * 64 * NSUM + i] = (uint32_t)m[i];
  }
}

int main(int argc, char** argv) {
  const int w = argc > 2 ? atoi(argv[1]) : 3840, h = argc > 2 ? atoi(argv[2]) : 2160;
  const int ctus_x = w / 64, ctus_y = h / 64, n_ctu = ctus_x * ctus_y;   // whole CTUs only
  const int pitch = ((w + 2 * SR + 64 + 255) & ~255), rows = h + 2 * SR + 64;
  std::vector<uint8_t> hc((size_t)pitch * rows), hr((size_t)pitch * rows);
  srand(7);
  for (auto& v : hr) v = (uint8_t)(rand() & 255);
  // current = reference displaced by (5, -3) plus noise: arg-mins are non-trivial
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < pitch; ++x) {
      const int sy = y - 3 < 0 ? 0 : (y - 3 >= rows ? rows - 1 : y - 3), sx = x + 5 >= pitch ? pitch - 1 : x + 5;
      int v = hr[(size_t)sy * pitch + sx] + (rand() % 5) - 2;
      hc[(size_t)y * pitch + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  uint8_t *dc, *dr; uint32_t *d_cost, *d_cand;
  CK(hipMalloc(&dc, hc.size())); CK(hipMalloc(&dr, hr.size()));
  CK(hipMalloc(&d_cost, (size_t)n_ctu * 64 * NSUM * 4)); CK(hipMalloc(&d_cand, (size_t)n_ctu * 64 * NSUM * 4));
  CK(hipMemcpy(dc, hc.data(), hc.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr.data(), hr.size(), hipMemcpyHostToDevice));
  const uint32_t lq = 498676;   // floor(65536 * sqrt(57.9))
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(northstar_kernel, dim3(n_ctu), dim3(256), 0, 0, dc, dr, pitch, ctus_x, lq, d_cost, d_cand);
  CK(hipDeviceSynchronize());
  const int reps = 5;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(northstar_kernel, dim3(n_ctu), dim3(256), 0, 0, dc, dr, pitch, ctus_x, lq, d_cost, d_cand);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  // check the 64x64 sum's arg-min of CTU 0 against a scalar search on the host
  std::vector<uint32_t> cost((size_t)64 * NSUM), cand((size_t)64 * NSUM);
  CK(hipMemcpy(cost.data(), d_cost, cost.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cand.data(), d_cand, cand.size() * 4, hipMemcpyDeviceToHost));
  uint32_t best = 0xffffffffu, best_c = 0;
  for (int c = 0; c < W * W; ++c) {
    const int my = c / W, mx = c % W;
    uint32_t sad = 0;
    for (int y = 0; y < 64; ++y)
      for (int x = 0; x < 64; ++x) sad += (uint32_t)abs((int)hc[(size_t)(SR + y) * pitch + SR + x] - (int)hr[(size_t)(my + y) * pitch + mx + x]);
    const int vx = (mx - SR) << 2, vy = (my - SR) << 2;
    const uint32_t tx = vx <= 0 ? ((uint32_t)(-vx) << 1) + 1u : (uint32_t)vx << 1, ty = vy <= 0 ? ((uint32_t)(-vy) << 1) + 1u : (uint32_t)vy << 1;
    const uint32_t bits = 2u * (31u - (uint32_t)__builtin_clz(tx)) + 1u + 2u * (31u - (uint32_t)__builtin_clz(ty)) + 1u;
    const uint32_t k = sad + ((lq * bits) >> 16);
    if (k < best) { best = k; best_c = (uint32_t)c; }
  }
  const bool ok = cost[11] == best && cand[11] == best_c;
  const double cands = (double)n_ctu * W * W;
  printf("{\"layout\": \"one wavefront per candidate, lane = 8x8 block, 425 AMP-off slots (squares and halves), SR 64\", \"size\": \"%dx%d\", \"ctus\": %d, "
         "\"ms_per_picture_pair\": %.3f, \"gsad_4x4_per_s\": %.1f, \"ctu0_64x64_argmin_matches_host\": %s, \"ctu0_64x64\": [%u, %u]}\n",
         w, h, n_ctu, ms, cands * 256 / (ms * 1e-3) / 1e9, ok ? "true" : "false", cost[11], cand[11]);
  return ok ? 0 : 1;
}
