// me_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) of the hmme engine.
//
// Hot kernel: me_search_kernel<FEN>.  One workgroup (256 threads = 4 waves) searches one CTU
// against one reference picture:
//   1. the (Wy+63) x (Wx+63) byte reference window is staged into LDS (coalesced dword loads,
//      re-aligned with v_alignbyte so that window column 0 sits at LDS byte 0 of each row);
//   2. waves pull *tasks* (a strip of candidate MVs) from an LDS counter.  In a task every lane
//      owns four horizontally adjacent candidates per iteration and walks the whole 64x64 CTU:
//      v_qsad_pk_u16_u8 leaves (current-block dwords arrive through scalar loads, i.e. as SGPR
//      operands), packed-u16 / linear-key reduction tree, per-lane min, wave butterfly --
//      generated straight-line code, see tools/gen_me_tree.py;
//   3. each wave folds its ten running-minimum registers into a per-CTU LDS table with 64-bit
//      (cost, y, x) keys (ds_min_u64) so that ties resolve in raster order like
//      TEncSearch::xPatternSearch's strict '<' (reference TEncSearch.cpp:3866-3889);
//   4. 593 (mv, sad) results are written out coalesced.
//
// No MFMA: this is integer absolute-difference work.  The kernel is VALU-bound.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hmme {

constexpr int kParts = 593;            // NUM_CTU_PARTS, reference TLibCommon/TypeDef.h:263
constexpr int kPDW = 49;               // LDS window pitch in dwords (196 B >= 129 + 63 + 3)
constexpr int kWinRowsMax = 192;       // (2*64+1) + 63
constexpr int kIdxBits = 10;           // key = cost << 10 | iter(2) | lane(6) | j(2)
constexpr uint32_t kInvCost = 3146751u;  // > any valid cost (<= 1047552 + 65535); + max SAD stays < 2^22
constexpr int kGroups = 10;
// lane-iterations per task (<= 4: 2 iteration bits in the key).  A task ends with the flush of ten running-minimum registers into the
// CTU's LDS table; whole-picture launches run 4 per task (measured on 2160p: 1 / 2 / 4 -> 3 252 / 3 256 / 3 284 GSAD/s,
// profiles/archive/r02f_*), split launches (one CTU dealt to many workgroups, where the number of tasks is the parallelism) 1: the per-CTU
// call at SR 64 goes from 9 to 17 workgroups of one iteration per wave, 0.086 -> 0.065 ms (profiles/archive/r02l_*)
#ifndef ME_ITER_PER_TASK
#define ME_ITER_PER_TASK 4
#endif
#ifndef ME_ITER_PER_TASK_SPLIT
#define ME_ITER_PER_TASK_SPLIT 1
#endif
#ifndef ME_GUIDED_TASKS
#define ME_GUIDED_TASKS 1
#endif
#ifndef ME_FRAC_PRIO   // A/B: 0 = the refinement kernel's waves all at the dispatch priority
#define ME_FRAC_PRIO 1
#endif
#ifndef ME_FAIR_PRIO   // A/B: 0 = every wave at the dispatch priority (the SIMD then favours its oldest wave)
#define ME_FAIR_PRIO 1
#endif
constexpr int kIterPerTask = ME_ITER_PER_TASK;
constexpr int kIterPerTaskSplit = ME_ITER_PER_TASK_SPLIT;
static_assert(ME_ITER_PER_TASK <= 4 && (!ME_GUIDED_TASKS || ME_ITER_PER_TASK == 4), "2 iteration bits in the key; the guided schedule deals 4 / 2 / 1");
constexpr int kThreads = 256;

// one CTU search: everything in integer pels except the quarter-pel predictor
struct MeJob {
  int16_t ctu_x, ctu_y;   // CTU origin in the picture
  int16_t lt_x, lt_y;     // window top-left   (xSetSearchRange, reference TEncSearch.cpp:3814-3830)
  int16_t rb_x, rb_y;     // window bottom-right, inclusive
  int16_t pred_x, pred_y; // AMVP predictor, quarter pels
};
static_assert(sizeof(MeJob) == 16, "MeJob layout");
// CTU origins are multiples of 64, so the low 6 bits of MeJob::ctu_x carry the index of the picture pair the job belongs to: one
// launch searches up to 16 (current, reference) pairs -- several references of one picture (hmme_search_frame_multi: the same
// current plane in every entry) or several pictures of a sequence (hmme_search_pairs_device).  Both sets travel by value.
constexpr int kMaxRefs = 16;
struct RefSet { const uint8_t* base[kMaxRefs]; };

#include "me_slotmap.inc"

// ---- helpers used by the generated tree ------------------------------------------------------
typedef uint16_t u16x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
// LDS reads of the generated tree are volatile: they are issued exactly where the generator put
// them (one 8x8 CU ahead of their use) instead of being hoisted and spilled by the scheduler
typedef __attribute__((address_space(3))) char lds_char_t;
typedef volatile __attribute__((address_space(3))) uint64_t lds_vu64_t;
typedef volatile __attribute__((address_space(3), aligned(4))) uint64_t lds_vu64a4_t;   // -> ds_read2_b32

// packed-u16 x4 sums never carry between halves (every partial SAD here is <= 65 280), so the packed add is
// a plain 64-bit add (one v_lshl_add_u64, ~4.6 cycles) and the packed subtract two 32-bit subtracts
// (v_sub_u32, ~2.7 cycles each) instead of two v_pk_*_u16 (~4.3 each)  [profiles/r01_valu_rates2_ubench.txt]
__device__ __forceinline__ uint64_t me_pkadd(uint64_t a, uint64_t b) { return a + b; }
__device__ __forceinline__ uint64_t me_pksub(uint64_t a, uint64_t b) {
  return ((uint64_t)((uint32_t)(a >> 32) - (uint32_t)(b >> 32)) << 32) | (uint64_t)((uint32_t)a - (uint32_t)b);
}

#define ME_MAXKEY 0xFFFFFFFFu
#define ME_QSAD(pair, cur, acc) __builtin_amdgcn_qsad_pk_u16_u8((pair), (cur), (acc))
// key_j = sad_j * mult + c_j : one v_mad_u32_u16 per candidate (op_sel picks the packed half)
#define ME_KEYS(v, p, mult)                                                                                    \
  uint32_t v##_0, v##_1, v##_2, v##_3;                                                                         \
  asm volatile("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(v##_0) : "v"((uint32_t)(p)), "s"(mult), "v"(c0));                   \
  asm volatile("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(v##_1) : "v"((uint32_t)(p)), "s"(mult), "v"(c1));   \
  asm volatile("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(v##_2) : "v"((uint32_t)((p) >> 32)), "s"(mult), "v"(c2));           \
  asm volatile("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(v##_3) : "v"((uint32_t)((p) >> 32)), "s"(mult), "v"(c3))
// keys are linear in the SAD:  K(a U b) = K(a) + K(b) - c ;  K(a \ b) = K(a) - K(b) + c
#define ME_LIN(v, a, b)                                                                                         \
  uint32_t v##_0, v##_1, v##_2, v##_3;                                                                          \
  asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(v##_0) : "v"(a##_0), "v"(b##_0), "v"(nc0));                              \
  asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(v##_1) : "v"(a##_1), "v"(b##_1), "v"(nc1));                              \
  asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(v##_2) : "v"(a##_2), "v"(b##_2), "v"(nc2));                              \
  asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(v##_3) : "v"(a##_3), "v"(b##_3), "v"(nc3))
// K(a) >= K(b) whenever b's rectangle is inside a's, so |K(a) - K(b)| + c is exact: one v_sad_u32
#define ME_SUB(v, a, b)                                                                                         \
  uint32_t v##_0, v##_1, v##_2, v##_3;                                                                          \
  asm volatile("v_sad_u32 %0, %1, %2, %3" : "=v"(v##_0) : "v"(a##_0), "v"(b##_0), "v"(c0));                                \
  asm volatile("v_sad_u32 %0, %1, %2, %3" : "=v"(v##_1) : "v"(a##_1), "v"(b##_1), "v"(c1));                                \
  asm volatile("v_sad_u32 %0, %1, %2, %3" : "=v"(v##_2) : "v"(a##_2), "v"(b##_2), "v"(c2));                                \
  asm volatile("v_sad_u32 %0, %1, %2, %3" : "=v"(v##_3) : "v"(a##_3), "v"(b##_3), "v"(c3))
// the key arithmetic above, this minimum and the masked merges below are `asm volatile`: not for side effects but to keep
// their mutual order in the ISA equal to the generator's order, which places every masked merge >= 2 of these instructions
// after the ones that produce its inputs (tools/gen_me_tree.py space_merges)
__device__ __forceinline__ uint32_t me_min4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  uint32_t r;
  asm volatile("v_min3_u32 %0, %1, %2, %3\n\tv_min_u32 %0, %0, %4" : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
  return r;
}
#define ME_MIN4(k) me_min4(k##_0, k##_1, k##_2, k##_3)

// butterfly transpose-reduce: merge two slot registers into one; lanes whose role bit is 0 keep
// slot a (min over the lane pair), lanes whose role bit is 1 keep slot b.
// Levels 0/1 (447 of the 588 merges): two DPP mins, the second one bank-masked so that it only
// overwrites the role-0 lanes.  s_nop 1 = the 2 wait states a DPP read needs after a VALU write of its
// source (hipcc does not pad hazards inside an asm statement).
// The _nn variants carry no padding: tools/gen_me_tree.py (space_merges) emits them only where >= 3 other ops separate the
// merge from the ops that produce its inputs, and tools/check_dpp_hazard.py verifies the distance in the final ISA (make check-isa).
#define ME_MERGE_DPP_MASKED(NAME, PAD, CTRL, MASK0)                                                          \
  __device__ __forceinline__ uint32_t NAME(uint32_t a, uint32_t b) {                                         \
    uint32_t r;                                                                                              \
    asm volatile(PAD "v_min_u32_dpp %0, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                     \
        "v_min_u32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:" MASK0                                    \
        : "=&v"(r) : "v"(a), "v"(b));                                                                        \
    return r;                                                                                                \
  }
ME_MERGE_DPP_MASKED(me_merge0, "s_nop 1\n\t", "row_ror:8", "0x3")          // role = lane bit 3: lanes 0-7 of a row = banks 0,1
ME_MERGE_DPP_MASKED(me_merge1, "s_nop 1\n\t", "row_half_mirror", "0x5")    // role = lane bit 2: lanes 0-3, 8-11 = banks 0,2
ME_MERGE_DPP_MASKED(me_merge0_nn, "", "row_ror:8", "0x3")
ME_MERGE_DPP_MASKED(me_merge1_nn, "", "row_half_mirror", "0x5")
__device__ __forceinline__ uint32_t me_merge2(uint32_t a, uint32_t b) {   // role = lane bit 5
  u32x2_t r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  return min(r.x, r.y);
}
__device__ __forceinline__ uint32_t me_merge3(uint32_t a, uint32_t b) {   // role = lane bit 4
  u32x2_t r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  return min(r.x, r.y);
}
template <int DPP_CTRL>
__device__ __forceinline__ uint32_t me_merge_dpp(uint32_t a, uint32_t b, bool role) {
  const uint32_t keep = role ? b : a;
  const uint32_t give = role ? a : b;
  return min(keep, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, DPP_CTRL, 0xf, 0xf, false));
}
#define me_merge4(a, b, role) me_merge_dpp<0x4E>(a, b, role)    // quad_perm [2,3,0,1] role = lane bit 1
#define me_merge5(a, b, role) me_merge_dpp<0xB1>(a, b, role)    // quad_perm [1,0,3,2] role = lane bit 0

// TComRdCost::xGetComponentBits (reference TComRdCost.cpp:278-292) in closed form
__device__ __forceinline__ uint32_t me_component_bits(int v) {
  const uint32_t t = (v <= 0) ? ((uint32_t)(-v) << 1) + 1u : ((uint32_t)v << 1);
  return 2u * (31u - (uint32_t)__builtin_clz(t)) + 1u;
}
// TComRdCost::getCost(x, y) with cost scale 2 (reference TComRdCost.h:172-189): uint32 wrap, >> 16
__device__ __forceinline__ uint32_t me_mv_cost(uint32_t lambda_q16, int x, int y, int pred_x, int pred_y) {
  return (lambda_q16 * (me_component_bits((x << 2) - pred_x) + me_component_bits((y << 2) - pred_y))) >> 16;
}

// 16-bit kernel (three candidates per lane, one lane-iteration per task): key = cost << 8 | lane(6) | j(2).  The 24-bit cost field
// holds HM's shifted sums with bi-prediction origins (<= 3 142 656 + 65 535) and the unshifted ones of
// hmme_search_params::shift_free (what cl/sad.cl computes: 10-bit <= 4 190 208 + 65 535, 9-bit bi-prediction origins
// <= 6 279 168 + 65 535; wider content when the samples of the call bound the sum below the marker, hmme.hip ctu_call); the invalid
// marker + the largest sum an invalid lane can add (< 8 000 000) stays < 2^24
constexpr int kIdxBits16 = 8;
constexpr uint32_t kInvCost16 = 8000000u;
constexpr int kThreads16 = 256;

// a CTU search cut into several workgroups: the 16-bit path cuts by candidate rows (LDS capacity), the 8-bit path
// by task range (latency of the per-CTU drop-in call, small pictures)
struct MeJob16 {
  MeJob j;
  int16_t y0, y1;      // 16-bit path: candidate rows [y0, y1); 8-bit split mode: tasks [y0, y1)
  int32_t job;         // index into the result arrays
};

// XCD-aware order of a launch's workgroups.  The dispatcher deals workgroup p of a grid to XCD p % 8, and each of the 8 XCDs has its own
// 4 MB L2.  The windows of neighbouring CTUs overlap by two thirds (192 of 192 + 64 columns, likewise rows): if neighbours run on the
// same XCD at about the same time their windows come out of that L2 instead of HBM.  So the N units of work of a launch (CTU searches
// in raster order, or their strips) are cut into 8 contiguous bands, and the p-th workgroup takes unit number (p >> 3) of band (p & 7):
// unit q <-> position p are a bijection on [0, N).  Results do not depend on it (every unit carries / derives its own output index).
__host__ __device__ inline int me_xcd_unit(int p, int n) {          // position -> unit
  const int x = p & 7, base = n >> 3, rem = n & 7;
  return x * base + (x < rem ? x : rem) + (p >> 3);
}
__host__ __device__ inline int me_xcd_position(int q, int n) {      // unit -> position
  const int base = n >> 3, rem = n & 7, big = rem * (base + 1);
  if (q < big) return (q % (base + 1)) * 8 + q / (base + 1);
  const int r = q - big;                                             // base >= 1 here: q >= big means some band has `base` units
  return (r % base) * 8 + rem + r / base;
}

// number of tasks me_search_kernel makes out of a wx x wy window (same arithmetic on host and device; the default is the split
// kernel's task size: host code and the prep kernels only count tasks for split launches)
// "fold": with 33 quads per row (the 129-wide window) and an odd number of rows, the 32-quad part's last iteration has an
// idle second row of lanes; its first lanes take the leftover quads of the LAST window row, so the narrow parts stop one
// row earlier (129 rows: 2 iterations of 64 rows instead of 3)
__host__ __device__ inline bool me_fold(int quads, int wy) { return (quads & 32) && (quads & 31) && (wy & 1); }
__host__ __device__ inline int me_num_tasks(int wx, int wy, int iter_per_task = kIterPerTaskSplit) {
  const int quads = (wx + 3) >> 2;
  const int wy_low = me_fold(quads, wy) ? wy - 1 : wy;
  int n = 0;
  for (int k = 5; k >= 0; --k)
    if (quads & (1 << k)) n += (((k == 5 ? wy : wy_low) + (64 >> k) - 1) / (64 >> k) + iter_per_task - 1) / iter_per_task;
  return n;
}
static_assert(sizeof(MeJob16) == 24, "MeJob16 layout");

// Task sizes of a whole-picture launch ("guided"): the lane-iterations of a part are dealt out 4 at a time while more than T4 (8)
// remain, then 2 at a time while more than T2 (2) remain, the last ones singly.  Large tasks keep the number of flushes low, small last ones keep the four waves of a
// workgroup level when the task counter runs dry (a wave that finishes a 4-iteration task early leaves its SIMD slot empty until the
// whole workgroup is done).  n4 / n2 / n1 = number of tasks of each size for `iters` iterations.
#ifndef ME_GUIDED_T4
#define ME_GUIDED_T4 8
#endif
#ifndef ME_GUIDED_T2
#define ME_GUIDED_T2 2
#endif
__host__ __device__ inline void me_guided(int iters, int& n4, int& n2, int& n1) {
  n4 = iters > ME_GUIDED_T4 ? (iters - ME_GUIDED_T4 + 3) >> 2 : 0;
  const int rem = iters - 4 * n4;                    // T4-3..T4, or iters itself when <= T4
  n2 = rem > ME_GUIDED_T2 ? (rem - ME_GUIDED_T2 + 1) >> 1 : 0;
  n1 = rem - 2 * n2;                                 // T2-1..T2 (0 only for iters == 0)
}
__host__ __device__ inline int me_num_tasks_guided(int wx, int wy) {
  const int quads = (wx + 3) >> 2;
  const int wy_low = me_fold(quads, wy) ? wy - 1 : wy;
  int n = 0;
  for (int k = 5; k >= 0; --k)
    if (quads & (1 << k)) {
      int n4, n2, n1;
      me_guided(((k == 5 ? wy : wy_low) + (64 >> k) - 1) / (64 >> k), n4, n2, n1);
      n += n4 + n2 + n1;
    }
  return n;
}

// 16-bit kernel: height of the strips a wx x wy window is cut into, given the most rows LDS holds (rows_max) and the number of
// strips the launch provides (max_strips).  The four waves of a workgroup pull lane-iterations from a counter and meet at a barrier
// after each column-parity pass, so a pass of n iterations costs ceil(n / 4) rounds: 25 iterations cost 7 rounds, 24 cost 6.
// Iterations of a pass over h rows: ceil(h * lanes_per_row / 64), lanes_per_row = ceil(candidates of one parity / 3) (three
// candidates per lane; the even pass has at least as many as the odd one).
// Strips are of EQUAL height ceil(wy / n): what a launch loses at its end is about half the lifetime of its longest workgroup, and
// a tall strip next to a short one made every launch measured slower than the even split of the same iterations (129 rows on
// 4 080 workgroups: 65 + 64 5.07 ms, 69 + 60 5.23, 81 + 48 6.10; 257 rows: 5 x 52 19.05 ms, 5 x 47 + 22 19.69).  The number of
// strips is the one with the fewest rounds, a strip counting half a round for its two window loads; ties go to fewer strips.
__host__ __device__ inline int me_strip_rows16(int wx, int wy, int rows_max, int max_strips) {
  const int lanes = (((wx + 1) >> 1) + 2) / 3;
  int best_h = (wy + max_strips - 1) / max_strips, best_cost = 0x7fffffff;
  for (int n = (wy + rows_max - 1) / rows_max; n <= max_strips && n <= wy; ++n) {
    const int h = (wy + n - 1) / n;
    const int used = (wy + h - 1) / h, rest = wy - (used - 1) * h;
    const int cost = 2 * ((used - 1) * ((((h * lanes + 63) >> 6) + 3) >> 2) + ((((rest * lanes + 63) >> 6) + 3) >> 2)) + used;
    if (cost < best_cost) { best_cost = cost; best_h = h; }
  }
  return best_h;
}

// The current picture is read from its CTU-BLOCKED copy (hmme_plane::d_blocks, written by me_fill_blocks_kernel beside the padded plane): CTU
// (cx, cy) is one contiguous block of 64 rows x 64 samples, blocks in raster order, partial edge CTUs completed by edge replication.  A
// current-block load of the search kernels is then `s_load_dwordx2 / x4` at the IMMEDIATE offset row * 64 * BPS + 8 * BPS * q from the block's
// address, whatever the picture's size: no pitch in the kernel, no s_mul_i32 in front of the 512 loads of a lane-iteration (5.5 % of its
// instructions -- round 5 removed them for the pitches of 2160p and 1080p planes only, by compile-time pitch instantiations that this
// layout replaces), the block's 64 (128) cache lines consecutive instead of one per plane row.  The per-CTU call hands over a dense
// 64 x 64 block already: one block, the same kernels.
constexpr int kBlkBytes8 = 64 * 64, kBlkBytes16 = 2 * 64 * 64;
// Pulls a workgroup's current block (64 * LINES consecutive 64-byte lines) into the scalar cache while the window is being staged:
// the first lane-iteration would otherwise meet every line cold, one CU at a time -- visible where a workgroup runs only a few
// iterations (the per-CTU call: 64 workgroups per search).  One wave issues the loads; nothing reads the results.
template <int LINES>
__device__ __forceinline__ void me_prefetch_cur(uint64_t curc) {
  // every load targets the SAME register `d`, an in-out operand of each statement and of the final wait: it stays allocated for the
  // whole sequence, so the compiler can never hand it to another value while loads into it are still in flight -- the inline scalar
  // loads are invisible to its own waitcnt / liveness tracking
  uint32_t d = 0;
#pragma unroll
  for (int l = 0; l < 64 * LINES; ++l) asm volatile("s_load_dword %0, %1, %2" : "+s"(d) : "s"(curc), "n"(64 * l));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d));
}

// Window staging shared by the search kernels: LDS dword i = window row i / PDW, dword i % PDW, realigned by `mis` bytes.  Eight
// loads are in flight per thread before the first one is waited for: the straightforward loop (load, wait, store) cost one memory
// round trip per 256 dwords -- 37 of them for a 129 x 129 window, ~4 % of a workgroup's lifetime, 63 per pass and strip (12 %) in
// the 16-bit kernel.
template <int PDW, int THREADS>
__device__ __forceinline__ void me_stage_window(uint32_t* win, const uint32_t* __restrict__ src_al, int pitch_dw, int n, uint32_t mis, int tid) {
  constexpr int U = 8;
  for (int i0 = tid; i0 < n; i0 += THREADS * U) {
    uint32_t lo[U], hi[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(i0 + u * THREADS, n - 1);          // clamped: every load is in bounds, surplus ones are not stored
      const int r = i / PDW, k = i - r * PDW;
      lo[u] = src_al[(long)r * pitch_dw + k];
      hi[u] = src_al[(long)r * pitch_dw + k + 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * THREADS;
      if (i < n) win[i] = __builtin_amdgcn_alignbyte(hi[u], lo[u], mis);
    }
  }
}

// Segment table of a SPLIT = 2 launch of me_search_kernel (written by me_prep_segments_kernel), one allocation:
//   int4    segs[W]        per workgroup: (job its segment starts in, first unit, end unit, h) in the list of UNITS of the launch's TAIL jobs
//                          (jobs h .. n - 1), or (job, 0, 0, -1): the workgroup searches that job whole (the head jobs 0 .. h - 1, one each)
//   MeJob16 jobs[n]        the jobs (y0 / y1 unused; `job` = index into the result arrays); me_finalize16_kernel decodes against them
//   int     prefix[n + 1]  entry j - h: units of the tail jobs before job j -- job j >= h owns the list's units [prefix[j - h], prefix[j - h + 1])
// Head and tail in ONE launch: the head's workgroups come first in the grid, and a segment starts on whichever slot a head job (the clipped
// windows of the picture's edge CTUs make short ones) leaves first -- as two launches the tail waited for the head's last workgroup.
// A unit is kSegUnit = 4 consecutive tasks (lane-iterations) of one job -- one per wave of the workgroup; a job's last unit may be short.
// Segments are cut at unit boundaries: where a segment leaves one job and enters the next the four waves meet at a barrier, and a piece
// of a job that is not a multiple of four tasks would leave waves idle in its last round on BOTH sides of every such boundary.
constexpr int kSegUnit = 4;
__host__ __device__ inline size_t me_seg_table_bytes(int n_wg, int n_jobs) { return sizeof(int4) * (size_t)n_wg + sizeof(MeJob16) * (size_t)n_jobs + sizeof(int) * ((size_t)n_jobs + 1); }
__host__ __device__ inline const int4* me_seg_table_segs(const void* t) { return (const int4*)t; }
__host__ __device__ inline const MeJob16* me_seg_table_jobs(const void* t, int n_wg) { return (const MeJob16*)((const int4*)t + n_wg); }
__host__ __device__ inline const int* me_seg_table_prefix(const void* t, int n_wg, int n_jobs) { return (const int*)(me_seg_table_jobs(t, n_wg) + n_jobs); }

// ---- the search kernel --------------------------------------------------------------------------
// windows wider or taller than 129 candidates (8-bit planes, search range 65..128) are cut into up to 2 x 2 tiles of at most
// 129 x 129: MeJob16::job carries the tile's (x, y) index in bits 30 and 29 above the output job index
constexpr int kTileStep = 129, kTileJobMask = 0x1fffffff;
// SPLIT = 0: one workgroup searches the whole window of jobs[blockIdx.x] (MeJob) and writes its 593 results.
// SPLIT = 1: jobs are MeJob16; the workgroup runs tasks [y0, y1) only and merges into g_best with 64-bit atomicMin
//            (decoded afterwards by me_finalize16_kernel) -- used where one CTU must fill many CUs.
// SPLIT = 2: the tasks of ALL jobs of the launch form one list (job after job), cut into gridDim.x equal SEGMENTS (me_prep_segments_kernel);
//            the workgroup runs segment blockIdx.x, which may end one job and begin the next (window staged again, table merged per job).
//            A launch that does not fill whole rounds of the chip's workgroup slots -- a small picture, the jobs beyond the last full
//            round of a larger one -- is dealt this way: every workgroup gets the same number of lane-iterations, whatever the number
//            of jobs and however the picture's edge clips their windows.  jobs_v: MeSegTable layout (below).
// n_seg_jobs: SPLIT = 2 only, the number of jobs the segment table covers
// curs: the CTU-blocked copies of the current pictures (above me_prefetch_cur), cur_ctus_x: CTUs per picture row (blocks per block row).
template <int FEN, int SPLIT>
__global__ void __launch_bounds__(kThreads, 2)
me_search_kernel(const RefSet curs, int cur_ctus_x, const RefSet refs, int ref_pitch,
                 const void* __restrict__ jobs_v, uint32_t lambda_q16, int16_t* __restrict__ out_mv,
                 uint32_t* __restrict__ out_sad, unsigned long long* __restrict__ g_best, int fair_prio, int n_seg_jobs) {
  __shared__ uint32_t win[kWinRowsMax * kPDW];
  __shared__ unsigned long long best64[kParts];
  __shared__ int task_ctr;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
#ifdef ME_SEARCH_T_TIMELINE   // timing-only build (results are overwritten): when each workgroup starts, has its window staged, runs dry, ends (100 MHz wall clock)
  const unsigned long long tl_start = wall_clock64();
#endif
  MeJob job;
  int t_first = 0, t_end = 0x7fffffff, out_job = blockIdx.x;
  unsigned long long tile_off = 0;   // SPLIT, tiled windows: (y, x) of this tile's first candidate in the CTU's whole window
  int seg_j = 0, seg_g = 0, seg_end = 0, seg_head = 0;   // SPLIT = 2: the job the segment is in, the next unit of the tail's list, the segment's end; the number of head jobs (-1: this workgroup searches job seg_j whole)
  if constexpr (SPLIT == 2) {
    const int4 sg = me_seg_table_segs(jobs_v)[blockIdx.x];
    seg_j = sg.x; seg_g = sg.y; seg_end = sg.z; seg_head = sg.w;
    if (seg_head >= 0 && seg_g >= seg_end) return;      // fewer units than workgroups (tiny windows): nothing for this one
  } else if constexpr (SPLIT == 1) {
    const MeJob16 jb = ((const MeJob16*)jobs_v)[blockIdx.x];
    job = jb.j; t_first = jb.y0; t_end = jb.y1; out_job = jb.job & kTileJobMask;
    tile_off = (unsigned long long)(((jb.job >> 29) & 1) * kTileStep) << 16 | (unsigned long long)(((jb.job >> 30) & 1) * kTileStep);
  } else {
    job = ((const MeJob*)jobs_v)[blockIdx.x];
    out_job = me_xcd_unit(blockIdx.x, gridDim.x);   // the job table of a whole-job launch is in XCD order (me_prep_jobs_kernel)
  }
#pragma unroll 1
  for (bool first_job = true;; first_job = false) {   // one pass, except SPLIT = 2: one pass per job the segment reaches into
  if constexpr (SPLIT == 2) {
    job = me_seg_table_jobs(jobs_v, gridDim.x)[seg_j].j;
    out_job = seg_j;
    if (seg_head >= 0) {
      const int* prefix = me_seg_table_prefix(jobs_v, gridDim.x, n_seg_jobs) - seg_head;
      const int p0 = prefix[seg_j], p1 = prefix[seg_j + 1];
      t_first = (seg_g - p0) * kSegUnit; t_end = (min(seg_end, p1) - p0) * kSegUnit;   // the job's last unit may be short: n_tasks clips
      seg_g = p1; ++seg_j;
    }                                  // else: a head job, whole (t_first = 0, t_end = everything; seg_g == seg_end: one pass)
    if (!first_job) __syncthreads();   // every thread has merged the previous job's table; window and table are free again
  }
#if ME_FAIR_PRIO
  if (fair_prio) __builtin_amdgcn_s_setprio(3);   // a new workgroup comes first: its window loads go out at once, its first tasks run ahead of the older workgroup's last
#endif
  const uint8_t* __restrict__ ref_base = refs.base[job.ctu_x & 63];
  const uint8_t* __restrict__ cur_base = curs.base[job.ctu_x & 63];
  job.ctu_x &= ~63;
  const int wx = job.rb_x - job.lt_x + 1, wy = job.rb_y - job.lt_y + 1;   // candidates per row / rows

  for (int s = tid; s < kParts; s += kThreads) best64[s] = ~0ull;
  if (tid == 0) task_ctr = t_first;

  // -- 0. the current block is read through the scalar cache straight into SGPRs (v_qsad_pk_u16_u8 takes its 4 current-block bytes
  //       from one): no LDS copy, no wave-uniform ds_read_b64 (each cost a full LDS slot), 35 VGPRs fewer
  const uintptr_t cur_addr = (uintptr_t)(cur_base + ((long)(job.ctu_y >> 6) * cur_ctus_x + (job.ctu_x >> 6)) * kBlkBytes8);
  const uint64_t curc = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)cur_addr) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(cur_addr >> 32)) << 32;
  if (tid < 64) me_prefetch_cur<1>(curc);
  // -- 1. stage the reference window: LDS row r, byte b  <->  ref(ctu_x + lt_x + b, ctu_y + lt_y + r)
  {
    const uint8_t* src = ref_base + (long)(job.ctu_y + job.lt_y) * ref_pitch + (job.ctu_x + job.lt_x);
    const uint32_t mis = (uint32_t)(uintptr_t)src & 3u;
    const uint32_t* src_al = (const uint32_t*)(src - mis);
    const int pitch_dw = ref_pitch >> 2;
    const int n = (wy + 63) * kPDW;
    me_stage_window<kPDW, kThreads>(win, src_al, pitch_dw, n, mis, tid);
  }
  __syncthreads();
#ifdef ME_SEARCH_T_TIMELINE
  const unsigned long long tl_staged = wall_clock64();
#endif

  // -- 2. task list: the ceil(wx/4) candidate quads of a row are split into power-of-two parts
  //       (129 -> 32 + 1); part k lays a wave out as 2^k quads x (64 >> k) rows per iteration.
  const int quads = (wx + 3) >> 2;
  constexpr int kIt = SPLIT ? kIterPerTaskSplit : kIterPerTask;
  constexpr bool kGuided = !SPLIT && ME_GUIDED_TASKS;
  const int n_tasks = kGuided ? me_num_tasks_guided(wx, wy) : min(me_num_tasks(wx, wy, kIt), t_end);
  const bool fold = me_fold(quads, wy);
  const int wy_low = fold ? wy - 1 : wy;

  const uint32_t mult_a = 1u << kIdxBits;
  const uint32_t mult_e = FEN ? (2u << kIdxBits) : (1u << kIdxBits);
  const bool rb1 = lane & 2, rb0 = lane & 1;
#if ME_FAIR_PRIO
  // Wave priority falls with the wave's progress (s_setprio 3 .. 0 over its expected share of the workgroup's lane-iterations).  A SIMD
  // holds one wave of each of the CU's two workgroups and issues from the OLDER one whenever it can: of two workgroups that start
  // together the older ran at full speed and the younger in its gaps (a fifth of the rate), then alone -- one wave per SIMD issues at
  // ~0.8 of what two do -- for the rest of its life (profiles/r05b_search_timeline.txt: lifetimes 415 / 675 us in a single-round 1080p
  // launch).  With the wave that is behind given the SIMD first, the two stay within a quarter of their work of each other and end
  // together; in a launch of several rounds the same rule shortens the lonely ends of each CU's last workgroups.
  // state: lane-iterations left at the current level << 2 | level (0..3 = s_setprio 3..0); without priorities the first level never ends
  int prio_quarter = 0x0fffffff;
  if (fair_prio) {
    if constexpr (!SPLIT) {
      int total = 0;
      for (int kk = 5; kk >= 0; --kk)
        if (quads & (1 << kk)) total += ((kk == 5 ? wy : wy_low) + (64 >> kk) - 1) / (64 >> kk);
      prio_quarter = max(1, (total + 15) >> 4);   // a quarter of a wave's share (four waves)
    } else {
      prio_quarter = max(1, ((n_tasks - t_first) * kIt + 15) >> 4);   // this workgroup's slice of the CTU's tasks
    }
  }
  int prio_state = prio_quarter << 2;
#endif

  static_assert(SPLIT != 2 || kIterPerTaskSplit == 1, "segment mode counts tasks in lane-iterations");
  int grab_next = 0, grab_end = 0;   // SPLIT = 2: the tasks this wave has drawn and not yet run
  while (true) {
    int t = 0;
    if constexpr (SPLIT == 2) {
      // a wave draws a quarter of what is left, at most 4 tasks, and runs them as ONE task of that many lane-iterations where they lie in
      // one part of the window (one flush of the running minima for up to four lane-iterations instead of one each); guided
      // self-scheduling over the four waves ends them level: 12 tasks go 3 + 3 + 2 + 1, + 1 + 1 + 1 = three each
      if (grab_next >= grab_end) {
        int want = 0;
        if (lane == 0) {
          const int left = n_tasks - *(volatile int*)&task_ctr;
          want = min(4, max(1, (left + 3) >> 2));
          t = atomicAdd(&task_ctr, want);
        }
        t = __builtin_amdgcn_readfirstlane(t);
        want = __builtin_amdgcn_readfirstlane(want);
        if (t >= n_tasks) break;
        grab_next = t; grab_end = min(t + want, n_tasks);
      }
      t = grab_next;
    } else {
      if (lane == 0) t = atomicAdd(&task_ctr, 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= n_tasks) break;
    }
#if ME_FAIR_PRIO
    if (prio_state < 4 && (prio_state & 3) < 3) {   // the level's share is used up: one step down
      prio_state += (prio_quarter << 2) + 1;
      if ((prio_state & 3) == 1) __builtin_amdgcn_s_setprio(2);
      else if ((prio_state & 3) == 2) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
#endif
    // decode task t -> (x0, k, first iteration)
    int x0 = 0, k = 0, it0 = 0, n_it = 0;
    {
      int xq = 0, rem = t;
      for (int kk = 5; kk >= 0; --kk) {
        if (!(quads & (1 << kk))) continue;
        const int iters = ((kk == 5 ? wy : wy_low) + (64 >> kk) - 1) / (64 >> kk);
        int nt;
        if constexpr (kGuided) {
          int n4, n2, n1;
          me_guided(iters, n4, n2, n1);
          nt = n4 + n2 + n1;
          if (rem < nt) {
            x0 = xq * 4; k = kk;
            if (rem < n4) { it0 = 4 * rem; n_it = 4; }
            else if (rem < n4 + n2) { it0 = 4 * n4 + 2 * (rem - n4); n_it = 2; }
            else { it0 = 4 * n4 + 2 * n2 + (rem - n4 - n2); n_it = 1; }
            break;
          }
        } else {
          nt = (iters + kIt - 1) / kIt;
          if (rem < nt) {
            x0 = xq * 4; k = kk; it0 = rem * kIt; n_it = min(kIt, iters - it0);
            if constexpr (SPLIT == 2) { n_it = min(grab_end - grab_next, iters - it0); grab_next += n_it; }   // kIt = 1: tasks are lane-iterations
            break;
          }
        }
        rem -= nt;
        xq += 1 << kk;
      }
    }
    const int ty = 64 >> k;
    const int lx = lane & ((1 << k) - 1), ly = lane >> k;
    const bool fold_part = fold && k == 5;

    // running minima of the task: register g, lane l <-> slot ME_SLOT_OF[g][l]
    uint32_t b0 = ME_MAXKEY, b1 = ME_MAXKEY, b2 = ME_MAXKEY, b3 = ME_MAXKEY, b4 = ME_MAXKEY, b5 = ME_MAXKEY,
             b6 = ME_MAXKEY, b7 = ME_MAXKEY, b8 = ME_MAXKEY, b9 = ME_MAXKEY;

#if ME_FAIR_PRIO
    prio_state -= n_it << 2;
#endif
    for (int it = 0; it < n_it; ++it) {
      int cx = x0 + 4 * lx, cy = (it0 + it) * ty + ly;
      if (fold_part && cy == wy) { cx += 128; cy = wy - 1; }   // idle second row of the last iteration: leftover quads of the last row
      const bool vy = cy < wy;
      // per-candidate constants: (mv cost | invalid marker) << 10 | iteration | lane | j
      const int mvy = job.lt_y + cy, mvx = job.lt_x + cx;
      const uint32_t by = me_component_bits((mvy << 2) - job.pred_y);
      uint32_t lane_now = (uint32_t)lane;
      asm("" : "+v"(lane_now));   // keep `lane << 2` out of the loop-invariant registers: at 255 VGPRs it was the one value spilled
      const uint32_t tag = ((uint32_t)it << 8) | (lane_now << 2);
      uint32_t cc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t cost = (lambda_q16 * (me_component_bits(((mvx + j) << 2) - job.pred_x) + by)) >> 16;
        cc[j] = (((vy && (cx + j) < wx) ? cost : kInvCost) << kIdxBits) | tag | (uint32_t)j;
      }
      const uint32_t c0 = cc[0], c1 = cc[1], c2 = cc[2], c3 = cc[3];
      const uint32_t nc0 = 0u - c0, nc1 = 0u - c1, nc2 = 0u - c2, nc3 = 0u - c3;
      // lanes outside the window still run (wave-uniform code) on clamped, in-bounds addresses
      const lds_char_t* lpc = (const lds_char_t*)(win + min(cy, wy - 1) * kPDW + (min(cx, wx - 1) >> 2));
      // scalar loads complete out of order: ME8_CUR_WAIT (placed by the generator one CU after the loads, before the next CU's are
      // issued) waits for the batch; the window dwords of the same batch are named first so that the compiler's LDS wait lands there
#define ME8_CUR(row, q)                                                                                                            \
  ({                                                                                                                               \
    uint64_t w_;                                                                                                                   \
    asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(w_) : "s"(curc), "n"((row) * 64 + 8 * (q)));                                    \
    w_;                                                                                                                            \
  })
#define ME8_CUR_WAIT(w0, w1, w2, w3, w4, w5, w6, w7, d0, d1, d2, d3, d4, d5, d6, d7, d8, d9, d10, d11, d12, d13, d14, d15)           \
  asm volatile("" : : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7), "v"(d8), "v"(d9), "v"(d10), "v"(d11),  \
               "v"(d12), "v"(d13), "v"(d14), "v"(d15));                                                                             \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w0), "+s"(w1), "+s"(w2), "+s"(w3), "+s"(w4), "+s"(w5), "+s"(w6), "+s"(w7))
      if constexpr (FEN) {
#include "me_tree_fen1.inc"
      } else {
#include "me_tree_fen0.inc"
      }
    }

    // -- 3. fold the wave's running minima into the CTU table; (cost, y, x) keys give raster-order ties
#define ME_FLUSH(g, key_)                                                                                          \
    {                                                                                                              \
      const int slot = ME_SLOT_OF[g][lane];                                                                        \
      const uint32_t key = (key_);                                                                                 \
      const uint32_t cost = key >> kIdxBits;                                                                       \
      if (slot >= 0 && cost < kInvCost) {                                                                          \
        const int kit = (key >> 8) & 3, kl = (key >> 2) & 63, kj = key & 3;                                        \
        int bx = x0 + 4 * (kl & ((1 << k) - 1)) + kj;                                                              \
        int byy = (it0 + kit) * ty + (kl >> k);                                                                    \
        if (fold_part && byy == wy) { bx += 128; byy = wy - 1; }                                                   \
        atomicMin(&best64[slot],                                                                                   \
                  ((unsigned long long)cost << 32) | ((unsigned long long)byy << 16) | (unsigned long long)bx);    \
      }                                                                                                            \
    }
    ME_FLUSH(0, b0) ME_FLUSH(1, b1) ME_FLUSH(2, b2) ME_FLUSH(3, b3) ME_FLUSH(4, b4)
    ME_FLUSH(5, b5) ME_FLUSH(6, b6) ME_FLUSH(7, b7) ME_FLUSH(8, b8) ME_FLUSH(9, b9)
#undef ME_FLUSH
  }
#ifdef ME_SEARCH_T_TIMELINE
  const unsigned long long tl_dry = wall_clock64();   // this wave found the task counter dry
#endif
  __syncthreads();
#ifdef ME_SEARCH_T_TIMELINE
  const unsigned long long tl_all = wall_clock64();
#endif

  // -- 4. results: integer MV (TComMv layout) and the pure SAD at the arg-min (ruiSAD, reference
  //       TEncSearch.cpp:3895: best - getCost(best))
  if constexpr (SPLIT) {
    for (int s = tid; s < kParts; s += kThreads) {
      const unsigned long long v = best64[s];
      if (v != ~0ull) atomicMin(&g_best[(long)out_job * kParts + s], v + tile_off);
    }
    if constexpr (SPLIT == 2) {
      if (seg_g < seg_end) continue;   // the segment goes on in the next job
    }
    return;
  }
  for (int s = tid; s < kParts; s += kThreads) {
    const unsigned long long v = best64[s];
    const int mvx = job.lt_x + (int)(v & 0xffff), mvy = job.lt_y + (int)((v >> 16) & 0xffff);
    const uint32_t cost = (uint32_t)(v >> 32);
    const long o = (long)out_job * kParts + s;
    out_mv[2 * o] = (int16_t)mvx;
    out_mv[2 * o + 1] = (int16_t)mvy;
    out_sad[o] = cost - me_mv_cost(lambda_q16, mvx, mvy, job.pred_x, job.pred_y);
  }
#ifdef ME_SEARCH_T_TIMELINE
  __syncthreads();
  if (lane == 0) {
    uint32_t* o = out_sad + (long)out_job * kParts;
    const int wv = tid >> 6;
    if (wv == 0) {
      const unsigned long long t1 = wall_clock64();
      uint32_t hw_id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
      uint32_t xcc_id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
      o[0] = (uint32_t)tl_start; o[1] = (uint32_t)(tl_start >> 32); o[2] = (uint32_t)(tl_staged - tl_start); o[3] = (uint32_t)(tl_all - tl_start);
      o[4] = (uint32_t)(t1 - tl_start); o[5] = hw_id; o[6] = xcc_id; o[7] = blockIdx.x;
    }
    o[8 + wv] = (uint32_t)(tl_dry - tl_start);
  }
#endif
  break;   // SPLIT = 0: the one job is done (SPLIT = 1 returned above, SPLIT = 2 went on or returned)
  }
}

// ---- frame helpers ---------------------------------------------------------------------------------

// TComDataCU::clipMv (reference TComDataCU.cpp:2907-2920), quarter pels, maxCU = 64
__host__ __device__ inline void clip_mv_q(int& x, int& y, int cu_x, int cu_y, int pic_w, int pic_h) {
  const int hor_max = (pic_w + 8 - cu_x - 1) << 2, hor_min = (-64 - 8 - cu_x + 1) << 2;
  const int ver_max = (pic_h + 8 - cu_y - 1) << 2, ver_min = (-64 - 8 - cu_y + 1) << 2;
  x = (int16_t)(x > hor_max ? hor_max : (x < hor_min ? hor_min : x));
  y = (int16_t)(y > ver_max ? ver_max : (y < ver_min ? ver_min : y));
}
// TEncSearch::xSetSearchRange (reference TEncSearch.cpp:3814-3830)
__host__ __device__ inline void set_search_range(int pred_x, int pred_y, int sr, int cu_x, int cu_y, int pic_w,
                                                 int pic_h, int& lt_x, int& lt_y, int& rb_x, int& rb_y) {
  int px = pred_x, py = pred_y;
  clip_mv_q(px, py, cu_x, cu_y, pic_w, pic_h);
  int ltx = (int16_t)(px - (sr << 2)), lty = (int16_t)(py - (sr << 2));
  int rbx = (int16_t)(px + (sr << 2)), rby = (int16_t)(py + (sr << 2));
  clip_mv_q(ltx, lty, cu_x, cu_y, pic_w, pic_h);
  clip_mv_q(rbx, rby, cu_x, cu_y, pic_w, pic_h);
  lt_x = ltx >> 2; lt_y = lty >> 2; rb_x = rbx >> 2; rb_y = rby >> 2;
}

// one job per CTU of the picture from the per-CTU predictors
// job i = reference (i / ctu_count), CTU ctu_first + (i % ctu_count); pred_q is [n_refs][n_ctu][2]
// jobs [job0, job0 + n_jobs) of the ctu_count * n_refs (CTU, reference) searches of a launch; jobs[] is indexed from job0
// xcd_order: jobs[] is written in the XCD-aware order of me_search_kernel<FEN, 0> (position li holds job job0 + me_xcd_unit(li, n_jobs));
// 0 for the refinement kernel, which takes job blockIdx.x
// job_counter (may be null): the refinement kernel's work counter, reset here for the launch that follows on the same stream
__global__ void me_prep_jobs_kernel(MeJob* jobs, const int16_t* __restrict__ pred_q, int ctu_first, int ctu_count,
                                    int n_refs, int pic_w, int pic_h, int sr, int job0, int n_jobs, int xcd_order, uint32_t* job_counter) {
  const int li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li == 0 && job_counter) *job_counter = 0;
  if (li >= n_jobs) return;
  const int i = job0 + (xcd_order ? me_xcd_unit(li, n_jobs) : li);
  const int r = i / ctu_count;
  const int ctus_x = (pic_w + 63) >> 6, n_ctu = ctus_x * ((pic_h + 63) >> 6);
  const int ctu = ctu_first + (i - r * ctu_count);
  const int cu_x = (ctu % ctus_x) * 64, cu_y = (ctu / ctus_x) * 64;
  const long pq = 2 * ((long)r * n_ctu + ctu);
  const int px = pred_q ? pred_q[pq] : 0, py = pred_q ? pred_q[pq + 1] : 0;
  int ltx, lty, rbx, rby;
  set_search_range(px, py, sr, cu_x, cu_y, pic_w, pic_h, ltx, lty, rbx, rby);
  MeJob j;
  j.ctu_x = (int16_t)(cu_x | r); j.ctu_y = (int16_t)cu_y;
  j.lt_x = (int16_t)ltx; j.lt_y = (int16_t)lty; j.rb_x = (int16_t)rbx; j.rb_y = (int16_t)rby;
  j.pred_x = (int16_t)px; j.pred_y = (int16_t)py;
  jobs[li] = j;
}

// picture area (int16 Pel, u16 or u8; pitch in elements) -> padded u8 / u16 plane, borders edge-replicated like
// TComPicYuv::extendPicBorder (reference TComPicYuv.cpp:214-262).  One thread per 4 output bytes.
// flag[0] is set when a sample lies outside [0, max_val].
template <typename SrcT, typename DstT>
__global__ void me_fill_plane_kernel(uint8_t* __restrict__ dst, int dst_pitch, int margin_x, int margin_y, int w, int h,
                                     const SrcT* __restrict__ src, int src_pitch, int max_val, int* __restrict__ flag) {
  constexpr int N = 4 / (int)sizeof(DstT);                        // samples per thread
  const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * N;   // plane column of the first sample
  const int y = blockIdx.y;                                       // plane row
  if (x0 * (int)sizeof(DstT) >= dst_pitch) return;
  const int sy = min(max(y - margin_y, 0), h - 1);
  uint32_t packed = 0;
  bool bad = false;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int sx = min(max(x0 + i - margin_x, 0), w - 1);
    const int v = (int)src[(long)sy * src_pitch + sx];
    bad |= (v < 0) | (v > max_val);
    packed |= (uint32_t)(v & (sizeof(DstT) == 1 ? 0xff : 0xffff)) << (8 * (int)sizeof(DstT) * i);
  }
  *(uint32_t*)(dst + (long)y * dst_pitch + (long)x0 * sizeof(DstT)) = packed;
  if (bad) atomicOr(flag, 1);
}

// the same picture area -> the plane's CTU-blocked copy (see me_prefetch_cur): block (cx, cy) holds samples (64 cx + c, 64 cy + r) at
// [r][c], coordinates beyond the picture clamped to its last column / row (what the padded plane holds there).  One thread per 4 output
// bytes, a row of a block per 16 (32) threads: 64-byte (128-byte) contiguous writes.  Range violations are the padded fill's to report.
template <typename SrcT, typename DstT>
__global__ void me_fill_blocks_kernel(uint8_t* __restrict__ dst, int ctus_x, int n_ctu, int w, int h, const SrcT* __restrict__ src, int src_pitch) {
  constexpr int N = 4 / (int)sizeof(DstT);                        // samples per thread
  constexpr int PER_ROW = 64 / N, PER_BLK = 64 * PER_ROW;         // threads per block row / per block
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int blk = (int)(t / PER_BLK), in = (int)(t - (long)blk * PER_BLK);
  if (blk >= n_ctu) return;
  const int r = in / PER_ROW, c0 = (in - r * PER_ROW) * N;
  const int sy = min((blk / ctus_x) * 64 + r, h - 1), x0 = (blk % ctus_x) * 64 + c0;
  uint32_t packed = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int v = (int)src[(long)sy * src_pitch + min(x0 + i, w - 1)];
    packed |= (uint32_t)(v & (sizeof(DstT) == 1 ? 0xff : 0xffff)) << (8 * (int)sizeof(DstT) * i);
  }
  *(uint32_t*)(dst + t * 4) = packed;
}

// ---- 16-bit sample path (bit depth 9..12, bi-prediction origins of any depth) -----------------------------------------------
// Same task / key / butterfly / merge machinery as me_search_kernel; differences (see tools/gen_me_tree.py, class Tree16):
// v_sad_u16 leaves on u16 samples, three candidates per lane (x, x+2, x+4), exact 32-bit sums (candidates 0 and 1 as one 64-bit
// register pair: one add for both) and
//   key = ((sum << fen_shift) >> (bitDepth-8)) << kIdxBits16 + c     (reference TComRdCost.cpp:520-521)
// formed as (S' & ~0xff) + c from sums that the tree's first add of two leaves has already shifted to the key's field,
// lanes packed linearly over the window (candidate triple q = iteration*64 + lane), one lane-iteration per task, and the window
// is cut into horizontal strips of candidate rows so that one strip's reference rows fit LDS (SR 128: 320 x 322 samples); strips
// of a CTU are separate workgroups that merge through 64-bit atomicMin on a global table.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32_t;
typedef volatile __attribute__((address_space(3))) u32x4_t lds_vu32x4_t;
#define ME_SAD16(a, b, acc) __builtin_amdgcn_sad_u16((a), (b), (acc))

// key of each of the lane's three candidates from its exact sum, and their minimum; `asm volatile` for the same reason as me_min4
// (ordered against the masked merges).  The sums arrive shifted to the key's sum field (the tree's first add of two leaf sums shifts
// the result, tools/gen_me_tree.py Tree16): key = (S' & ~0xff) + c -- the mask drops the bits HM's >> (bitDepth-8) drops.
// This three-register form is what the generator emits with PAIR64 = False (the A/B build); the kernel as shipped uses me_keymin3_p
__device__ __forceinline__ uint32_t me_keymin3(uint32_t s0, uint32_t s1, uint32_t s2, uint32_t mask, uint32_t c0, uint32_t c1, uint32_t c2) {
  uint32_t r, t, u;
  asm volatile("v_and_b32 %0, %6, %3\n\tv_and_b32 %1, %6, %4\n\tv_and_b32 %2, %6, %5\n\t"
               "v_add_u32 %0, %0, %7\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %9\n\t"
               "v_min3_u32 %0, %0, %1, %2"
               : "=&v"(r), "=&v"(t), "=&v"(u) : "v"(s0), "v"(s1), "v"(s2), "s"(mask), "v"(c0), "v"(c1), "v"(c2));
  return r;
}

#define ME16_KEYMIN(s0, s1, s2) me_keymin3(s0, s1, s2, keymask16, c0, c1, c2)
// the same with candidates 0 and 1 as one 64-bit pair (tools/gen_me_tree.py PAIR64): two v_and_b32 on its halves, ONE 64-bit add for both
// keys (v_lshl_add_u64, shift 0; neither half carries into the other: a key is below 2^32), and + add for the third, v_min3_u32 -- six
// instructions where three separate candidates took seven.  Only the minimum is `asm volatile` (ordered against the masked merges)
__device__ __forceinline__ uint64_t me_pair(uint32_t lo, uint32_t hi) { return (uint64_t)lo | (uint64_t)hi << 32; }
__device__ __forceinline__ uint32_t me_keymin3_p(uint64_t s01, uint32_t s2, uint32_t mask, uint64_t c01, uint32_t c2) {
  const uint64_t k01 = (s01 & me_pair(mask, mask)) + c01;
  const uint32_t k2 = (s2 & mask) + c2;
  uint32_t r;
  asm volatile("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"((uint32_t)k01), "v"((uint32_t)(k01 >> 32)), "v"(k2));
  return r;
}
#define ME16_KEYMIN_P(s01, s2) me_keymin3_p(s01, s2, keymask16, c01, c2)

#ifdef ME_SEARCH_T_TIMELINE   // timing-only builds: per workgroup of me_search16_kernel -- start, [per pass: window staged, wave 0 dry, all dry], end (100 MHz wall clock), hardware id
__device__ uint32_t g_timeline16[16384 * 12];
#endif
// curs / cur_ctus_x: the CTU-blocked copies of the current pictures (u16 samples: 8 KiB per block), as in me_search_kernel
template <int FEN, int PDW>
__global__ void __launch_bounds__(kThreads16, 2)
me_search16_kernel(const RefSet curs, int cur_ctus_x, const RefSet refs, int ref_pitch,
                   const MeJob16* __restrict__ jobs, uint32_t lambda_q16, int sh, unsigned long long* __restrict__ g_best, int fair_prio) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  unsigned long long* best64 = (unsigned long long*)smem;              // [593] (+1 pad)
  int* task_ctr = (int*)(smem + 2 * 594);
  uint32_t* win = smem + 2 * 594 + 4;                                  // [(ny + 63)][PDW]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
#if ME_FAIR_PRIO
  if (fair_prio) __builtin_amdgcn_s_setprio(3);   // wave priority falls with the wave's progress: me_search_kernel
#endif
#ifdef ME_SEARCH_T_TIMELINE
  const unsigned long long tl0 = wall_clock64();
  uint32_t tl[8];
  int tln = 0;
#define ME16_STAMP() tl[tln++] = (uint32_t)(wall_clock64() - tl0)
#else
#define ME16_STAMP()
#endif
  const MeJob16 jb = jobs[blockIdx.x];
  MeJob job = jb.j;
  const uint8_t* __restrict__ ref_base = refs.base[job.ctu_x & 63];
  const uint8_t* __restrict__ cur_base = curs.base[job.ctu_x & 63];
  job.ctu_x &= ~63;
  const int wx = job.rb_x - job.lt_x + 1;
  const int ny = jb.y1 - jb.y0;                                        // candidate rows of this strip

  for (int s = tid; s < kParts; s += kThreads16) best64[s] = ~0ull;
  // key = ((sum << lsh) & ~0xff) + c  ==  (((sum << fen_shift) >> sh) << kIdxBits16) + c; the shift rides on the tree's first adds
  const uint32_t lsh_a = kIdxBits16 - sh, lsh_e = FEN ? kIdxBits16 + 1 - sh : lsh_a;
  const uint32_t keymask16 = ~((1u << kIdxBits16) - 1u);
  const bool rb1 = lane & 2, rb0 = lane & 1;
  constexpr int ME16_PDW = PDW;
  // The current block comes through the scalar cache into SGPRs (v_sad_u16 takes one SGPR operand): no LDS slot and no VGPR for
  // it -- as a wave-uniform ds_read_b128 each of its 512 reads per lane-iteration cost a full LDS slot, a third of the kernel's LDS
  // time.  Scalar loads complete out of order: ME16_CUR_WAIT (placed by the generator half a CU after the loads, before the next
  // batch is issued) waits for the whole batch and ties the loaded quads to the wait; the window dwords of the same batch are
  // named as inputs so that the compiler's own LDS wait lands in front of it, not behind the next batch.
  const uintptr_t cur_addr = (uintptr_t)(cur_base + ((long)(job.ctu_y >> 6) * cur_ctus_x + (job.ctu_x >> 6)) * kBlkBytes16);
  const uint64_t curc = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)cur_addr) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(cur_addr >> 32)) << 32;
  if (tid < 64) me_prefetch_cur<2>(curc);
#define ME16_CUR(row, q)                                                                                                           \
  ({                                                                                                                               \
    u32x4_t w_;                                                                                                                    \
    asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(w_) : "s"(curc), "n"((row) * 128 + 16 * (q)));                                  \
    w_;                                                                                                                            \
  })
#define ME16_CUR_WAIT(w0, w1, w2, w3, d0, d1, d2, d3, d4, d5, d6, d7, d8, d9, d10, d11)                                               \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w0), "+s"(w1), "+s"(w2), "+s"(w3)                                                       \
               : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7), "v"(d8), "v"(d9), "v"(d10), "v"(d11))

  // Two passes: the even window columns, then the odd ones, each over a window loaded with a shift of `par` samples.
  // A lane owns the candidates (x, x + 2, x + 4): all read dword-aligned u16 pairs, each one dword further on -- the six dwords
  // three 64-bit reads of a window row deliver.  Lane bases are 3 dwords apart: 4-byte-aligned reads (ds_read2_b32).
#pragma unroll 1
  for (int par = 0; par < 2; ++par) {
    if (par) __syncthreads();                                          // every wave is done with the previous window
    if (tid == 0) *task_ctr = 0;
    {
      const uint8_t* src = ref_base + (long)(job.ctu_y + job.lt_y + jb.y0) * ref_pitch + 2 * (job.ctu_x + job.lt_x + par);
      const uint32_t mis = (uint32_t)(uintptr_t)src & 3u;
      const uint32_t* src_al = (const uint32_t*)(src - mis);
      const int pitch_dw = ref_pitch >> 2;
      const int n = (ny + 63) * PDW;
      me_stage_window<PDW, kThreads16>(win, src_al, pitch_dw, n, mis, tid);
    }
    __syncthreads();
    ME16_STAMP();

  const int n_par = (wx + 1 - par) >> 1;                               // candidates of this column parity per window row
  const int pairs = (n_par + 2) / 3;                                   // lanes per window row
  const int n_iters = (ny * pairs + 63) >> 6;
  const int n_tasks = n_iters;                                         // one lane-iteration per task: a pass is ~25 iterations for 4 waves

  while (true) {
    int t = 0;
    if (lane == 0) t = atomicAdd(task_ctr, 1);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= n_tasks) break;
    const int it0 = t;
    constexpr int n_it = 1;
#if ME_FAIR_PRIO
    // Priority by the WORKGROUP's progress (the pass and the half of its iterations the task counter has reached), the same for its
    // four waves: they meet at a barrier after each pass, and per-wave levels (me_search_kernel's scheme) let the wave that had pulled
    // one iteration more fall behind the others by another -- barrier waits of 100 us instead of 35 (profiles/r05q_search16_timeline.txt)
    if (fair_prio) {
      const int lvl = 2 * par + (2 * t >= n_tasks ? 1 : 0);
      if (lvl == 0) __builtin_amdgcn_s_setprio(3);
      else if (lvl == 1) __builtin_amdgcn_s_setprio(2);
      else if (lvl == 2) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
#endif
    uint32_t b0 = ME_MAXKEY, b1 = ME_MAXKEY, b2 = ME_MAXKEY, b3 = ME_MAXKEY, b4 = ME_MAXKEY, b5 = ME_MAXKEY,
             b6 = ME_MAXKEY, b7 = ME_MAXKEY, b8 = ME_MAXKEY, b9 = ME_MAXKEY;
    for (int it = 0; it < n_it; ++it) {
      const int q = (it0 + it) * 64 + lane;
      const int row = q / pairs, pr = q - row * pairs;
      const int cx = par + 6 * pr, cy = jb.y0 + row;
      const bool vy = row < ny;
      const int mvy = job.lt_y + cy, mvx = job.lt_x + cx;
      const uint32_t by = me_component_bits((mvy << 2) - job.pred_y);
      const uint32_t tag = (uint32_t)lane << 2;
      uint32_t cc[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const uint32_t cost = (lambda_q16 * (me_component_bits(((mvx + 2 * j) << 2) - job.pred_x) + by)) >> 16;
        cc[j] = (((vy && (cx + 2 * j) < wx) ? cost : kInvCost16) << kIdxBits16) | tag | (uint32_t)j;
      }
      const uint32_t c0 = cc[0], c1 = cc[1], c2 = cc[2];
      const uint64_t c01 = me_pair(c0, c1);
      const lds_char_t* lpd = (const lds_char_t*)(win + min(row, ny - 1) * PDW + 3 * pr);
      if constexpr (FEN) {
#include "me_tree16_fen1.inc"
      } else {
#include "me_tree16_fen0.inc"
      }
    }
    // where a winning candidate lies: the key names the lane that owned it, and that lane knows its row and first column -- one
    // ds_bpermute_b32 (all lanes active here) instead of a division by `pairs` per running-minimum register (ten per task; the task
    // is one lane-iteration of this kernel, and its time follows its instruction count)
    const uint32_t my_yx = [&] {
      const int q = it0 * 64 + lane, row = q / pairs;
      return (uint32_t)(jb.y0 + row) << 16 | (uint32_t)(par + 6 * (q - row * pairs));
    }();
#define ME_FLUSH16(g, key_)                                                                                        \
    {                                                                                                              \
      const int slot = ME_SLOT_OF[g][lane];                                                                        \
      const uint32_t key = (key_);                                                                                 \
      const uint32_t cost = key >> kIdxBits16;                                                                     \
      const uint32_t yx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(key & 0xfcu), (int)my_yx);   /* lane (key >> 2) & 63, in bytes */ \
      if (slot >= 0 && cost < kInvCost16)                                                                          \
        atomicMin(&best64[slot], ((unsigned long long)cost << 32) | (unsigned long long)(yx + 2 * (key & 3)));     \
    }
    ME_FLUSH16(0, b0) ME_FLUSH16(1, b1) ME_FLUSH16(2, b2) ME_FLUSH16(3, b3) ME_FLUSH16(4, b4)
    ME_FLUSH16(5, b5) ME_FLUSH16(6, b6) ME_FLUSH16(7, b7) ME_FLUSH16(8, b8) ME_FLUSH16(9, b9)
#undef ME_FLUSH16
  }
    ME16_STAMP();   // this wave found the pass's counter dry
#ifdef ME_SEARCH_T_TIMELINE
    __syncthreads();
    ME16_STAMP();   // all waves dry
#endif
  }   // par
#undef ME16_CUR
#undef ME16_CUR_WAIT
  __syncthreads();
  for (int s = tid; s < kParts; s += kThreads16) atomicMin(&g_best[(long)jb.job * kParts + s], best64[s]);
#ifdef ME_SEARCH_T_TIMELINE
  if (tid == 0 && blockIdx.x < 16384) {
    uint32_t* o = g_timeline16 + blockIdx.x * 12;
    uint32_t hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    o[0] = (uint32_t)tl0; o[1] = (uint32_t)(tl0 >> 32);
    for (int i = 0; i < 6; ++i) o[2 + i] = tl[i];
    o[8] = (uint32_t)(wall_clock64() - tl0); o[9] = hw_id; o[10] = (uint32_t)ny; o[11] = (uint32_t)wx;
  }
#endif
#undef ME16_STAMP
}

// strips of one CTU have merged into g_best: decode (cost, y, x) -> TComMv + pure SAD.  Every entry read is set back to all ones: the
// table is left as the next launch needs it (hmme.hip merge_table: no preset launch in front of every split / strip / segment launch)
__global__ void me_finalize16_kernel(unsigned long long* __restrict__ g_best, const MeJob16* __restrict__ jobs,
                                     const int* __restrict__ first_strip_of_job, int n_jobs, uint32_t lambda_q16,
                                     int16_t* __restrict__ out_mv, uint32_t* __restrict__ out_sad) {
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= (long)n_jobs * kParts) return;
  const MeJob job = jobs[first_strip_of_job[o / kParts]].j;
  const unsigned long long v = g_best[o];
  g_best[o] = ~0ull;
  const int mvx = job.lt_x + (int)(v & 0xffff), mvy = job.lt_y + (int)((v >> 16) & 0xffff);
  out_mv[2 * o] = (int16_t)mvx;
  out_mv[2 * o + 1] = (int16_t)mvy;
  out_sad[o] = (uint32_t)(v >> 32) - me_mv_cost(lambda_q16, mvx, mvy, job.pred_x, job.pred_y);
}

// one MeJob16 per (CTU, strip) from the per-CTU predictors
// ---- per-CTU call: everything stays on the compute queue ------------------------------------------------------------
// the call block (jobs, merge-table preset, current block, window) is pulled from mapped pinned host memory by the GPU itself
// -- no copy-engine hop and no cross-queue dependency in front of the search kernel
__global__ void me_stage_call_kernel(const uint4* __restrict__ host_block, uint4* __restrict__ dev_block, int n16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) dev_block[i] = host_block[i];
}
// explicit weighted prediction (TEncSearch::setWpScalingDistParam, TEncSearch.cpp:5594-5635): the search of a slice with weighted
// prediction prices |org - (((w0 * ref + round) >> shift) + offset)| (TComRdCostWeightPrediction::xGetSADw,
// TComRdCostWeightPrediction.cpp:79-81).  The prediction of a sample does not depend on the candidate, so the staged window of a
// per-CTU call is weighted ONCE, into a second buffer (u16 samples, `bias` added so that block and window stay unsigned), and the 16-bit search
// kernel runs on it unchanged.
__global__ void me_weight_window_kernel(const uint8_t* __restrict__ win, uint8_t* __restrict__ out, int pitch, int rows, int cols, int w0, int round,
                                        int shift, int offset_bias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int y = i / cols, x = i - y * cols;
  const uint16_t v = ((const uint16_t*)(win + (long)y * pitch))[x];
  ((uint16_t*)(out + (long)y * pitch))[x] = (uint16_t)(((w0 * (int)v + round) >> shift) + offset_bias);   // the raw window stays: the refinement interpolates IT
}

// read-and-clear of a latched range-violation flag in one step: a fill kernel on another stream that sets it concurrently is either
// seen by this take or by the next one, never lost between a read and a separate clear
__global__ void me_take_flag_kernel(int* flag, int* out) { *out = atomicExch(flag, 0); }

// single-job finalize (one workgroup) that writes the results straight into mapped pinned host memory and then publishes a
// sequence number there: the host polls that word instead of sleeping in hipStreamSynchronize
__global__ void __launch_bounds__(640)
me_finalize1_kernel(const unsigned long long* __restrict__ g_best, const MeJob16* __restrict__ jobs, uint32_t lambda_q16,
                    int16_t* __restrict__ out_mv, uint32_t* __restrict__ out_sad, volatile uint32_t* done_flag, uint32_t seq) {
  const int o = threadIdx.x;
  if (o < kParts) {
    const MeJob job = jobs[0].j;
    const unsigned long long v = g_best[o];
    const int mvx = job.lt_x + (int)(v & 0xffff), mvy = job.lt_y + (int)((v >> 16) & 0xffff);
    out_mv[2 * o] = (int16_t)mvx;
    out_mv[2 * o + 1] = (int16_t)mvy;
    out_sad[o] = (uint32_t)(v >> 32) - me_mv_cost(lambda_q16, mvx, mvy, job.pred_x, job.pred_y);
  }
  __threadfence_system();
  __syncthreads();
  if (o == 0) { *done_flag = seq; __threadfence_system(); }
}

// completion word of a per-CTU call whose last kernel is not the finalize kernel (search + refinement)
__global__ void me_publish_kernel(volatile uint32_t* done_flag, uint32_t seq) {
  __threadfence_system();
  *done_flag = seq;
  __threadfence_system();
}

// jobs [0, tail_first) are cut into n_strips strips, jobs from tail_first on into tail_strips (the last, partial round of
// workgroups of a launch is dealt in finer pieces, hmme.hip plan_tail)
__global__ void me_prep_jobs16_kernel(MeJob16* jobs, int* first_strip_of_job, const int16_t* __restrict__ pred_q,
                                      int ctu_first, int ctu_count, int n_refs, int pic_w, int pic_h, int sr, int n_strips_head, int rows_max,
                                      int tail_first, int tail_strips) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ctu_count * n_refs) return;
  const int r = i / ctu_count;
  const int ctus_x = (pic_w + 63) >> 6, n_ctu = ctus_x * ((pic_h + 63) >> 6);
  const int ctu = ctu_first + (i - r * ctu_count);
  const int cu_x = (ctu % ctus_x) * 64, cu_y = (ctu / ctus_x) * 64;
  const long pq = 2 * ((long)r * n_ctu + ctu);
  const int px = pred_q ? pred_q[pq] : 0, py = pred_q ? pred_q[pq + 1] : 0;
  int ltx, lty, rbx, rby;
  set_search_range(px, py, sr, cu_x, cu_y, pic_w, pic_h, ltx, lty, rbx, rby);
  MeJob j;
  j.ctu_x = (int16_t)(cu_x | r); j.ctu_y = (int16_t)cu_y;
  j.lt_x = (int16_t)ltx; j.lt_y = (int16_t)lty; j.rb_x = (int16_t)rbx; j.rb_y = (int16_t)rby;
  j.pred_x = (int16_t)px; j.pred_y = (int16_t)py;
  const int wy = rby - lty + 1;
  const int n_strips = i < tail_first ? n_strips_head : tail_strips;
  const int base = i < tail_first ? i * n_strips_head : tail_first * n_strips_head + (i - tail_first) * tail_strips;
  const int n_units = tail_first * n_strips_head + (ctu_count * n_refs - tail_first) * tail_strips;   // workgroups of the launch
  first_strip_of_job[i] = me_xcd_position(base, n_units);
  // strips of the height me_strip_rows16 picks for this window, the last one takes the rest; strips beyond the window (clipped
  // windows need fewer) are empty: y0 == y1
  // head jobs: the planner's choice within n_strips; tail jobs: exactly tail_strips equal pieces (the host chose that number for
  // the sake of the launch's last round, not for this window)
  const int h = i < tail_first ? me_strip_rows16(rbx - ltx + 1, wy, rows_max, n_strips) : max(1, (wy + n_strips - 1) / n_strips);
  for (int s = 0; s < n_strips; ++s) {
    MeJob16 js;
    js.j = j;
    js.y0 = (int16_t)min(wy, s * h);
    js.y1 = (int16_t)min(wy, (s + 1) * h);
    js.job = i;
    // XCD-aware order (me_xcd_position): the strips of a CTU, whose window rows overlap by 63, and the CTUs next to it run on ONE
    // XCD, one after the other; every XCD gets all strips of its CTUs, so strips of unequal height cannot pile up on one XCD (round 2
    // rotated the strip order per job for that)
    jobs[me_xcd_position(base + s, n_units)] = js;
  }
}

// 8-bit segment mode (me_search_kernel SPLIT = 2): the jobs' units (kSegUnit tasks) as one list, cut into n_wg equal segments.  ONE workgroup of
// kSegPrepThreads threads writes the whole table (n_jobs <= kSegPrepThreads: a tail is shorter than one round of the chip's workgroup slots):
// a job and its unit count per thread, an inclusive scan in LDS, then the segments -- segment s owns units [N s / n_wg, N (s + 1) / n_wg)
// and starts in the job whose range holds its first unit (binary search in the prefix sums).
constexpr int kSegPrepThreads = 512;
// the head of a segment launch: jobs 0 .. n_head - 1, one workgroup each, in the XCD-aware order of a whole-job launch (me_xcd_unit)
__global__ void me_prep_whole_segments_kernel(void* table, int* first_strip_of_job, const int16_t* __restrict__ pred_q, int ctu_first, int ctu_count,
                                              int n_refs, int pic_w, int pic_h, int sr, int n_wg, int n_head) {
  const int li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= n_head) return;
  const int r = li / ctu_count;
  const int ctus_x = (pic_w + 63) >> 6, n_ctu = ctus_x * ((pic_h + 63) >> 6);
  const int ctu = ctu_first + (li - r * ctu_count);
  const int cu_x = (ctu % ctus_x) * 64, cu_y = (ctu / ctus_x) * 64;
  const long pq = 2 * ((long)r * n_ctu + ctu);
  const int px = pred_q ? pred_q[pq] : 0, py = pred_q ? pred_q[pq + 1] : 0;
  int ltx, lty, rbx, rby;
  set_search_range(px, py, sr, cu_x, cu_y, pic_w, pic_h, ltx, lty, rbx, rby);
  MeJob16 js;
  js.j.ctu_x = (int16_t)(cu_x | r); js.j.ctu_y = (int16_t)cu_y;
  js.j.lt_x = (int16_t)ltx; js.j.lt_y = (int16_t)lty; js.j.rb_x = (int16_t)rbx; js.j.rb_y = (int16_t)rby;
  js.j.pred_x = (int16_t)px; js.j.pred_y = (int16_t)py;
  js.y0 = 0; js.y1 = 0x7fff; js.job = li;
  ((MeJob16*)me_seg_table_jobs(table, n_wg))[li] = js;
  first_strip_of_job[li] = li;
  ((int4*)me_seg_table_segs(table))[me_xcd_position(li, n_head)] = make_int4(li, 0, 0, -1);
}
// the tail: the launch's jobs job0 .. job0 + n_jobs - 1 as entries idx0 .. of the table's jobs and as its segments idx0 .. n_wg - 1.
// idx0 = job0: the head's jobs are entries / workgroups 0 .. job0 - 1 of the same table (one launch); idx0 = 0: a table of the tail alone
__global__ void __launch_bounds__(kSegPrepThreads)
me_prep_segments_kernel(void* table, int* first_strip_of_job, const int16_t* __restrict__ pred_q, int ctu_first, int ctu_count, int n_refs,
                        int pic_w, int pic_h, int sr, int n_wg, int job0, int n_jobs, int idx0) {
  __shared__ int scan[kSegPrepThreads + 1];
  const int li = threadIdx.x;
  int nt = 0;
  if (li < n_jobs) {
    const int i = job0 + li;              // which (CTU, reference): the launch's job i
    const int r = i / ctu_count;
    const int ctus_x = (pic_w + 63) >> 6, n_ctu = ctus_x * ((pic_h + 63) >> 6);
    const int ctu = ctu_first + (i - r * ctu_count);
    const int cu_x = (ctu % ctus_x) * 64, cu_y = (ctu / ctus_x) * 64;
    const long pq = 2 * ((long)r * n_ctu + ctu);
    const int px = pred_q ? pred_q[pq] : 0, py = pred_q ? pred_q[pq + 1] : 0;
    int ltx, lty, rbx, rby;
    set_search_range(px, py, sr, cu_x, cu_y, pic_w, pic_h, ltx, lty, rbx, rby);
    MeJob16 js;
    js.j.ctu_x = (int16_t)(cu_x | r); js.j.ctu_y = (int16_t)cu_y;
    js.j.lt_x = (int16_t)ltx; js.j.lt_y = (int16_t)lty; js.j.rb_x = (int16_t)rbx; js.j.rb_y = (int16_t)rby;
    js.j.pred_x = (int16_t)px; js.j.pred_y = (int16_t)py;
    nt = me_num_tasks(rbx - ltx + 1, rby - lty + 1);
    js.y0 = 0; js.y1 = (int16_t)nt; js.job = idx0 + li;
    nt = (nt + kSegUnit - 1) / kSegUnit;   // from here on: units
    ((MeJob16*)me_seg_table_jobs(table, n_wg))[idx0 + li] = js;
    first_strip_of_job[idx0 + li] = idx0 + li;   // me_finalize16_kernel: entry e decodes against jobs[e]
  }
  scan[li + 1] = nt;
  if (li == 0) scan[0] = 0;
  __syncthreads();
  for (int d = 1; d < kSegPrepThreads; d <<= 1) {   // inclusive scan over scan[1 ..]
    const int v = li + 1 > d ? scan[li + 1 - d] : 0;
    __syncthreads();
    scan[li + 1] += v;
    __syncthreads();
  }
  int* prefix = (int*)me_seg_table_prefix(table, n_wg, idx0 + n_jobs);
  if (li <= n_jobs) prefix[li] = scan[li];
  const long total = scan[n_jobs];
  int4* segs = (int4*)me_seg_table_segs(table) + idx0;
  const int n_seg = n_wg - idx0;
  for (int s = li; s < n_seg; s += kSegPrepThreads) {
    const int g0 = (int)(total * s / n_seg), g1 = (int)(total * (s + 1) / n_seg);
    int lo = 0, hi = n_jobs;                         // the job j with scan[j] <= g0 < scan[j + 1] (every job has at least one task)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (scan[mid] <= g0) lo = mid; else hi = mid;
    }
    segs[s] = make_int4(idx0 + lo, g0, g1, idx0);
  }
}

// 8-bit planes, search range 65..128: four tile jobs per (CTU, reference), each a complete search of its sub-window through the
// SPLIT kernel; tile (0,0) comes first and keeps the window's own top-left, which me_finalize16_kernel decodes against
__global__ void me_prep_jobs_tile_kernel(MeJob16* jobs, int* first_strip_of_job, const int16_t* __restrict__ pred_q,
                                         int ctu_first, int ctu_count, int n_refs, int pic_w, int pic_h, int sr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ctu_count * n_refs) return;
  const int r = i / ctu_count;
  const int ctus_x = (pic_w + 63) >> 6, n_ctu = ctus_x * ((pic_h + 63) >> 6);
  const int ctu = ctu_first + (i - r * ctu_count);
  const int cu_x = (ctu % ctus_x) * 64, cu_y = (ctu / ctus_x) * 64;
  const long pq = 2 * ((long)r * n_ctu + ctu);
  const int px = pred_q ? pred_q[pq] : 0, py = pred_q ? pred_q[pq + 1] : 0;
  int ltx, lty, rbx, rby;
  set_search_range(px, py, sr, cu_x, cu_y, pic_w, pic_h, ltx, lty, rbx, rby);
  first_strip_of_job[i] = i * 4;
  for (int t = 0; t < 4; ++t) {
    const int tx = t & 1, ty = t >> 1;
    const int x0 = ltx + tx * kTileStep, y0 = lty + ty * kTileStep;
    const bool empty = x0 > rbx || y0 > rby;
    MeJob16 js;
    js.j.ctu_x = (int16_t)(cu_x | r); js.j.ctu_y = (int16_t)cu_y;
    js.j.lt_x = (int16_t)(empty ? ltx : x0); js.j.lt_y = (int16_t)(empty ? lty : y0);
    js.j.rb_x = (int16_t)min(rbx, js.j.lt_x + kTileStep - 1); js.j.rb_y = (int16_t)min(rby, js.j.lt_y + kTileStep - 1);
    js.j.pred_x = (int16_t)px; js.j.pred_y = (int16_t)py;
    js.y0 = 0; js.y1 = empty ? 0 : 0x7fff;   // task range: everything, or nothing for a tile the clipped window does not reach
    js.job = i | tx << 30 | ty << 29;
    jobs[i * 4 + t] = js;
  }
}


// ---- fractional-pel refinement of the integer winners ---------------------------------------------------------
// The step after the path: TEncSearch::xPatternSearchFracDIF (reference TEncSearch.cpp:4294-4331) for all 593
// slots of a CTU -- half-pel then quarter-pel refinement around each slot's integer MV with HM's 8-tap luma
// interpolation (TComInterpolationFilter.cpp:57-63, :170-260) and Hadamard (xGetHADs, TComRdCost.cpp:1537-1604)
// or SAD distortion.  Work item = one 4x4 block of one DISTINCT (position, MV) pair of the CTU ("Work sharing" below: at most 6 144
// per CTU and stage, far fewer where slots share their motion); one lane per item.  An 8x8 Hadamard is assembled from the four 4x4
// transforms of its quadrants (H8 = [[H4, H4], [H4, -H4]]) with two quad_perm DPP butterflies, so slots whose size is a multiple of 8
// (8x8 Hadamard blocks) and the others (4x4 blocks) run the same code.  Per-slot distortions accumulate in LDS (ds_add_u32); 593
// threads then add the MV cost and pick the winner in HM's point order (strict '<', tables TEncSearch.cpp:51-75).
// per-slot distortion sums of the nine refinement points
// BPS = bytes per sample of the planes (1: 8-bit video, 2: 9..12 bit)
// 8-bit planes: the nine sums of a slot are three 64-bit words of three 21-bit fields each (points 3k, 3k + 1, 3k + 2 of word k), added
// with ds_add_u64 -- a third of the LDS atomics, which on content whose slots share their motion were 40 % of an item's time (up to sixteen
// lanes of a wave add to the same large slot: profiles/r05k_frac_phases_before_after.txt).  A field never carries into the next: the largest sum is the
// 64x64 slot's, 64 8x8 Hadamard blocks of at most 8 * 64 * 255 / 4 = 32 640 each (Parseval) = 2 088 960 < 2^21; SAD 64 * 64 * 255.
// Wider samples (and the biased ones of bi-prediction origins) keep nine 32-bit sums.
#ifndef ME_FRAC_PACK3
#define ME_FRAC_PACK3 1
#endif
constexpr bool frac_pack3(int bps) { return ME_FRAC_PACK3 && bps == 1; }
constexpr int frac_acc_row(int bps) { return frac_pack3(bps) ? 6 : 9; }   // dwords per slot
constexpr int frac_acc_dw(int bps) { return (593 * frac_acc_row(bps) + 15) & ~15; }
static_assert(64 * 32640 < (1 << 21) && 4096 * 255 < (1 << 21), "me_frac_kernel: a packed sum field holds the largest 8-bit slot sum");
constexpr int frac_threads(int bps) { return 256; }
// sums [593][9] | slot states | tap tables, counters | the two work lists | current block | cover table (uint16 [64][18] + [256][6])
#ifndef ME_FRAC_T_LDS_PAD   // timing-only: LDS bytes a workgroup asks for beyond its need (occupancy experiments)
#define ME_FRAC_T_LDS_PAD 0
#endif
// 8-bit planes: + the patch rows of each lane's NEXT item, written by LDS-DMA while the current item is evaluated (me_frac_stage):
// 12 rows x 16 B x 256 lanes = 48 KiB.  80 KiB a workgroup, two workgroups a CU -- which is what the kernel's registers allow anyway
#ifdef ME_FRAC_RIDE_ANY
#define ME_FRAC_RIDE_LISTED 1
#else
#define ME_FRAC_RIDE_LISTED 0
#endif
// ME_FRAC_TREE=1: the implicit items of a wave whose positions share one key add their sums over groups of positions, one add per (slot,
// group), instead of every lane adding to every slot of its position (me_frac_tree_add).  Built and measured in round 6, left OFF:
// on content whose slots share their motion it took the kernel's LDS bank-conflict cycles from 0.63 to 0.19 of its LDS cycles and its LDS
// activity from 0.19 to 0.10 of the CU cycles -- and its time from 0.133 to 0.136 ms per 2160p pair, because those conflicts never were what
// the launch waited for: the exchanges, the packing and the designated lanes' slot arithmetic are 12 % more VALU instructions in a kernel
// whose time follows its instruction count.  10-bit Hadamard gained 3.6 % (nine 32-bit adds per slot saved instead of three 64-bit ones),
// but the u16 SAD variants went past 256 VGPRs with it (55..96 spilled dwords).  profiles/r06h_frac_tree_ab.txt
#ifndef ME_FRAC_TREE
#define ME_FRAC_TREE 0
#endif
#ifndef ME_FRAC_GLDS   // measured: no faster than plain loads on any content (profiles/r05j_frac_glds_ab.txt) -- the items do not wait for their rows
#define ME_FRAC_GLDS 0
#endif
constexpr bool frac_glds(int bps) { return ME_FRAC_GLDS && bps == 1; }
constexpr int frac_pf_dw(int bps) { return frac_glds(bps) ? 12 * 4 * 256 : 0; }
constexpr size_t frac_lds_bytes(int bps) { return (size_t)(frac_acc_dw(bps) + 600 + 160 + (64 * 18 + 256 * 6) / 2 + 1024 * bps + (64 * 18 + 256 * 6) / 2 + frac_pf_dw(bps)) * 4 + ME_FRAC_T_LDS_PAD; }

__device__ __forceinline__ uint32_t me_mv_cost_q(uint32_t lambda_q16, int vx_q, int vy_q, int pred_x, int pred_y) {
  return (lambda_q16 * (me_component_bits(vx_q - pred_x) + me_component_bits(vy_q - pred_y))) >> 16;
}

// luma taps of fraction f (0..3), tap t (0..7): TComInterpolationFilter::m_lumaFilter
__host__ __device__ constexpr int me_luma_tap(int f, int t) {
  constexpr int k1[8] = {-1, 4, -10, 58, 17, -5, 1, 0}, k2[8] = {-1, 4, -11, 40, 40, -11, 4, -1}, k3[8] = {0, 1, -5, 17, 58, -10, 4, -1};
  const int k0 = t == 3 ? 64 : 0;
  return f == 0 ? k0 : (f == 1 ? k1[t] : (f == 2 ? k2[t] : k3[t]));
}
// the 8 taps of quarter position q (-3..3, relative to the integer MV) as a 9-tap window that starts one sample
// earlier when the integer part of q is -1: tap j (0..8) multiplies patch sample (output column + j)
__host__ __device__ constexpr int me_tap9(int q, int j) {
  const int t = j - ((q >> 2) + 1);
  return (t < 0 || t > 7) ? 0 : me_luma_tap(q & 3, t);
}
// horizontal taps packed for the dot-product instructions: dword k of the tap window of output column c over the
// 12-sample patch row (BPS 1: four i8 per dword, v_dot4c_i32_i8; BPS 2: two i16 per dword, v_dot2c_i32_i16)
template <int BPS>
__host__ __device__ constexpr uint32_t me_htap_dw(int q, int c, int k) {
  uint32_t w = 0;
  for (int i = 0; i < 4 / BPS; ++i) {
    const int j = (4 / BPS) * k + i - c;
    const int t = (j < 0 || j > 8) ? 0 : me_tap9(q, j);
    w |= BPS == 1 ? (uint32_t)(t & 0xff) << (8 * i) : (uint32_t)(t & 0xffff) << (16 * i);
  }
  return w;
}
// Quarter-pel stage: the three offsets of a component around its half-pel winner h (-1, 0, +1 half samples) all fit ONE 8-sample
// window: h = -1 -> q = -3..-1, integer part -1: samples -4..+3; h = 0 -> q = -1..1: samples -3..+4 (q = -1 is fraction 3 at integer
// part -1, whose first tap is 0, so its seven live taps start at sample -3 as well); h = +1 -> q = 1..3: samples -3..+4.  The lane
// fetches its patch from that window's first sample on -- 11 x 11 samples instead of 12 x 12, 8 taps instead of a 9-tap window with a
// zero at one end.  me_tap8(h3, d3, j): tap of window sample j (0..7) for winner h3 - 1 and offset d3 - 1 (both 0..2).
__host__ __device__ constexpr int me_win8_first(int h3) { return h3 == 0 ? -4 : -3; }   // first window sample relative to the output sample
__host__ __device__ constexpr int me_tap8(int h3, int d3, int j) {
  const int q = 2 * (h3 - 1) + (d3 - 1);
  const int t = j + me_win8_first(h3) - ((q >> 2) - 3);   // tap index: sample = out + (q >> 2) - 3 + t
  return (t < 0 || t > 7) ? 0 : me_luma_tap(q & 3, t);
}
static_assert(me_luma_tap(3, 0) == 0, "the tap q = -1 loses to the shared window must be zero");
// dword k of the packed 8-tap window (BPS 1: four i8 per dword; BPS 2: two i16)
template <int BPS>
__host__ __device__ constexpr uint32_t me_htap8_dw(int h3, int d3, int k) {
  uint32_t w = 0;
  for (int i = 0; i < 4 / BPS; ++i) {
    const int j = (4 / BPS) * k + i;
    const int t = j < 8 ? me_tap8(h3, d3, j) : 0;
    w |= BPS == 1 ? (uint32_t)(t & 0xff) << (8 * i) : (uint32_t)(t & 0xffff) << (16 * i);
  }
  return w;
}
constexpr int kFracTabH = 8, kFracTabV = 8;   // LDS tap tables: 9 rows (winner x offset) of packed horizontal / float vertical taps
constexpr int kFracRows1 = 11;                // patch rows (and samples per row) of the quarter-pel stage

#define ME_FRAC_BFLY(R, T, PERM) "v_fmac_f32_dpp " #R ", " #R ", " #T " quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ float me_dpp_f(float v, const int ctrl_b1) {
  return ctrl_b1 ? __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false))
                 : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));
}

// max(|a|, |b|) in ONE instruction: fmaxf() of two fabsf() compiles to three (llvm.maxnum quiets signalling NaNs first: a
// v_max_f32 x, |x|, |x| per operand in IEEE mode); no NaN ever reaches these
__device__ __forceinline__ float me_absmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// distortion of one 4x4 difference block d (row-major): SAD, or the Hadamard sum (xCalcHADs4x4; with KIND8 the quad's four 4x4
// transforms are combined into the 8x8 transform, xCalcHADs8x8, and all four lanes return the block's value)
// want4 (wave-uniform, KIND8 only): also return in `own4` what this lane's 4x4 block alone contributes to a slot made of 4x4 blocks
// (the AMP shapes, 8x4, 4x8) -- the same difference block, the same 4x4 transform, only the 8x8 combination left out
template <int HAD, int KIND8>
__device__ __forceinline__ uint32_t me_frac_dist(const float (&d)[16], float s1, float s2, bool want4, uint32_t& own4) {
  uint32_t contrib;
  if (!HAD) {
    float sad = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sad += __builtin_fabsf(d[i]);
    if (KIND8) {   // the quad's four 4x4 SADs belong to the same slots: hand out their sum
      own4 = (uint32_t)sad;
      sad += me_dpp_f(sad, 1);
      sad += me_dpp_f(sad, 0);
    }
    contrib = (uint32_t)sad;
  } else {
    float m[16];
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // 4x4 Walsh-Hadamard: rows, then columns (xCalcHADs4x4)
      const float a = d[4 * r] + d[4 * r + 3], b = d[4 * r + 1] + d[4 * r + 2], e = d[4 * r + 1] - d[4 * r + 2], f = d[4 * r] - d[4 * r + 3];
      m[4 * r] = a + b; m[4 * r + 1] = a - b; m[4 * r + 2] = f + e; m[4 * r + 3] = f - e;
    }
    // columns, first level; z = {a, b, e, f}[k].  The second level would be a + b, a - b, f + e, f - e -- but only the sum of the
    // absolute coefficients is wanted and |a + b| + |a - b| = 2 max(|a|, |b|): the level and its sixteen absolute adds become eight
    // v_max_f32 (|.| is a source modifier) and eight adds of HALF the sum.  The levels of a Walsh-Hadamard transform commute, so with
    // KIND8 the two levels across the quad run BEFORE that last one -- and the lane's own 4x4 sum (want4) is the same eight maxima
    // taken before them.
    float z[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      z[k] = m[k] + m[12 + k]; z[4 + k] = m[4 + k] + m[8 + k]; z[8 + k] = m[4 + k] - m[8 + k]; z[12 + k] = m[k] - m[12 + k];
    }
    float sum = 0.f;
    if (KIND8) {   // combine the quad's four 4x4 transforms into the 8x8 transform (xCalcHADs8x8)
      if (want4) {   // xCalcHADs4x4 of this block, as the 4x4 kind computes it: ((2 * s4) + 1) >> 1
        float s4 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float t = me_absmax(z[k], z[4 + k]) + me_absmax(z[8 + k], z[12 + k]); s4 = k ? s4 + t : t; }
        own4 = (uint32_t)s4;
      }
      // two butterflies across the quad, each one v_fmac_f32 with a DPP source: z += t * z[neighbour], t = +-1 by
      // role.  Odd roles hold the negated difference, which the absolute sum does not see.  One asm block keeps
      // 16 instructions between a register's write and its DPP read (the hazard the assembler does not pad for).
      asm("s_nop 1\n\t"
          ME_FRAC_BFLY(%0, %16, "[1,0,3,2]") ME_FRAC_BFLY(%1, %16, "[1,0,3,2]") ME_FRAC_BFLY(%2, %16, "[1,0,3,2]") ME_FRAC_BFLY(%3, %16, "[1,0,3,2]")
          ME_FRAC_BFLY(%4, %16, "[1,0,3,2]") ME_FRAC_BFLY(%5, %16, "[1,0,3,2]") ME_FRAC_BFLY(%6, %16, "[1,0,3,2]") ME_FRAC_BFLY(%7, %16, "[1,0,3,2]")
          ME_FRAC_BFLY(%8, %16, "[1,0,3,2]") ME_FRAC_BFLY(%9, %16, "[1,0,3,2]") ME_FRAC_BFLY(%10, %16, "[1,0,3,2]") ME_FRAC_BFLY(%11, %16, "[1,0,3,2]")
          ME_FRAC_BFLY(%12, %16, "[1,0,3,2]") ME_FRAC_BFLY(%13, %16, "[1,0,3,2]") ME_FRAC_BFLY(%14, %16, "[1,0,3,2]") ME_FRAC_BFLY(%15, %16, "[1,0,3,2]")
          ME_FRAC_BFLY(%0, %17, "[2,3,0,1]") ME_FRAC_BFLY(%1, %17, "[2,3,0,1]") ME_FRAC_BFLY(%2, %17, "[2,3,0,1]") ME_FRAC_BFLY(%3, %17, "[2,3,0,1]")
          ME_FRAC_BFLY(%4, %17, "[2,3,0,1]") ME_FRAC_BFLY(%5, %17, "[2,3,0,1]") ME_FRAC_BFLY(%6, %17, "[2,3,0,1]") ME_FRAC_BFLY(%7, %17, "[2,3,0,1]")
          ME_FRAC_BFLY(%8, %17, "[2,3,0,1]") ME_FRAC_BFLY(%9, %17, "[2,3,0,1]") ME_FRAC_BFLY(%10, %17, "[2,3,0,1]") ME_FRAC_BFLY(%11, %17, "[2,3,0,1]")
          ME_FRAC_BFLY(%12, %17, "[2,3,0,1]") ME_FRAC_BFLY(%13, %17, "[2,3,0,1]") ME_FRAC_BFLY(%14, %17, "[2,3,0,1]") ME_FRAC_BFLY(%15, %17, "[2,3,0,1]")
          : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]), "+v"(z[8]), "+v"(z[9]),
            "+v"(z[10]), "+v"(z[11]), "+v"(z[12]), "+v"(z[13]), "+v"(z[14]), "+v"(z[15])
          : "v"(s1), "v"(s2));
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float t = me_absmax(z[k], z[4 + k]) + me_absmax(z[8 + k], z[12 + k]); sum = k ? sum + t : t; }
      sum += sum;
      sum += me_dpp_f(sum, 1);
      sum += me_dpp_f(sum, 0);
      contrib = ((uint32_t)sum + 2) >> 2;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float t = me_absmax(z[k], z[4 + k]) + me_absmax(z[8 + k], z[12 + k]); sum = k ? sum + t : t; }
      contrib = (uint32_t)sum;   // (2 * sum + 1) >> 1
    }
  }
  return contrib;
}

#ifndef ME_FRAC_SCALAR_STAGE1   // A/B: the quarter-pel stage one sample per instruction
// The Hadamard sum of a 4x4 difference block that arrives as pairs of horizontal neighbours (P[r][h]: row r, columns 2h, 2h + 1), two
// butterflies per instruction (v_pk_add_f32).  Only the sum of the absolute coefficients is wanted, so the transform runs in natural
// (Sylvester) order -- the same sixteen Walsh functions as xCalcHADs4x4's, in another order and with other signs: three levels pair
// different registers element by element, the one level that pairs the two halves of a register comes last, in scalar code (measured
// against the same level first: 2 % slower; and against this function in the half-pel stage, whose differences are born as single
// registers: 4 % slower, profiles/r04s_frac_instruction_mix_ab.txt).  The 8x8 combination across the quad works on corresponding
// coefficients of the four quadrants, which every lane orders alike.
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND8>
__device__ __forceinline__ uint32_t me_frac_had_pk(const v2f (&P)[4][2], float s1, float s2, bool want4, uint32_t& own4) {
  v2f A[4][2], B[4][2], C[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) { A[r][0] = P[r][0] + P[r][1]; A[r][1] = P[r][0] - P[r][1]; }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    B[0][h] = A[0][h] + A[1][h]; B[1][h] = A[0][h] - A[1][h]; B[2][h] = A[2][h] + A[3][h]; B[3][h] = A[2][h] - A[3][h];
    C[0][h] = B[0][h] + B[2][h]; C[2][h] = B[0][h] - B[2][h]; C[1][h] = B[1][h] + B[3][h]; C[3][h] = B[1][h] - B[3][h];
  }
  // the level that pairs the two halves of a register comes last and is never carried out: |x + y| + |x - y| = 2 max(|x|, |y|)
  // (me_frac_dist); with KIND8 the two levels across the quad run before it, the lane's own 4x4 sum is taken before those
  float z[16];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) { z[4 * r + 2 * h] = C[r][h].x; z[4 * r + 2 * h + 1] = C[r][h].y; }
  float sum = 0.f;
  if (KIND8) {
    if (want4) {
      float s4 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float t = me_absmax(z[2 * i], z[2 * i + 1]); s4 = i ? s4 + t : t; }
      own4 = (uint32_t)s4;
    }
    asm("s_nop 1\n\t"
        ME_FRAC_BFLY(%0, %16, "[1,0,3,2]") ME_FRAC_BFLY(%1, %16, "[1,0,3,2]") ME_FRAC_BFLY(%2, %16, "[1,0,3,2]") ME_FRAC_BFLY(%3, %16, "[1,0,3,2]")
        ME_FRAC_BFLY(%4, %16, "[1,0,3,2]") ME_FRAC_BFLY(%5, %16, "[1,0,3,2]") ME_FRAC_BFLY(%6, %16, "[1,0,3,2]") ME_FRAC_BFLY(%7, %16, "[1,0,3,2]")
        ME_FRAC_BFLY(%8, %16, "[1,0,3,2]") ME_FRAC_BFLY(%9, %16, "[1,0,3,2]") ME_FRAC_BFLY(%10, %16, "[1,0,3,2]") ME_FRAC_BFLY(%11, %16, "[1,0,3,2]")
        ME_FRAC_BFLY(%12, %16, "[1,0,3,2]") ME_FRAC_BFLY(%13, %16, "[1,0,3,2]") ME_FRAC_BFLY(%14, %16, "[1,0,3,2]") ME_FRAC_BFLY(%15, %16, "[1,0,3,2]")
        ME_FRAC_BFLY(%0, %17, "[2,3,0,1]") ME_FRAC_BFLY(%1, %17, "[2,3,0,1]") ME_FRAC_BFLY(%2, %17, "[2,3,0,1]") ME_FRAC_BFLY(%3, %17, "[2,3,0,1]")
        ME_FRAC_BFLY(%4, %17, "[2,3,0,1]") ME_FRAC_BFLY(%5, %17, "[2,3,0,1]") ME_FRAC_BFLY(%6, %17, "[2,3,0,1]") ME_FRAC_BFLY(%7, %17, "[2,3,0,1]")
        ME_FRAC_BFLY(%8, %17, "[2,3,0,1]") ME_FRAC_BFLY(%9, %17, "[2,3,0,1]") ME_FRAC_BFLY(%10, %17, "[2,3,0,1]") ME_FRAC_BFLY(%11, %17, "[2,3,0,1]")
        ME_FRAC_BFLY(%12, %17, "[2,3,0,1]") ME_FRAC_BFLY(%13, %17, "[2,3,0,1]") ME_FRAC_BFLY(%14, %17, "[2,3,0,1]") ME_FRAC_BFLY(%15, %17, "[2,3,0,1]")
        : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]), "+v"(z[8]), "+v"(z[9]),
          "+v"(z[10]), "+v"(z[11]), "+v"(z[12]), "+v"(z[13]), "+v"(z[14]), "+v"(z[15])
        : "v"(s1), "v"(s2));
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float t = me_absmax(z[2 * i], z[2 * i + 1]); sum = i ? sum + t : t; }
    sum += sum;
    sum += me_dpp_f(sum, 1);
    sum += me_dpp_f(sum, 0);
    return ((uint32_t)sum + 2) >> 2;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { const float t = me_absmax(z[2 * i], z[2 * i + 1]); sum = i ? sum + t : t; }
  return (uint32_t)sum;   // (2 * sum + 1) >> 1
}
#endif

// The evaluation of one work item.  Stage 0 (me_frac_eval0): the nine half-pel points around the integer MV; stage 1 (me_frac_eval1):
// the eight quarter-pel points around the slot's half-pel winner.  P: patch rows of 3 * BPS dwords, 8-bit samples XORed with 0x80
// (signed bytes p - 128: the 128 * 64 this removes IS the -8192 offset of the first pass).  orgM: current samples + kRoundMagic.
// KIND8: the lane is one quadrant (role 0..3 = TL, TR, BL, BR) of an 8x8 Hadamard block; all four lanes of the quad return the
// block's distortion.  out[point]: distortion of the refinement points in HM's point order (s_acMvRefineH / Q, TEncSearch.cpp:51-75).
// Arithmetic: first pass in integer dot products (v_dot4_i32_i8 / v_dot2c_i32_i16, shift bd-8), second pass, rounding, clipping
// and the Hadamard transform in fp32 -- every value is an integer (or an integer + a fraction of a few bits) below 2^23, so fp32 is
// exact, the order of summation is free, and v_fmac_f32 issues at the full VALU rate where v_mad_i32_i24 does not
// (tools/ubench/valu_rates3).  bd = bit depth of the video (8 when BPS == 1): the two passes shift by bd-8 and 20-bd around the
// 14-bit intermediate (TComInterpolationFilter.cpp:170-212: headRoom = 14 - bd).  Second pass: floor((S + 2^(sh2-1) + (8192 << 6))
// >> sh2) = nearest integer of (S + 524288 + 0.5) * 2^-sh2 (never a tie).  clip_lo: 0, or the bias 2^bd that block and window of a
// bi-prediction origin carry (hmme.hip ctu_call): the taps sum to 64, so a bias B = 2^bd passes both filter stages exactly
// (64 * B >> (bd - 8) = 2^14, 64 * 2^14 >> (20 - bd) = B): the predicted sample comes out as pred + B, is clipped to [B, B + maxv]
// and meets a current sample that carries the same B.
// explicit weighted prediction in the refinement (xGetHADsw / xGetSADw, TComRdCostWeightPrediction.cpp:407-470, :55-90): the interpolated,
// clipped prediction p is weighted sample by sample, pred = ((w0 * p + round) >> shift) + offset, before the difference is taken.
// ws = w0 * 2^-shift, rs = round * 2^-shift: fma(ws, p, rs) IS (w0 * p + round) / 2^shift exactly (|w0 * p + round| < 2^24), v_floor_f32
// the shift; org_sub = what to take off a staged current sample: its staging bias + offset.  WP = 0: none of this is compiled in.
struct FracWp { float ws, rs, org_sub; };
// The item's sixteen current samples (+ kRoundMagic) as the evaluation reads them, row by row: sixteen registers.  (Round 6 put them into a
// block of LDS for the u16 variants, which sit at the 256-VGPR limit and spill 5..9 dwords: nothing spilled any more and 8-10 % slower --
// 24-40 B of scratch per lane cost less than 36 more LDS reads per item.  profiles/r06h_frac_tree_ab.txt)
struct FracOrgRegs {
  const float (&m)[16];
  __device__ __forceinline__ float4 row(int r) const { return make_float4(m[4 * r], m[4 * r + 1], m[4 * r + 2], m[4 * r + 3]); }
};
// a whole-picture launch derives each job's window itself (what me_prep_jobs_kernel writes into a job table: pair, CTU, predictor,
// xSetSearchRange + clipMv) -- one kernel launch less per refinement; the per-CTU call hands over the job the host prepared
struct FracPrep { const int16_t* pred_q; uint32_t ctus, dims; int sr; };   // ctus: ctu_first | ctu_count << 16; dims: width | height << 16 (few scalar registers: they stay live across the kernel)
constexpr float kRoundMagic = 12582912.0f;   // 1.5 * 2^23: x + magic rounds x to the nearest integer (ties to even)
// First pass on 8-bit planes without v_cvt_f32_i32: the dot-product chain of a filtered sample starts from the BIT PATTERN of
// kRoundMagic (0x4B400000; ulp 1, so adding an integer k, |k| < 2^22, to the pattern gives the pattern of kRoundMagic + k -- here
// |k| <= 128 * 112) and one v_sub_f32 takes the magic off again: v_dot4_i32_i8 (the three-operand form, addend = the magic) +
// v_sub_f32 in place of v_mov_b32 0 + v_dot4c + v_cvt_f32_i32 (13.6 -> 7.7 issue cycles per sample, profiles/r01d_ubench_valu_rates3.txt).
// (clamp = 1 keeps the instruction in its three-operand form -- the accumulating v_dot4c has no clamp bit -- and never clamps here)
__device__ __forceinline__ int me_dot4_magic(uint32_t p, uint32_t t) { return __builtin_amdgcn_sdot4((int)p, (int)t, 0x4B400000, true); }
// STAGE 1 as the kernel runs it: the eight quarter-pel points around the half-pel winner (h3x - 1, h3y - 1) from the 11 x 11 patch that
// starts at the shared window's first sample (me_tap8): HM's two filter passes sample for sample -- the taps left out are zeros.
// P: 11 rows x 11 samples; tab_h / tab_v: [h3 * 3 + offset] rows of 8 taps.
template <int HAD, int BPS, int KIND8, int WP, class Org>
__device__ __forceinline__ void me_frac_eval1(const uint32_t (&P)[kFracRows1][3 * BPS], const Org orgM, int h3x, int h3y, int role, int bd,
                                              float clip_lo, const uint32_t* tab_h, const float* tab_v, bool want4, const FracWp wp,
                                              uint32_t (&out)[9], uint32_t (&out4)[9]) {
  constexpr int PW = 3 * BPS;
  constexpr int idxQ[3][3] = {{3, 1, 4}, {5, 0, 6}, {7, 2, 8}};   // [dy+1][dx+1], s_acMvRefineQ order (reference TEncSearch.cpp:64-75)
  const float s1 = (role & 1) ? -1.f : 1.f, s2 = (role & 2) ? -1.f : 1.f;
  const int sh1 = BPS == 1 ? 0 : bd - 8;
  const int off1 = BPS == 1 ? 0 : -(8192 << sh1);
  const float sc2 = BPS == 1 ? 1.f / 4096.f : __int_as_float((127 - (20 - bd)) << 23);
  const float maxv = clip_lo + (BPS == 1 ? 255.f : (float)((1 << bd) - 1));
  out[0] = 0; out4[0] = 0;   // the centre IS the half-pel winner: carried over by the slot's winner thread, not evaluated again
#pragma unroll
  for (int dxi = 0; dxi < 3; ++dxi) {
    const uint32_t* row = tab_h + (h3x * 3 + dxi) * kFracTabH;
    float tmp[kFracRows1][4];   // first pass into the 14-bit intermediates
    if constexpr (BPS == 1) {
      // output column c reads bytes c .. c + 7 of the row: the packed taps moved up by c bytes (column 0 stays inside two dwords)
      const uint32_t w0 = row[0], w1 = row[1];
      uint32_t T[4][3];
      T[0][0] = w0; T[0][1] = w1; T[0][2] = 0;
#pragma unroll
      for (int c = 1; c < 4; ++c) {
        T[c][0] = w0 << (8 * c);
        T[c][1] = __builtin_amdgcn_alignbyte(w1, w0, 4 - c);
        T[c][2] = w1 >> (8 * (4 - c));
      }
#pragma unroll
      for (int r = 0; r < kFracRows1; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#ifndef ME_FRAC_CVT_FIRST_PASS
          int a = me_dot4_magic(P[r][0], T[c][0]);
#else
          int a = __builtin_amdgcn_sdot4((int)P[r][0], (int)T[c][0], 0, false);
#endif
          a = __builtin_amdgcn_sdot4((int)P[r][1], (int)T[c][1], a, false);
          if (c > 0) a = __builtin_amdgcn_sdot4((int)P[r][2], (int)T[c][2], a, false);
#ifndef ME_FRAC_CVT_FIRST_PASS
          tmp[r][c] = __int_as_float(a) - kRoundMagic;
#else
          tmp[r][c] = (float)a;
#endif
        }
    } else {
      typedef short v2s __attribute__((ext_vector_type(2)));
      uint32_t w[4], h[5];   // h: the taps moved up by one sample
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = row[k];
      h[0] = w[0] << 16;
#pragma unroll
      for (int k = 1; k < 4; ++k) h[k] = __builtin_amdgcn_alignbyte(w[k], w[k - 1], 2);
      h[4] = w[3] >> 16;
#pragma unroll
      for (int r = 0; r < kFracRows1; ++r) {
        int a0 = off1, a1 = off1, a2 = off1, a3 = off1;   // columns 0..3: samples c .. c + 7 = dwords c / 2 .. (c + 7) / 2
#pragma unroll
        for (int k = 0; k < 4; ++k) a0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, P[r][k]), __builtin_bit_cast(v2s, w[k]), a0, false);
#pragma unroll
        for (int k = 0; k < 5; ++k) a1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, P[r][k]), __builtin_bit_cast(v2s, h[k]), a1, false);
#pragma unroll
        for (int k = 0; k < 4; ++k) a2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, P[r][k + 1]), __builtin_bit_cast(v2s, w[k]), a2, false);
#pragma unroll
        for (int k = 0; k < 5; ++k) a3 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, P[r][k + 1]), __builtin_bit_cast(v2s, h[k]), a3, false);
        tmp[r][0] = (float)(a0 >> sh1); tmp[r][1] = (float)(a1 >> sh1); tmp[r][2] = (float)(a2 >> sh1); tmp[r][3] = (float)(a3 >> sh1);
      }
    }
#pragma unroll
    for (int dyi = 0; dyi < 3; ++dyi) {
      if (dxi == 1 && dyi == 1) continue;
      float cv[8];   // taps * 2^-sh2 (exact): the accumulator is the sample value with its fraction
#pragma unroll
      for (int j = 0; j < 8; ++j) cv[j] = tab_v[(h3y * 3 + dyi) * kFracTabV + j];
      float d[16];
#ifndef ME_FRAC_SCALAR_STAGE1   // A/B: the quarter-pel stage one sample per instruction
      v2f dp[4][2];
      if constexpr (!WP) {   // two columns per instruction: v_pk_fma_f32 / v_pk_add_f32
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float4 org = orgM.row(r);
#pragma unroll
          for (int cp = 0; cp < 2; ++cp) {
            v2f a = {524288.5f * sc2, 524288.5f * sc2};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const v2f t = {tmp[r + j][2 * cp], tmp[r + j][2 * cp + 1]}, cc = {cv[j], cv[j]};
              a = __builtin_elementwise_fma(cc, t, a);
            }
            const v2f y = v2f{__builtin_amdgcn_fmed3f(a.x, clip_lo, maxv), __builtin_amdgcn_fmed3f(a.y, clip_lo, maxv)} + v2f{kRoundMagic, kRoundMagic};
            const v2f dd = (cp ? v2f{org.z, org.w} : v2f{org.x, org.y}) - y;
            d[4 * r + 2 * cp] = dd.x; d[4 * r + 2 * cp + 1] = dd.y;
            dp[r][cp] = dd;
          }
        }
      } else
#endif
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4 org = orgM.row(r);
        const float og[4] = {org.x, org.y, org.z, org.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float a = 524288.5f * sc2;   // second pass (TComInterpolationFilter.cpp:195-212)
#pragma unroll
          for (int j = 0; j < 8; ++j) a = __builtin_fmaf(cv[j], tmp[r + j][c], a);
          const float y = __builtin_amdgcn_fmed3f(a, clip_lo, maxv) + kRoundMagic;   // clip, then round (the bounds are integers)
          d[4 * r + c] = og[c] - (WP ? __builtin_floorf(__builtin_fmaf(wp.ws, y - kRoundMagic, wp.rs)) : y);
        }
      }
      uint32_t own4 = 0;
#ifndef ME_FRAC_SCALAR_STAGE1
      const uint32_t contrib = (HAD && !WP) ? me_frac_had_pk<KIND8>(dp, s1, s2, want4, own4) : me_frac_dist<HAD, KIND8>(d, s1, s2, want4, own4);
#else
      const uint32_t contrib = me_frac_dist<HAD, KIND8>(d, s1, s2, want4, own4);
#endif
      out[idxQ[dyi][dxi]] = contrib;
      out4[idxQ[dyi][dxi]] = own4;
    }
  }
}

// Work sharing.  A kind-8 slot (width and height multiples of 8) is a set of 8x8 Hadamard blocks, any other slot a set
// of 4x4 blocks; each of the 64 8x8 positions of the CTU is covered by kFracCover8 = 18 kind-8 slots (7 partition modes
// at 64 and 32, three at 16, one at 8) and each of the 256 4x4 positions by kFracCover4 = 6 others (the 4 AMP shapes at
// 16, 8x4, 4x8).  Slots that cover a position with the same integer MV (and, in the quarter-pel stage, the same half-pel
// winner) get the same nine distortions from it, so each stage first lists the DISTINCT (position, MV) pairs, evaluates
// those -- one lane per 4x4 block, a quad of lanes per 8x8 block -- and adds the result to every slot that shares it.
// `cover`: uint16 [64][18] then [256][6] slot ids, ascending (built by the host from the slot table).
constexpr int kFracCover8 = 18, kFracCover4 = 6, kFracPairs8 = 64 * kFracCover8, kFracPairs4 = 256 * kFracCover4;
// slot state word: (mx - lt_x) | (my - lt_y) << 9 | (half_x + 1) << 18 | (half_y + 1) << 20; the sharing key of a stage
constexpr uint32_t kFracKey0 = 0x3ffffu, kFracKey1 = 0x3fffffu;

// `src`: window sample (-4,-4) of this CTU in the reference plane, `gpitch` bytes per row.  The 12x12 patch (11x11 in the quarter-pel stage) comes straight
// from global memory: the windows of neighbouring CTUs overlap and stay in L2, and an LDS copy of the window (tried first)
// was no faster while it capped the search range at 64 and the occupancy at one 16-bit workgroup per CU.
// STAGE 0 (the nine half-pel points) with the filters SHARED between the points that use them.  The half-pel sample to the right of
// integer column x is the half-pel sample to the left of x + 1, and likewise for rows: the horizontal half-pel pass is computed for five
// columns (HH[r][k], k = 0..4: the samples left of columns 0..3 and right of column 3) instead of 2 x 4, the vertical half-pel pass for
// five rows of each column (rows above 0..3 and below 3) instead of 2 x 4, and every filtered value is rounded and clipped once, not once
// per point that reads it: 396 second-pass FMAs instead of 816, 165 first-pass dot products instead of 264.  Same values as nine
// separate evaluations bit for bit: every intermediate is exact in fp32 (above), so the order of summation is free.
template <int HAD, int BPS, int KIND8, int WP, class Org>
__device__ __forceinline__ void me_frac_eval0(const uint32_t (&P)[12][3 * BPS], const Org orgM, int role, int bd, float clip_lo,
                                              bool want4, const FracWp wp, uint32_t (&out)[9], uint32_t (&out4)[9]) {
  constexpr int PW = 3 * BPS;
  constexpr int idxH[3][3] = {{5, 1, 6}, {3, 0, 4}, {7, 2, 8}};   // [dy+1][dx+1], s_acMvRefineH order (TEncSearch.cpp:51-75)
  const float s1 = (role & 1) ? -1.f : 1.f, s2 = (role & 2) ? -1.f : 1.f;
  const int sh1 = BPS == 1 ? 0 : bd - 8;
  const int off1 = BPS == 1 ? 0 : -(8192 << sh1);
  const float sc2 = BPS == 1 ? 1.f / 4096.f : __int_as_float((127 - (20 - bd)) << 23);
  const float maxv = clip_lo + (BPS == 1 ? 255.f : (float)((1 << bd) - 1));
  const float init = 524288.5f * sc2;
  float cv[8];                                  // half-pel taps * 2^-sh2
#pragma unroll
  for (int t = 0; t < 8; ++t) cv[t] = (float)me_luma_tap(2, t) * sc2;
  const float c64 = 64.f * sc2;
  // first pass of patch row r with the tap window of (quarter offset q, output column c)
  auto first = [&](int r, int q, int c) -> float {
#ifndef ME_FRAC_CVT_FIRST_PASS
    if constexpr (BPS == 1) {
      int a = 0;
      bool started = false;
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        if (me_htap_dw<BPS>(q, c, k) == 0) continue;
        a = started ? __builtin_amdgcn_sdot4((int)P[r][k], (int)me_htap_dw<BPS>(q, c, k), a, false) : me_dot4_magic(P[r][k], me_htap_dw<BPS>(q, c, k));
        started = true;
      }
      return __int_as_float(a) - kRoundMagic;
    }
#endif
    int a = off1;
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      if (me_htap_dw<BPS>(q, c, k) == 0) continue;
      if constexpr (BPS == 1) {
        a = __builtin_amdgcn_sdot4((int)P[r][k], (int)me_htap_dw<BPS>(q, c, k), a, false);
      } else {
        typedef short v2s __attribute__((ext_vector_type(2)));
        a = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, P[r][k]), __builtin_bit_cast(v2s, me_htap_dw<BPS>(q, c, k)), a, false);
      }
    }
    return (float)(BPS == 1 ? a : a >> sh1);
  };
  auto clipround = [&](float a) -> float {
    const float y = __builtin_amdgcn_fmed3f(a, clip_lo, maxv) + kRoundMagic;
    return WP ? __builtin_floorf(__builtin_fmaf(wp.ws, y - kRoundMagic, wp.rs)) : y;   // WP: orgM carries no magic either (me_frac_compute)
  };
  // one point: d = org - pred over the 4x4 block, pred(r, c) = y[r0 + r][c0 + c]
#define ME_FRAC_POINT(Y, R0, C0, DYI, DXI)                                                   \
  {                                                                                          \
    float d[16];                                                                             \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                          \
      const float4 org_ = orgM.row(r);                                                       \
      const float og_[4] = {org_.x, org_.y, org_.z, org_.w};                                 \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) d[4 * r + c] = og_[c] - Y[(R0) + r][(C0) + c]; \
    }                                                                                        \
    uint32_t own4 = 0;                                                                       \
    out[idxH[DYI][DXI]] = me_frac_dist<HAD, KIND8>(d, s1, s2, want4, own4);                  \
    out4[idxH[DYI][DXI]] = own4;                                                             \
  }
  // ---- half-pel columns: HH[r][k] = horizontal half-pel sample left of column k (k = 0..3) / right of column 3 (k = 4)
  {
    float HH[12][5];
#pragma unroll
    for (int r = 0; r < 12; ++r) {
#pragma unroll
      for (int k = 0; k < 4; ++k) HH[r][k] = first(r, -2, k);
      HH[r][4] = first(r, 2, 3);
    }
    float yG[5][5], yZ[4][5];                     // vertical half-pel rows above 0..3 and below 3 / the integer rows
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        float a = init;
#pragma unroll
        for (int t = 0; t < 8; ++t) a = __builtin_fmaf(cv[t], HH[j + t][k], a);
        yG[j][k] = clipround(a);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) yZ[r][k] = clipround(__builtin_fmaf(c64, HH[r + 4][k], init));
    }
    ME_FRAC_POINT(yG, 0, 0, 0, 0) ME_FRAC_POINT(yG, 0, 1, 0, 2) ME_FRAC_POINT(yG, 1, 0, 2, 0) ME_FRAC_POINT(yG, 1, 1, 2, 2)
    ME_FRAC_POINT(yZ, 0, 0, 1, 0) ME_FRAC_POINT(yZ, 0, 1, 1, 2)
  }
  // ---- integer columns
#ifndef ME_FRAC_CVT_FIRST_PASS
  if constexpr (!WP) {
    // the first pass of an integer column is (64 * p - (8192 << sh1)) >> sh1 = p * 2^(6 - sh1) - 8192: the sample itself goes into the
    // vertical filter (v_cvt_f32_ubyteN of the raw byte / a conversion of the 16-bit half), the taps carry the 2^(6 - sh1) and the
    // constant the -8192 -- the same real number, exact in fp32 like the general form.  The integer-position sample needs neither
    // filter nor clip: it is a sample of the window (with the window's bias, if any), inside [clip_lo, maxv] as it stands.  Weighted
    // calls take the general form below: their integer-position prediction is weighted like the others
    float V0[12][4];
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      if constexpr (BPS == 1) {
        const uint32_t raw = P[r][1] ^ 0x80808080u;   // patch columns 4..7 = block columns 0..3
#pragma unroll
        for (int c = 0; c < 4; ++c) V0[r][c] = (float)((raw >> (8 * c)) & 0xff);
      } else {
        V0[r][0] = (float)(P[r][2] & 0xffff); V0[r][1] = (float)(P[r][2] >> 16);
        V0[r][2] = (float)(P[r][3] & 0xffff); V0[r][3] = (float)(P[r][3] >> 16);
      }
    }
    const float up = (float)(64 >> sh1);
    float cw[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) cw[t] = cv[t] * up;
    const float init_i = init - 8192.f * 64.f * sc2;
    float yG[5][4], yZ[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        float a = init_i;
#pragma unroll
        for (int t = 0; t < 8; ++t) a = __builtin_fmaf(cw[t], V0[j + t][c], a);
        yG[j][c] = clipround(a);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) yZ[r][c] = V0[r + 4][c] + kRoundMagic;
    }
    ME_FRAC_POINT(yG, 0, 0, 0, 1) ME_FRAC_POINT(yG, 1, 0, 2, 1) ME_FRAC_POINT(yZ, 0, 0, 1, 1)
  } else
#endif
  {
    float V0[12][4];
#pragma unroll
    for (int r = 0; r < 12; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) V0[r][c] = first(r, 0, c);
    float yG[5][4], yZ[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        float a = init;
#pragma unroll
        for (int t = 0; t < 8; ++t) a = __builtin_fmaf(cv[t], V0[j + t][c], a);
        yG[j][c] = clipround(a);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) yZ[r][c] = clipround(__builtin_fmaf(c64, V0[r + 4][c], init));
    }
    ME_FRAC_POINT(yG, 0, 0, 0, 1) ME_FRAC_POINT(yG, 1, 0, 2, 1) ME_FRAC_POINT(yZ, 0, 0, 1, 1)
  }
#undef ME_FRAC_POINT
}

// ---- tree-shaped accumulation of a wave whose sixteen implicit items share ONE key -------------------------------------------------------
// The per-entry walk (me_frac_compute) adds an item's nine distortions to every slot of its position that shares its key, slot by slot:
// 64 x 18 + 256 x 6 = 2 688 lane adds per CTU and stage, and because the large slots cover many positions up to sixteen lanes of a
// wave add to the same LDS address in one instruction (0.63 of the kernel's LDS cycles were bank-conflict cycles on content whose slots
// share their motion, profiles/r05l_frac_counters_by_content.txt).  Where all sixteen 8x8 positions of a wave -- two position rows, a
// 64 x 16 strip of the CTU -- carry the same key K, what a slot with key K gets from the wave is the SUM of the values of the strip's
// positions inside it: sums over aligned groups of positions, formed once with six lane exchanges (A .. G below; a .. f for the slots made
// of 4x4 blocks), each added by ONE lane of its group -- 744 adds per CTU and stage, no two lanes of an instruction at one address.
// A slot whose key is not K takes nothing here: its positions are listed items of their own (me_frac_dedupe), as before.
// Lane bits inside the wave: 0 rx, 1 ry (quadrant), 2 x0, 3 x1, 4 x2 (position column), 5 y0 (position row).  The designated lanes and
// the slot numbers are tools/frac_tree_model.py's, which checks them against the slot table: every slot is tiled exactly once.
template <int BPS> struct FracSum;
template <> struct FracSum<1> { unsigned long long p[3]; };     // three 21-bit fields per word (frac_pack3)
template <> struct FracSum<2> { uint32_t d[9]; };
template <int BIT>
__device__ __forceinline__ uint32_t me_lane_xor_get(uint32_t v) {   // the value of lane (l ^ (1 << BIT)), BIT = 0..3
  if constexpr (BIT == 0) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
  else if constexpr (BIT == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  else if constexpr (BIT == 2) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);                     // bit mode: and 0x1f, or 0, xor 4
  else return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);                          // row_ror:8
}
template <int BIT>
__device__ __forceinline__ uint32_t me_lane_xor_sum(uint32_t v) {   // v + the value of lane (l ^ (1 << BIT)), BIT = 0..5
  if constexpr (BIT == 4) { const u32x2_t r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return r.x + r.y; }
  else if constexpr (BIT == 5) { const u32x2_t r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return r.x + r.y; }
  else return v + me_lane_xor_get<BIT>(v);
}
template <int BIT>
__device__ __forceinline__ unsigned long long me_lane_xor_sum(unsigned long long v) {
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  if constexpr (BIT == 4) {
    const u32x2_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return ((unsigned long long)a.x | (unsigned long long)b.x << 32) + ((unsigned long long)a.y | (unsigned long long)b.y << 32);
  } else if constexpr (BIT == 5) {
    const u32x2_t a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return ((unsigned long long)a.x | (unsigned long long)b.x << 32) + ((unsigned long long)a.y | (unsigned long long)b.y << 32);
  } else {
    return v + ((unsigned long long)me_lane_xor_get<BIT>(lo) | (unsigned long long)me_lane_xor_get<BIT>(hi) << 32);
  }
}
template <int BIT, int STAGE>
__device__ __forceinline__ FracSum<1> me_frac_xsum(const FracSum<1>& v) {
  FracSum<1> r;
#pragma unroll
  for (int k = 0; k < 3; ++k) r.p[k] = me_lane_xor_sum<BIT>(v.p[k]);
  return r;
}
template <int BIT, int STAGE>
__device__ __forceinline__ FracSum<2> me_frac_xsum(const FracSum<2>& v) {
  FracSum<2> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.d[i] = i < STAGE ? 0u : me_lane_xor_sum<BIT>(v.d[i]);   // stage 1: point 0 is carried over, nothing adds to it
  return r;
}
template <int STAGE, int BPS>
__device__ __forceinline__ void me_frac_tree_add(const uint32_t (&dist)[9], const uint32_t (&dist4)[9], bool want4, uint32_t key, const uint32_t* st,
                                                 uint32_t* acc, int tid) {
  constexpr uint32_t keymask = STAGE ? kFracKey1 : kFracKey0;
  // everything below depends on the lane's place alone: without this the compiler computes the designated-lane masks and the slot numbers
  // once per kernel and keeps them -- 28 SGPRs and 16 VGPRs -- across the item evaluation, which has none to spare (56 B of scratch per lane)
  asm volatile("" : "+v"(tid));
  const int role = tid & 3, rx = role & 1, ry = role >> 1, x = (tid >> 2) & 7, y = tid >> 5;
  const int w = y >> 1, x0 = x & 1, y0 = y & 1, r16 = (y >> 1) * 4 + (x >> 1), r32 = (y >> 2) * 2 + (x >> 2);
  auto amp_row = [](int r) { return (4 - r) & 3; };        // row (of a CU's four) -> 2NxnU.p0, 2NxnU.p1, 2NxnD.p0, 2NxnD.p1 = sub-families 0, 3, 2, 1
  auto amp_col = [](int c) { return 4 + ((4 - c) & 3); };  // column -> nLx2N.p0, nLx2N.p1, nRx2N.p0, nRx2N.p1 = 4, 7, 6, 5
  auto pack = [](const uint32_t (&d)[9]) {
    FracSum<BPS> v;
    if constexpr (BPS == 1) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
        v.p[k] = (unsigned long long)(d[3 * k] | d[3 * k + 1] << 21) | (unsigned long long)(d[3 * k + 1] >> 11 | d[3 * k + 2] << 10) << 32;
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i) v.d[i] = d[i];
    }
    return v;
  };
  auto put = [&](bool mine, int slot, const FracSum<BPS>& v) {   // the designated lane adds the group's sum if the slot shares the wave's key
    if (mine && ((st[slot] ^ key) & keymask) == 0) {
      if constexpr (BPS == 1) {
        unsigned long long* a = (unsigned long long*)acc + slot * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) atomicAdd(&a[k], v.p[k]);
      } else {
#pragma unroll
        for (int i = STAGE; i < 9; ++i) atomicAdd(&acc[slot * 9 + i], v.d[i]);
      }
    }
  };
  // ---- slots made of 8x8 blocks: the quad's value (the same in its four lanes)
  const FracSum<BPS> A = pack(dist);
  put(role == 0, 384 + y * 8 + x, A);                                                            //  8: 2Nx2N
  const FracSum<BPS> C = me_frac_xsum<5, STAGE>(A);                                              // 8 x 16
  put(y0 == 0 && role >= 2, role == 2 ? 480 + (y >> 1) * 8 + x                                   // 16: Nx2N part x & 1
                                      : 512 + amp_col(x & 3) * 4 + r32, C);                      // 32: the AMP part this column belongs to (alone or as a piece)
  const FracSum<BPS> B = me_frac_xsum<2, STAGE>(A);                                              // 16 x 8
  put(x0 == 0 && role == 1, 448 + (y >> 1) * 8 + y0 * 4 + (x >> 1), B);                          // 16: 2NxN part y & 1
  const FracSum<BPS> E = me_frac_xsum<3, STAGE>(B);                                              // 32 x 8
  put((x & 3) == 1 && role == 1, 512 + amp_row(y & 3) * 4 + r32, E);                             // 32: the AMP part this position row belongs to
  const FracSum<BPS> D = me_frac_xsum<5, STAGE>(B);                                              // 16 x 16
  {
    const int cx = x >> 1;
    const int slot = role == 0 ? 544 + r16                                                       // 16: 2Nx2N
                   : role == 1 ? 568 + (y >> 2) * 4 + 2 * (x >> 2) + (cx & 1)                    // 32: Nx2N part
                   : role == 2 ? 512 + ((cx & 1) ? 7 : 6) * 4 + r32                              // 32: the two-column piece of nLx2N.p1 / nRx2N.p0
                               : 576 + amp_col(cx);                                              // 64: the AMP part this 16-column belongs to
    put(x0 == 1 && y0 == 1, slot, D);
  }
  const FracSum<BPS> F = me_frac_xsum<3, STAGE>(D);                                              // 32 x 16
  {
    const int slot = role == 0 ? 584 + r32                                                       // 32: 2Nx2N
                   : role == 1 ? 560 + (y >> 2) * 4 + ((y >> 1) & 1) * 2 + (x >> 2)              // 32: 2NxN part
                   : role == 2 ? 512 + ((w & 1) ? 3 : 2) * 4 + r32                               // 32: the two-row piece of 2NxnU.p1 / 2NxnD.p0
                               : 590 + (x >> 2);                                                 // 64: Nx2N part
    put((x & 3) == 2 && y0 == 0, slot, F);
    put((x & 3) == 3 && y0 == 0 && role == 0, 576 + ((x >> 2) ? 7 : 6), F);                     // 64: the 32-column piece of nLx2N.p1 / nRx2N.p0
  }
  const FracSum<BPS> G = me_frac_xsum<4, STAGE>(F);                                              // 64 x 16: the wave's strip
  {
    const int slot = role == 0 ? 592 : role == 1 ? 588 + (w >> 1) : role == 2 ? 576 + (w == 0 ? 0 : 3) : 576 + (w == 3 ? 1 : 2);
    put(x == 7 && y0 == 1, slot, G);
  }
  // ---- slots made of 4x4 blocks: the lane's own 4x4 value
  if (want4) {   // wave-uniform
    const FracSum<BPS> u = pack(dist4);
    const FracSum<BPS> a = me_frac_xsum<0, STAGE>(u), b = me_frac_xsum<1, STAGE>(u);             // 8 x 4, 4 x 8
    {
      FracSum<BPS> ab;   // element by element: a select between the two structs goes through the stack (56 B of scratch per lane)
      if constexpr (BPS == 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) ab.p[k] = rx == ry ? a.p[k] : b.p[k];
      } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) ab.d[i] = rx == ry ? a.d[i] : b.d[i];
      }
      put(true, rx == ry ? y * 16 + ry * 8 + x : 128 + y * 16 + 2 * x + rx, ab);
    }
    const int r = 2 * y0 + ry, c = 2 * x0 + rx;                                                  // 4x4 row / column inside the 16x16 CU
    const FracSum<BPS> cs = me_frac_xsum<2, STAGE>(a);                                           // 16 x 4
    put(rx == 0 && x0 == 1, 256 + amp_row(r) * 16 + r16, cs);
    const FracSum<BPS> ds = me_frac_xsum<5, STAGE>(b);                                           // 4 x 16
    put(ry == 1 && y0 == 1, 256 + amp_col(c) * 16 + r16, ds);
    const FracSum<BPS> es = me_frac_xsum<1, STAGE>(cs);                                          // 16 x 8: rows 2,3 of 2NxnU.p1 / rows 0,1 of 2NxnD.p0
    put(rx == 1 && ry == 1 && x0 == 0, 256 + (y0 ? 3 : 2) * 16 + r16, es);
    const FracSum<BPS> fs = me_frac_xsum<0, STAGE>(ds);                                          // 8 x 16: columns 2,3 of nLx2N.p1 / columns 0,1 of nRx2N.p0
    put(rx == 1 && ry == 0 && y0 == 0, 256 + (x0 ? 7 : 6) * 16 + r16, fs);
  }
}

// One work item = one 4x4 block of one (position, MV) pair; `pair` indexes the cover table (kind-8 pairs first), `role` is the lane's
// quadrant of an 8x8 Hadamard block.  An item is handled in two steps:
//   me_frac_fetch    slot state + address -> 12 raw rows of PW + 1 aligned dwords each (global_load_dwordx4 / x3), nothing waited for
//   me_frac_compute  byte-align the rows (first use = the wait), evaluate the 9 / 8 points, add to the slots that share the key
template <int BPS>
struct FracRaw {
  uint32_t w[12][3 * BPS + 1];
  uint32_t sv;      // state word of the item's first slot (the sharing key of the stage)
  uint32_t o;       // byte offset of the patch inside its first dword
};

// LDS-DMA form of me_frac_fetch (8-bit planes): the rows of the item go from the plane straight into the wave's LDS block
// (global_load_lds_dwordx4: lane l's 16 bytes of row r land at block + r * 1 KiB + l * 16), no register holds them while they travel.
// Returns what me_frac_take needs beside the rows: the item's state word and the byte offset of the patch inside its first dword.
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
template <int STAGE, int KIND8>
__device__ __forceinline__ uint32_t me_frac_prefetch(const uint8_t* __restrict__ src, int gpitch, const uint32_t* st, const uint16_t* cover, int pair, int role,
                                                     uint32_t* pf_wave) {
  constexpr int NCOV = KIND8 ? kFracCover8 : kFracCover4;
  const int q = KIND8 ? pair : pair - kFracPairs8;
  const int pos = q / NCOV;
  const uint32_t sv = st[cover[pair]];
  const int bx = KIND8 ? 2 * (pos & 7) + (role & 1) : (pos & 15), by = KIND8 ? 2 * (pos >> 3) + (role >> 1) : (pos >> 4);
  const int skip_x = STAGE ? 4 + me_win8_first((int)((sv >> 18) & 3)) : 0, skip_y = STAGE ? 4 + me_win8_first((int)((sv >> 20) & 3)) : 0;
  const int prow = by * 4 + (int)((sv >> 9) & 0x1ff) + skip_y, pcol = bx * 4 + (int)(sv & 0x1ff) + skip_x;
  const uint8_t* a = src + (long)prow * gpitch + pcol;
  const uint32_t o = (uint32_t)(uintptr_t)a & 3u;
  const uint8_t* rowp = a - o;
  constexpr int ROWS = STAGE ? kFracRows1 : 12;
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
    __builtin_amdgcn_global_load_lds((const void*)(rowp + (long)r * gpitch), (lds_u32_t*)(pf_wave + r * 256), 16, 0, 0);
  return (sv & kFracKey1) | o << 30;
}
// the rows of the item me_frac_prefetch asked for, out of the wave's LDS block into registers; returns once they are THERE, so the block
// may be overwritten by the next request
template <int STAGE>
__device__ __forceinline__ void me_frac_take(const uint32_t* pf_wave, int lane, uint32_t meta, FracRaw<1>& R) {
  constexpr int ROWS = STAGE ? kFracRows1 : 12;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA completion is counted with the vector-memory operations
#pragma unroll
  for (int r = 0; r < 12; ++r) {
    if (r < ROWS) {
      const uint4 v = *(const uint4*)(pf_wave + r * 256 + lane * 4);
      R.w[r][0] = v.x; R.w[r][1] = v.y; R.w[r][2] = v.z; R.w[r][3] = v.w;
    } else {
      R.w[r][0] = R.w[r][1] = R.w[r][2] = R.w[r][3] = 0u;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  R.sv = meta & kFracKey1;
  R.o = meta >> 30;
}

template <int STAGE, int BPS, int KIND8>
__device__ __forceinline__ void me_frac_fetch(const uint8_t* __restrict__ src, int gpitch, const uint32_t* st, const uint16_t* cover, int pair, int role,
                                              FracRaw<BPS>& R) {
  constexpr int PW = 3 * BPS, NCOV = KIND8 ? kFracCover8 : kFracCover4;
  const int q = KIND8 ? pair : pair - kFracPairs8;
  const int pos = q / NCOV;
  const uint32_t sv = st[cover[pair]];
  const int bx = KIND8 ? 2 * (pos & 7) + (role & 1) : (pos & 15), by = KIND8 ? 2 * (pos >> 3) + (role >> 1) : (pos >> 4);
  // patch (0,0) = block sample (-4,-4) = window row (by*4 + my - lt_y), sample (bx*4 + mx - lt_x) (halo offsets cancel); in the
  // quarter-pel stage the patch starts at the first sample of the component's shared 8-tap window (me_win8_first): (-4 or -3, -4 or -3)
  const int skip_x = STAGE ? 4 + me_win8_first((int)((sv >> 18) & 3)) : 0, skip_y = STAGE ? 4 + me_win8_first((int)((sv >> 20) & 3)) : 0;
  const int prow = by * 4 + (int)((sv >> 9) & 0x1ff) + skip_y, pcol = (bx * 4 + (int)(sv & 0x1ff) + skip_x) * BPS;   // pcol in bytes
  const uint8_t* a = src + (long)prow * gpitch + pcol;
  const uint32_t o = (uint32_t)(uintptr_t)a & 3u;
  const uint32_t* __restrict__ rowp = (const uint32_t*)(a - o);
  const int gp = gpitch >> 2;
  // 11 u16 samples and their alignment fill six dwords at most; the seventh only feeds sample 11, which no tap reads
  constexpr int ROWS = STAGE ? kFracRows1 : 12, NL = (STAGE && BPS == 2) ? PW : PW + 1;
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int k = 0; k <= PW; ++k) R.w[r][k] = k < NL ? rowp[r * gp + k] : 0u;
  R.sv = sv;
  R.o = o;
}

template <int STAGE, int HAD, int BPS, int KIND8, int WP>
__device__ __forceinline__ void me_frac_compute(const FracRaw<BPS>& R, const uint32_t* curl, const uint32_t* st, const uint16_t* cover, int pair, int role,
                                                int bd, float clip_lo, const FracWp wp, const uint32_t* tab_h, const float* tab_v, uint32_t* acc, bool ride) {
  constexpr int PW = 3 * BPS, NCOV = KIND8 ? kFracCover8 : kFracCover4;
  constexpr uint32_t keymask = STAGE ? kFracKey1 : kFracKey0;
  const int q = KIND8 ? pair : pair - kFracPairs8;
  const int pos = q / NCOV, j = q - pos * NCOV;
  const uint16_t* cov = cover + (KIND8 ? 0 : kFracPairs8) + pos * NCOV;
  const uint32_t sv = R.sv;
  const int bx = KIND8 ? 2 * (pos & 7) + (role & 1) : (pos & 15), by = KIND8 ? 2 * (pos >> 3) + (role >> 1) : (pos >> 4);
  constexpr int ROWS = STAGE ? kFracRows1 : 12;
  uint32_t P[ROWS][PW];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int k = 0; k < PW; ++k) P[r][k] = __builtin_amdgcn_alignbyte(R.w[r][k + 1], R.w[r][k], R.o) ^ (BPS == 1 ? 0x80808080u : 0u);
  float orgM[16];   // current samples + kRoundMagic (exact: integers below 2^24); WP: current samples - staging bias - offset, no magic
  const float org_add = WP ? -wp.org_sub : kRoundMagic;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if constexpr (BPS == 1) {
      const uint32_t w = curl[(by * 4 + r) * 16 + bx];
#pragma unroll
      for (int c = 0; c < 4; ++c) orgM[4 * r + c] = (float)((w >> (8 * c)) & 0xff) + org_add;
    } else {
      const uint2 w = *(const uint2*)&curl[(by * 4 + r) * 32 + bx * 2];
      orgM[4 * r] = (float)(w.x & 0xffff) + org_add; orgM[4 * r + 1] = (float)(w.x >> 16) + org_add;
      orgM[4 * r + 2] = (float)(w.y & 0xffff) + org_add; orgM[4 * r + 3] = (float)(w.y >> 16) + org_add;
    }
  }
  uint32_t dist[9], dist4[9];
  // Slots made of 4x4 blocks (the AMP shapes at 16, 8x4, 4x8: six per 4x4 position) whose key equals this item's get this lane's 4x4
  // block for nothing: same patch, same interpolated samples, same difference, same 4x4 transform -- the work-list pass left such
  // (4x4 position, key) pairs out of the 4x4 list (me_frac_dedupe4).  On coherent content that is every one of them.
  // `ride` (wave-uniform): only a position's IMPLICIT item carries riders.  A listed kind-8 item used to as well; but one rider in a
  // wave makes all 64 lanes compute their own 4x4 sums at all nine points (7 % of an item), and on content whose slots do not share
  // their motion nearly every wave had one and nearly no lane needed it -- as an item of its own that rider costs one LANE of a wave.
  uint32_t match4 = 0;
  const uint16_t* cov4 = cover + kFracPairs8 + (by * 16 + bx) * kFracCover4;
  if (KIND8 && ride) {
#pragma unroll
    for (int k = 0; k < kFracCover4; ++k) match4 |= (((st[cov4[k]] ^ sv) & keymask) == 0 ? 1u : 0u) << k;
  }
  const bool want4 = KIND8 && __any(match4 != 0);
  {
    const FracOrgRegs org = {orgM};
    if constexpr (STAGE == 0) me_frac_eval0<HAD, BPS, KIND8, WP>(P, org, role, bd, clip_lo, want4, wp, dist, dist4);
    else me_frac_eval1<HAD, BPS, KIND8, WP>(P, org, (int)((sv >> 18) & 3), (int)((sv >> 20) & 3), role, bd, clip_lo, tab_h, tab_v, want4, wp, dist, dist4);
  }
#ifndef ME_FRAC_T_NOATOMICS   // timing-only builds (tools/r04_frac_breakdown.sh; results are wrong by design): ME_FRAC_T_NOATOMICS, ME_FRAC_T_NOITEMS
#if ME_FRAC_TREE && !ME_FRAC_RIDE_LISTED   // (with riders on listed items `ride` no longer means "the implicit item")
  if constexpr (KIND8) {
    // the implicit items (lane tid = position tid >> 2, quadrant tid & 3) of a wave whose sixteen positions share one key: sums over
    // groups of positions, one add per (slot, group)
    static_assert(BPS == 2 || frac_pack3(1), "me_frac_tree_add adds the packed sums of 8-bit planes");
    if (ride && __all(((sv ^ (uint32_t)__builtin_amdgcn_readfirstlane((int)sv)) & keymask) == 0)) {
      me_frac_tree_add<STAGE, BPS>(dist, dist4, want4, sv, st, acc, pos * 4 + role);
      return;
    }
  }
#endif
  // the nine (stage 1: eight, point 0 is carried over, not evaluated -- its distortion here is 0) distortions as they are added to a slot
  unsigned long long pk[3], pk4[3];
  if constexpr (frac_pack3(BPS)) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      pk[k] = (unsigned long long)(dist[3 * k] | dist[3 * k + 1] << 21) | (unsigned long long)(dist[3 * k + 1] >> 11 | dist[3 * k + 2] << 10) << 32;
      pk4[k] = (unsigned long long)(dist4[3 * k] | dist4[3 * k + 1] << 21) | (unsigned long long)(dist4[3 * k + 1] >> 11 | dist4[3 * k + 2] << 10) << 32;
    }
  }
  auto add_to = [&](int slot, const uint32_t (&d)[9], const unsigned long long (&p)[3]) {
    if constexpr (frac_pack3(BPS)) {
      unsigned long long* a = (unsigned long long*)acc + slot * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) atomicAdd(&a[k], p[k]);
    } else {
#pragma unroll
      for (int i = STAGE; i < 9; ++i) atomicAdd(&acc[slot * 9 + i], d[i]);
    }
  };
  if (KIND8 && want4) {
#pragma unroll
    for (int k = 0; k < kFracCover4; ++k)
      if (match4 >> k & 1) add_to(cov4[k], dist4, pk4);
  }
  // Every slot of this position with the same key takes the distortions; the quad's lanes split the slot list.
  // (Round 3 tried to relieve these adds -- the large slots cover many positions, so lanes of one wave often add to the same address:
  // each quad starting its walk at another list position and two points per ds_add_u64 cut the kernel's LDS conflict cycles by 27 %
  // and its LDS waits by 90 %, and changed its time by -2 % .. +3 %: LDS is busy 10 % of the CU cycles here.  profiles/archive/r03d_frac_ab.txt)
  // (slot ids first, then their keys, then the adds: two LDS round trips for the lane's whole share -- walking the list entry by entry
  // was two dependent round trips PER entry at the end of every item, with nothing else for the wave to issue)
  {
    constexpr int STEP = KIND8 ? 4 : 1, MAXN = KIND8 ? 5 : kFracCover4;
    const int j0 = j + (KIND8 ? role : 0);
    int s2[MAXN];
    uint32_t k2[MAXN];
#pragma unroll
    for (int k = 0; k < MAXN; ++k) s2[k] = cov[min(j0 + k * STEP, NCOV - 1)];
#pragma unroll
    for (int k = 0; k < MAXN; ++k) k2[k] = st[s2[k]];
#pragma unroll
    for (int k = 0; k < MAXN; ++k) {
      const int j2 = j0 + k * STEP;
      if (j2 < NCOV && (j2 == j || ((k2[k] ^ sv) & keymask) == 0)) add_to(s2[k], dist, pk);
    }
  }
#else   // every distortion still computed, ONE atomic per item: the difference to the product build is what the accumulation costs
  {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) t += dist[i] + (KIND8 ? dist4[i] & match4 : 0u);
    atomicAdd(&acc[(cov[j] * frac_acc_row(BPS) + (t & 3)) % (593 * frac_acc_row(BPS))], t);
  }
#endif
}

// distinct (position, key) pairs of one stage -> work lists; the first slot (lowest index in the cover list) with a given key is the one
// evaluated.  Entry 0 of a position's list -- its 8x8 slot -- is always a first: it is the position's IMPLICIT item (me_frac_stage) and never
// listed.  PARTS waves share a position's list: part PART decides entries PART, PART + PARTS, ... (an 8x8 position's 18 entries are
// 153 compares: one wave per part -- the part is wave-uniform, so each wave runs only its own 32..45 compares -- keeps the four
// waves of the workgroup equally busy).  A wave whose 64 positions each have ONE key (slots that share their motion: most of a real
// picture) has nothing to list and skips the compares.
template <int NCOV, int PARTS, int PART>
__device__ __forceinline__ void me_frac_dedupe(const uint32_t* st, const uint16_t* cov, uint32_t keymask, int pair0,
                                               uint32_t* counter, uint16_t* list) {
  uint32_t key[NCOV];
#pragma unroll
  for (int j = 0; j < NCOV; ++j) key[j] = st[cov[j]] & keymask;
  uint32_t differ = 0;
#pragma unroll
  for (int j = 1; j < NCOV; ++j) differ |= key[j] ^ key[0];
  if (__all(differ == 0)) return;
#pragma unroll
  for (int j = PART ? PART : PARTS; j < NCOV; j += PARTS) {
    bool first = true;
#pragma unroll
    for (int k = 0; k < j; ++k) first = first && key[k] != key[j];
    if (first) list[atomicAdd(counter, 1u)] = (uint16_t)(pair0 + j);
  }
}

// the 4x4 kind: a (4x4 position, key) pair is listed unless its key is that of the position's 8x8 slot -- then the lane of the position's
// implicit item that owns this 4x4 block hands its result to the 4x4-kind slots as well (me_frac_compute, `ride`)
__device__ __forceinline__ void me_frac_dedupe4(const uint32_t* st, const uint16_t* cover, int p4, uint32_t keymask, uint32_t* counter, uint16_t* list) {
  const uint16_t* cov = cover + kFracPairs8 + p4 * kFracCover4;
  const uint16_t* cov8 = cover + (((p4 >> 4) >> 1) * 8 + ((p4 & 15) >> 1)) * kFracCover8;   // the 8x8 position that holds block (p4 & 15, p4 >> 4)
  uint32_t key[kFracCover4];
  bool first[kFracCover4];
  const uint32_t k80 = st[cov8[0]] & keymask;
  uint32_t differ = 0;
#pragma unroll
  for (int j = 0; j < kFracCover4; ++j) {
    key[j] = st[cov[j]] & keymask;
    differ |= key[j] ^ k80;
  }
  if (__all(differ == 0)) return;   // every 4x4-kind slot of the wave's positions rides on its position's implicit 8x8 item
#pragma unroll
  for (int j = 0; j < kFracCover4; ++j) {
    first[j] = key[j] != k80;
#pragma unroll
    for (int k = 0; k < j; ++k) first[j] = first[j] && key[k] != key[j];
  }
#ifdef ME_FRAC_RIDE_ANY   // A/B: riders on listed kind-8 items too (round 4 / early round 5)
#pragma unroll
  for (int k = 1; k < kFracCover8; ++k) {
    const uint32_t k8 = st[cov8[k]] & keymask;
#pragma unroll
    for (int j = 0; j < kFracCover4; ++j) first[j] = first[j] && key[j] != k8;
  }
#endif
#pragma unroll
  for (int j = 0; j < kFracCover4; ++j)
    if (first[j]) list[atomicAdd(counter, 1u)] = (uint16_t)(kFracPairs8 + p4 * kFracCover4 + j);
}

// One stage's items.  Every 8x8 position has an item that needs no list: entry 0 of its cover list (the 8x8 slot itself) with that slot's
// key -- 64 positions x 4 quadrants = one item for each of the 256 lanes.  A lane requests that item's patch rows FIRST, builds its share
// of the work lists (the remaining distinct (position, key) pairs) while the rows are on their way, evaluates the item, and only then meets
// the other waves at the barrier that completes the lists.  On content whose slots share their motion the lists stay empty and a stage is
// one item per lane; the list phase used to be a barrier-separated 2.5 .. 3.5 us of dependent LDS round trips in front of every stage's
// items (profiles/r05k_frac_phases_before_after.txt).  The listed items follow: kind-8 items (whole quads), then kind-4 items.  (Round 4 built the walk
// over the listed items software-pipelined -- the next item's rows requested before the current item is evaluated -- slower on every content,
// profiles/r04a_frac_pipelined_vs_plain_ab.txt.)
template <int STAGE, int HAD, int BPS, int WP>
__device__ __forceinline__ void me_frac_stage(const uint8_t* __restrict__ src, int gpitch, const uint32_t* curl, const uint32_t* st, const uint16_t* cover,
                                              uint16_t* list8, uint16_t* list4, uint32_t* counter, uint32_t* pf, int tid, int bd, float clip_lo,
                                              const FracWp wp, const uint32_t* tab_h, const float* tab_v, uint32_t* acc) {
  constexpr int NT = frac_threads(BPS);
  static_assert(NT == 256, "me_frac_stage: one implicit item per lane (64 positions x 4 quadrants), the work-list pass on 4 waves");
  constexpr uint32_t keymask = STAGE ? kFracKey1 : kFracKey0;
  const int role = tid & 3;
  // the work lists: 64 8x8 positions x 4 waves (wave w decides entries w, w + 4, ... of each position's list), then 256 4x4 positions
  auto lists = [&]() {
    const int p8 = tid & 63;
    const uint16_t* c8 = cover + p8 * kFracCover8;
    switch (__builtin_amdgcn_readfirstlane(tid >> 6)) {
      case 0: me_frac_dedupe<kFracCover8, 4, 0>(st, c8, keymask, p8 * kFracCover8, &counter[0], list8); break;
      case 1: me_frac_dedupe<kFracCover8, 4, 1>(st, c8, keymask, p8 * kFracCover8, &counter[0], list8); break;
      case 2: me_frac_dedupe<kFracCover8, 4, 2>(st, c8, keymask, p8 * kFracCover8, &counter[0], list8); break;
      default: me_frac_dedupe<kFracCover8, 4, 3>(st, c8, keymask, p8 * kFracCover8, &counter[0], list8); break;
    }
    me_frac_dedupe4(st, cover, tid, keymask, &counter[1], list4);
  };
  if constexpr (frac_glds(BPS)) {
    // 8-bit planes: a lane's NEXT item is requested (LDS-DMA into the wave's block of `pf`) before the current one is evaluated -- the
    // registers that would have to hold rows in flight do not exist here (round 4's register-pipelined walk was slower on every
    // content), an LDS block nobody else wants does: the kernel's registers allow two workgroups a CU, which leaves each 80 KiB.
    uint32_t* pf_wave = pf + (tid >> 6) * (12 * 256);
    const int lane = tid & 63;
    int pair = (tid >> 2) * kFracCover8;
    uint32_t meta = me_frac_prefetch<STAGE, 1>(src, gpitch, st, cover, pair, role, pf_wave);
    lists();
    __syncthreads();   // the work lists are complete (the rows asked for have landed as well: the barrier's fence waits for them)
#if ME_FRAC_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    const int n8 = 4 * (int)counter[0], n4 = (int)counter[1];
    int i8 = tid - NT;
#pragma unroll 1
    for (;;) {
      FracRaw<BPS> R;
      me_frac_take<STAGE>(pf_wave, lane, meta, R);
      const int cur = pair;
      i8 += NT;
      const bool more = i8 < n8;
      if (more) {
        pair = list8[i8 >> 2];
        meta = me_frac_prefetch<STAGE, 1>(src, gpitch, st, cover, pair, role, pf_wave);
      }
#ifndef ME_FRAC_T_NOITEMS
      me_frac_compute<STAGE, HAD, BPS, 1, WP>(R, curl, st, cover, cur, role, bd, clip_lo, wp, tab_h, tab_v, acc, ME_FRAC_RIDE_LISTED || i8 < NT);
#endif
      if (!more) break;
    }
    int i4 = tid;
    if (i4 < n4) {
      pair = list4[i4];
      meta = me_frac_prefetch<STAGE, 0>(src, gpitch, st, cover, pair, 0, pf_wave);
#pragma unroll 1
      for (;;) {
        FracRaw<BPS> R;
        me_frac_take<STAGE>(pf_wave, lane, meta, R);
        const int cur = pair;
        i4 += NT;
        const bool more = i4 < n4;
        if (more) {
          pair = list4[i4];
          meta = me_frac_prefetch<STAGE, 0>(src, gpitch, st, cover, pair, 0, pf_wave);
        }
#ifndef ME_FRAC_T_NOITEMS
        me_frac_compute<STAGE, HAD, BPS, 0, WP>(R, curl, st, cover, cur, 0, bd, clip_lo, wp, tab_h, tab_v, acc, false);
#endif
        if (!more) break;
      }
    }
  } else {
    // (the listed items in chunks of 64 lanes drawn from a counter by the four waves were measured too: no better than this static walk
    // on mixed content, 1 % worse on unrelated pictures, and 14-46 spilled dwords in the u16 kernels)
    int n8 = 0;
#pragma unroll 1
    for (int i8 = tid - NT;; i8 += NT) {   // first turn (i8 < 0 in every lane): the implicit item
      int pair = (tid >> 2) * kFracCover8;
      if (i8 >= 0) {
        if (i8 >= n8) break;
        pair = list8[i8 >> 2];
      }
      FracRaw<BPS> R;
      me_frac_fetch<STAGE, BPS, 1>(src, gpitch, st, cover, pair, role, R);
      if (i8 < 0) {   // the work lists, while the implicit item's rows are on their way
        lists();
#if ME_FRAC_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
      }
#ifndef ME_FRAC_T_NOITEMS
      me_frac_compute<STAGE, HAD, BPS, 1, WP>(R, curl, st, cover, pair, role, bd, clip_lo, wp, tab_h, tab_v, acc, ME_FRAC_RIDE_LISTED || i8 < 0);
#endif
      if (i8 < 0) {
        __syncthreads();   // the work lists are complete
        n8 = 4 * (int)counter[0];
      }
    }
    const int n4 = (int)counter[1];
#ifndef ME_FRAC_T_NOITEMS
    // (dealt from the last thread down: the partly filled last turn of the kind-8 walk above falls on the first waves, this one's on
    // the last -- the waves reach the barrier that ends the stage closer together)
#pragma unroll 1
    for (int i4 = NT - 1 - tid; i4 < n4; i4 += NT) {
      const int pair = list4[i4];
      FracRaw<BPS> R;
      me_frac_fetch<STAGE, BPS, 0>(src, gpitch, st, cover, pair, 0, R);
      me_frac_compute<STAGE, HAD, BPS, 0, WP>(R, curl, st, cover, pair, 0, bd, clip_lo, wp, tab_h, tab_v, acc, false);
    }
#endif
  }
#if ME_FRAC_PRIO
  __builtin_amdgcn_s_setprio(3);
#endif
}

// Which job the k-th workgroup (or the k-th draw from the job counter) takes.  A launch ends one job time after its last job STARTS, and
// the CTUs on the picture's edge are the slow ones: their slots reach into the padding (a partial bottom row most of all: 2160 = 33 x 64
// + 48), find MVs of their own there and share little -- 2-3 x the time of an interior CTU (profiles/r04k_frac_timeline.txt: the bottom
// row dealt last kept a 2160p launch alive for an extra job time; profiles/r05k_frac_phases_before_after.txt: so did the TOP row once the table
// was simply dealt from its end).  So the edge CTUs of every pair go first -- bottom row, top row, left and right column -- then the
// interiors.  Launches over a CTU sub-range (and pictures less than three CTUs wide or high) keep the plain last-first order.
__host__ __device__ inline int me_frac_deal(int k, int n_jobs, const FracPrep& prep) {
  const int first = (int)(prep.ctus & 0xffff), count = (int)(prep.ctus >> 16);
  const int X = ((int)(prep.dims & 0xffff) + 63) >> 6, Y = ((int)(prep.dims >> 16) + 63) >> 6, n_ctu = X * Y;
  if (first != 0 || count != n_ctu || X < 3 || Y < 3 || n_jobs % n_ctu) return n_jobs - 1 - k;
  const int E = 2 * X + 2 * (Y - 2), I = n_ctu - E, pairs = n_jobs / n_ctu;
  int pair, ctu;
  if (k < pairs * E) {
    pair = k / E;
    int e = k - pair * E;
    if (e < X) ctu = (Y - 1) * X + e;
    else if (e < 2 * X) ctu = e - X;
    else { e -= 2 * X; ctu = (1 + (e >> 1)) * X + ((e & 1) ? X - 1 : 0); }
  } else {
    const int m = k - pairs * E;
    pair = m / I;
    const int i = m - pair * I;
    ctu = (1 + i / (X - 2)) * X + 1 + i % (X - 2);
  }
  return pair * n_ctu + ctu;
}

// WAVES: waves per SIMD the register budget is cut for.  The kernel runs at two (8-bit planes: 237 VGPRs, nothing spilled; two workgroups
// per CU).  Round 4 also ran the 8-bit kernel at three waves for large launches (168 VGPRs + 50 spilled dwords per lane: 104 MB of
// scratch writes per 2160p launch); the round-5 flow made the two-wave build the faster one on content that shares its motion and
// the only one without scratch -- on unrelated pictures the third wave's occupancy would still be worth 6 %.
template <int HAD, int BPS, int WP, int WAVES = 2>
__global__ void __launch_bounds__(frac_threads(BPS), WAVES)
me_frac_kernel(const RefSet curs, int cur_pitch, const RefSet refs, int ref_pitch,
               const MeJob* __restrict__ jobs, const FracPrep prep, int n_jobs, uint32_t* __restrict__ job_counter, const uint16_t* __restrict__ cover_g,
               const int16_t* __restrict__ int_mv, uint32_t lambda_q16, int bit_depth_bias, const FracWp wp, int16_t* __restrict__ out_qmv,
               uint32_t* __restrict__ out_cost) {
  // bit_depth_bias: bit depth in the low 8 bits; bit 8 set = block and window carry the bias 2^bitDepth of a bi-prediction origin
  // (2*org - pred, TEncSearch.cpp:3702-3712: current samples in [-maxv, 2*maxv]; per-CTU calls only, u16 staging)
  const int bit_depth = bit_depth_bias & 0xff;
  static_assert(BPS == 2 || !WP, "weighted calls and bi-prediction origins stage u16 samples (hmme.hip ctu_call)");
  const float clip_lo = (BPS == 2 && (bit_depth_bias & 0x100)) ? (float)(1 << bit_depth) : 0.f;
  constexpr int NT = frac_threads(BPS);
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t* acc = smem;                 // [593][frac_acc_row(BPS)] distortion sums of the current stage (8-bit planes: three packed 64-bit words per slot)
  uint32_t* st = smem + frac_acc_dw(BPS);   // [593] slot state: (mx - lt_x) | (my - lt_y) << 9 | (half_x + 1) << 18 | (half_y + 1) << 20
  uint32_t* tab_h = st + 600;           // [9][kFracTabH] packed horizontal taps of the quarter-pel stage: row = (half-pel winner + 1) * 3 + (offset + 1), me_tap8
  float* tab_v = (float*)(tab_h + 9 * kFracTabH);   // [9][kFracTabV] vertical taps, same rows
  uint32_t* counter = tab_h + 152;      // [2] lengths of the two work lists
  uint16_t* list8 = (uint16_t*)(tab_h + 160);                  // distinct (8x8 position, key) pairs
  uint16_t* list4 = (uint16_t*)(tab_h + 160 + kFracPairs8 / 2);   // distinct (4x4 position, key) pairs
  uint32_t* curl = tab_h + 160 + (kFracPairs8 + kFracPairs4) / 2;   // 64 x 64 current block
  uint16_t* cover = (uint16_t*)(curl + 1024 * BPS);   // the cover table (uint16 [64][18] then [256][6]): read per item and per dedupe, so it lives here
  uint32_t* pf = (uint32_t*)(cover + kFracPairs8 + kFracPairs4);   // 8-bit planes: [4 waves][12 rows][64 lanes][16 B] patch rows in flight (me_frac_stage)

  const int tid = threadIdx.x;
  const int bd = BPS == 1 ? 8 : bit_depth;
#if ME_FRAC_PRIO
  // The short phases of a job -- set-up, work lists, winners: chains of dependent LDS round trips and compares, a few hundred
  // instructions -- run at wave priority 3, the items (thousands of VALU instructions per lane, no waiting) at 0.  A SIMD issues from its
  // oldest ready wave: next to another workgroup's items a short phase got the issue slots those left over and took five times its
  // stand-alone time, half of a job on content whose slots share their motion.  With the priorities the short phases run at their own
  // latency and the items fill every slot they leave, which is nearly all of them.
  __builtin_amdgcn_s_setprio(3);
#endif
  // tables that do not depend on the job
  static_assert(9 * (kFracTabH + kFracTabV) <= 152, "me_frac_kernel: the tap tables end where the list counters start");
  if (tid < 9 * kFracTabH) tab_h[tid] = (tid & 7) < 2 * BPS ? me_htap8_dw<BPS>((tid >> 3) / 3, (tid >> 3) % 3, tid & 7) : 0u;
  if (tid >= 128 && tid < 128 + 9 * kFracTabV) {
    const int i = tid - 128, row = i / kFracTabV, j = i - row * kFracTabV;
    tab_v[i] = (float)me_tap8(row / 3, row % 3, j) * (BPS == 1 ? 1.f / 4096.f : __int_as_float((127 - (20 - bd)) << 23));
  }
  for (int i = tid; i < (kFracPairs8 + kFracPairs4) / 8; i += NT) ((uint4*)cover)[i] = ((const uint4*)cover_g)[i];

  // One workgroup per job is the default launch (job_counter == null: workgroup b takes job n_jobs - 1 - b).  With a counter
  // (HMME_FRAC_GRID: fewer workgroups than jobs, each taking job after job as it becomes free) the loop below runs more than once;
  // measured, that launch is the slower one once both deal the jobs last-first.
  uint32_t* next_job = counter + 2;     // LDS word: the job this workgroup works on
#pragma unroll 1
  for (int turn = blockIdx.x;; turn += gridDim.x) {
  int deal = turn;                      // without a counter: blockIdx.x, blockIdx.x + gridDim.x, ...
  if (job_counter) {
    if (tid == 0) *next_job = atomicAdd(job_counter, 1u);
    __syncthreads();                    // (the barrier that ends the previous job's last stage keeps this write behind every read of it)
    deal = (int)*next_job;
  }
  if (deal >= n_jobs || deal < 0) break;
#ifndef ME_FRAC_FORWARD
  const int jb = me_frac_deal(deal, n_jobs, prep);
#else
  const int jb = deal;
#endif
#ifdef ME_FRAC_T_TIMELINE   // timing-only build: when does each job start and end (100 MHz wall clock), in which workgroup, and its phases
  const unsigned long long t_job0 = wall_clock64();
  const unsigned long long c_job0 = clock64();   // shader clock: cycles / wall time = the clock the job ran at
  unsigned long long t_ph[7];   // set-up | per stage: lists, items, winners
  int n_ph = 0;
#define ME_FRAC_STAMP() t_ph[n_ph++] = wall_clock64()
#else
#define ME_FRAC_STAMP()
#endif
  MeJob job;
  if (jobs) {   // a table (job-walking launches, HMME_FRAC_JOB_TABLE, CTU ranges beyond 16 bits), else the job derived here
    job = jobs[jb];
  } else {   // job jb = pair jb / ctu_count, CTU ctu_first + jb % ctu_count
    const int ctu_first = (int)(prep.ctus & 0xffff), ctu_count = (int)(prep.ctus >> 16), pic_w = (int)(prep.dims & 0xffff), pic_h = (int)(prep.dims >> 16);
    const int r = jb / ctu_count;
    const int ctus_x = (pic_w + 63) >> 6, n_ctu = ctus_x * ((pic_h + 63) >> 6);
    const int ctu = ctu_first + (jb - r * ctu_count);
    const int cu_x = (ctu % ctus_x) * 64, cu_y = (ctu / ctus_x) * 64;
    const long pq = 2 * ((long)r * n_ctu + ctu);
    const int px = prep.pred_q ? prep.pred_q[pq] : 0, py = prep.pred_q ? prep.pred_q[pq + 1] : 0;
    int ltx, lty, rbx, rby;
    set_search_range(px, py, prep.sr, cu_x, cu_y, pic_w, pic_h, ltx, lty, rbx, rby);
    job.ctu_x = (int16_t)(cu_x | r); job.ctu_y = (int16_t)cu_y;
    job.lt_x = (int16_t)ltx; job.lt_y = (int16_t)lty; job.rb_x = (int16_t)rbx; job.rb_y = (int16_t)rby;
    job.pred_x = (int16_t)px; job.pred_y = (int16_t)py;
  }
  const uint8_t* __restrict__ ref_base = refs.base[job.ctu_x & 63];
  const uint8_t* __restrict__ cur_base = curs.base[job.ctu_x & 63];
  job.ctu_x &= ~63;
  const int16_t* mvs = int_mv + (long)jb * kParts * 2;

  // integer MVs outside the CTU's window (not produced by the search) are clamped to it: the patch stays inside the LDS window.
  // One dword per slot (a TComMv), the three loads of a thread in flight together (593 = 2 * 256 + 81)
  {
    static_assert(kParts <= 3 * NT, "slot state set-up assumes three slots per thread");
    const uint32_t* mvw = (const uint32_t*)mvs;
    uint32_t w[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) w[k] = tid + k * NT < kParts ? mvw[tid + k * NT] : 0u;
    for (int i = tid; i < frac_acc_dw(BPS) / 4; i += NT) ((uint4*)acc)[i] = make_uint4(0, 0, 0, 0);
    const int ltx = job.lt_x, lty = job.lt_y, rbx = job.rb_x, rby = job.rb_y;   // ints: min / max of an int and an int16_t resolve to the double overloads
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (tid + k * NT < kParts) {
        const int mx = min(max((int)(int16_t)(w[k] & 0xffff), ltx), rbx), my = min(max((int)(int16_t)(w[k] >> 16), lty), rby);
        st[tid + k * NT] = (uint32_t)(mx - ltx) | (uint32_t)(my - lty) << 9 | 1u << 18 | 1u << 20;
      }
  }
  if (tid < 2) counter[tid] = 0;
  for (int i = tid; i < 256 * BPS; i += NT) {
    const int r = i / (4 * BPS), q = i - r * (4 * BPS);
    *(uint4*)&curl[r * 16 * BPS + 4 * q] = *(const uint4*)(cur_base + (long)(job.ctu_y + r) * cur_pitch + job.ctu_x * BPS + 16 * q);
  }
  const uint8_t* src = ref_base + (long)(job.ctu_y + job.lt_y - 4) * ref_pitch + (job.ctu_x + job.lt_x - 4) * BPS;   // window sample (-4,-4)
  __syncthreads();
  ME_FRAC_STAMP();

#pragma unroll 1
  for (int stage = 0; stage < 2; ++stage) {
    ME_FRAC_STAMP();   // (timeline builds: the list phase is no phase of its own any more -- its stamp is the stage's start)
    if (stage == 0) me_frac_stage<0, HAD, BPS, WP>(src, ref_pitch, curl, st, cover, list8, list4, counter, pf, tid, bd, clip_lo, wp, tab_h, tab_v, acc);
    else me_frac_stage<1, HAD, BPS, WP>(src, ref_pitch, curl, st, cover, list8, list4, counter, pf, tid, bd, clip_lo, wp, tab_h, tab_v, acc);
    __syncthreads();
    ME_FRAC_STAMP();
    if (tid < 2) counter[tid] = 0;
    // winners (a rolled loop on purpose: with the three slots of a thread unrolled the kernel's register demand rises past 256)
    constexpr int ph[9][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, -1}, {1, -1}, {-1, 1}, {1, 1}};
    constexpr int pq[9][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {-1, 0}, {1, 0}, {-1, 1}, {1, 1}};
#pragma unroll 1
    for (int k = 0; k < 3; ++k) {
      const int s = tid + k * NT;
      if (s >= kParts) break;
      const uint32_t sv = st[s];
      uint32_t sums1[9];
      if constexpr (frac_pack3(BPS)) {
        const unsigned long long* a = (const unsigned long long*)acc + s * 3;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const unsigned long long v = a[q];
          sums1[3 * q] = (uint32_t)v & 0x1fffffu; sums1[3 * q + 1] = (uint32_t)(v >> 21) & 0x1fffffu; sums1[3 * q + 2] = (uint32_t)(v >> 42);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) sums1[i] = acc[s * 9 + i];
      }
      const int mx = (int)(sv & 0x1ff) + job.lt_x, my = (int)((sv >> 9) & 0x1ff) + job.lt_y;
      const int hx = stage ? (int)((sv >> 18) & 3) - 1 : 0, hy = stage ? (int)((sv >> 20) & 3) - 1 : 0;
      const int bxq = 4 * mx + 2 * hx, byq = 4 * my + 2 * hy;   // centre of this stage in quarter units
      uint32_t best = 0xffffffffu, best_acc = 0;
      int bi = 0;
      // the nine points are three x- and three y-offsets: six xGetComponentBits instead of eighteen
      const int step = stage ? 1 : 2;
      uint32_t bits_x[3], bits_y[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        bits_x[t] = me_component_bits(bxq + (t - 1) * step - job.pred_x);
        bits_y[t] = me_component_bits(byq + (t - 1) * step - job.pred_y);
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        // the point orders of the two stages differ in entries 3..6 (s_acMvRefineH / s_acMvRefineQ, TEncSearch.cpp:51-75)
        const uint32_t bx_h = bits_x[ph[i][0] + 1], by_h = bits_y[ph[i][1] + 1], bx_q = bits_x[pq[i][0] + 1], by_q = bits_y[pq[i][1] + 1];
        const uint32_t bits = (ph[i][0] == pq[i][0] && ph[i][1] == pq[i][1]) ? bx_h + by_h : (stage ? bx_q + by_q : bx_h + by_h);
        // whole-PU distortion >> (bitDepth - 8) (TComRdCost.cpp:520-521, :1604), then the MV cost (getCost: uint32 product >> 16)
        const uint32_t a = sums1[i];
        const uint32_t d = (a >> (bd - 8)) + ((lambda_q16 * bits) >> 16);
        if (d < best) { best = d; bi = i; best_acc = a; }
      }
      if (stage == 0) {
        st[s] = (sv & kFracKey0) | (uint32_t)(ph[bi][0] + 1) << 18 | (uint32_t)(ph[bi][1] + 1) << 20;
        // this thread owns row s of acc between the two barriers around this loop: it leaves the row as stage 1 needs it -- point 0 (the
        // centre of the quarter-pel stage IS the half-pel winner: nothing adds to it in stage 1) carried over, the other eight cleared
        if constexpr (frac_pack3(BPS)) {
          unsigned long long* a = (unsigned long long*)acc + s * 3;
          a[0] = best_acc; a[1] = 0; a[2] = 0;
        } else {
          acc[s * 9] = best_acc;
#pragma unroll
          for (int i = 1; i < 9; ++i) acc[s * 9 + i] = 0;
        }
      } else {
        const long o = (long)jb * kParts + s;
        out_qmv[2 * o] = (int16_t)(bxq + pq[bi][0]);
        out_qmv[2 * o + 1] = (int16_t)(byq + pq[bi][1]);
        out_cost[o] = best;
      }
    }
    __syncthreads();   // slot states, cleared sums and list counters are in place before the quarter-pel stage lists its work / the next job starts
    ME_FRAC_STAMP();
  }
#ifdef ME_FRAC_T_TIMELINE
  if (tid == 0) {
    const unsigned long long t1 = wall_clock64();
    uint32_t* o = out_cost + (long)jb * kParts;
    o[0] = (uint32_t)t_job0; o[1] = (uint32_t)(t_job0 >> 32); o[2] = (uint32_t)t1; o[3] = (uint32_t)(t1 >> 32); o[4] = blockIdx.x;
    for (int i = 0; i < 7; ++i) o[5 + i] = (uint32_t)(t_ph[i] - (i ? t_ph[i - 1] : t_job0));   // phase durations, 10 ns units
    o[12] = (uint32_t)(clock64() - c_job0);
  }
  __syncthreads();
#endif
  }   // jobs of this workgroup
}

}  // namespace hmme
