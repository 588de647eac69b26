// hmme.hip -- host side of the C ABI declared in include/hmme.h (see that header for the
// reference interface each entry point replaces).  Plain HIP runtime; no OpenCL, no fallback path.
#include "../../include/hmme.h"
#include "../../include/hmme_test.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "me_kernels.hpp"

using hmme::MeJob;
using hmme::MeJob16;
using hmme::RefSet;

namespace {
constexpr int kMarginX = 128;  // samples; >= 72 needed by clipMv's bounds (+3 for dword staging); keeps CTU rows 64B-aligned
constexpr int kMarginY = 80;   // TComPicYuv: maxCUHeight + 16 (reference TComPicYuv.cpp:91-92)
constexpr int kWinPitch = 1024;  // per-CTU path: bytes per packed window row (>= 2 * (257 + 63) + 8)
constexpr int kRefineHalo = 4;   // samples the 8-tap interpolation reaches beyond the window on every side (TComInterpolationFilter.cpp:170-260)
constexpr int kWinRows = 2 * 128 + 1 + 63 + 2 * kRefineHalo;
// per-CTU call block: up to 64 MeJob16, the job's first strip (always 0), the 593-entry 64-bit merge table (all ones),
// the 64 x 64 current block (1 or 2 bytes per sample), the caller's integer MVs (refine-only call), the packed window.
// kCallFracJob: the whole-window MeJob of the refinement kernel (jobs[] may hold window tiles)
constexpr size_t kCallJobs = 0, kCallFirst = 1536, kCallFracJob = 1600, kCallBest = 2048, kCallCtu = 7168, kCallImv = 15360, kCallWin = 17920;
constexpr int kCallMaxJobs = 64;
static_assert(sizeof(MeJob16) * kCallMaxJobs <= kCallFirst && kCallFracJob + sizeof(MeJob) <= kCallBest && kCallBest + 8 * HMME_NUM_CTU_PARTS <= kCallCtu &&
              kCallCtu + 64 * 64 * 2 <= kCallImv && kCallImv + 4 * HMME_NUM_CTU_PARTS <= kCallWin && kCallWin % 256 == 0, "per-CTU call block layout");
// pinned result block: 593 MVs, 593 SADs, completion word of the search; 593 quarter-pel MVs, 593 costs, completion word of the refinement
constexpr size_t kResMv = 0, kResSad = 4 * HMME_NUM_CTU_PARTS, kResDone = 8 * HMME_NUM_CTU_PARTS, kResQmv = kResDone + 64,
                 kResCost = kResQmv + 4 * HMME_NUM_CTU_PARTS, kResDone2 = kResCost + 4 * HMME_NUM_CTU_PARTS, kResBytes = kResDone2 + 64;
// per workgroup of the 16-bit path -> 2 workgroups per CU (HMME_LDS_BUDGET16: A/B knob in bytes)
// A value that cannot hold a one-row strip at the larger pitch (or exceeds the 160 KB of a CU) is ignored with a message: strips_for()
// would otherwise never terminate / every launch would fail.
constexpr size_t kLdsBudget16Default = 78 * 1024;
size_t lds_budget16_from_env();
const size_t kLdsBudget16 = lds_budget16_from_env();
// LDS window pitch of the 16-bit kernel in dwords -- a template parameter of the kernel (the odd row of a row pair is addressed by
// an immediate).  A lane reads dwords 3 * (lane in row) + 0..33 of its window row, lanes-per-row L = ceil(ceil(wx / 2) / 3), so a
// row needs 3 * (L - 1) + 34 dwords; the pitch also decides which banks the rows of one wave-wide read share (64 lanes cover 1.5 .. 6
// window rows): with pitch == 3 * L (mod 32) the 64 lanes form ONE stride-3 progression over the 32 banks and a ds_read2_b32 costs
// its minimum.  Measured (profiles/archive/r02Y_pdw_sweep.txt, GSAD/s at 2160p 10-bit): SR 128 -- 160: 1 782, 161: 1 812, 162: 1 780,
// 164..170: 1 698..1 724; SR 64 -- 130: 1 687, 97..106 otherwise: 1 630..1 651.  Four pitches are compiled, the right ones for the
// full windows of SR 32 / 64 / 96 / 128 (round 2 had two, and SR 32 / 96 ran 5 % below the tuned ranges); any other window takes the
// smallest pitch that holds it, preferring the right residue (pick_pdw16).
constexpr int kPdw16[4] = {65, 130, 131, 161};
constexpr int kPdw16Large = kPdw16[3];
inline int pdw16_index(int pdw) { return pdw == kPdw16[0] ? 0 : pdw == kPdw16[1] ? 1 : pdw == kPdw16[2] ? 2 : 3; }
// pitch for windows up to wx candidates wide
int pick_pdw16(int wx) {
  const int lanes = (((wx + 1) >> 1) + 2) / 3, need = 3 * (lanes - 1) + 34, want = (3 * lanes) & 31;
  int best = kPdw16Large;
  bool best_match = false;
  for (int i = 3; i >= 0; --i) {
    if (kPdw16[i] < need) continue;
    const bool match = (kPdw16[i] & 31) == want;
    if (match || !best_match) { best = kPdw16[i]; best_match = match; }   // descending: the smallest fitting one wins among equals
  }
  return best;
}
thread_local std::string g_create_error;   // hmme_last_error(NULL): per host thread, like the contexts themselves
constexpr size_t lds_bytes16_c(int pdw, int strip_rows) { return (size_t)(2 * 594 + 4 + (strip_rows + 63) * pdw) * 4; }
size_t lds_budget16_from_env() {
  const char* e = std::getenv("HMME_LDS_BUDGET16");
  if (!e) return kLdsBudget16Default;
  const long v = std::atol(e);
  if (v < (long)lds_bytes16_c(kPdw16Large, 1) || v > 160 * 1024) {
    fprintf(stderr, "hmme: HMME_LDS_BUDGET16=%s outside [%zu, %d] bytes: using the default %zu\n", e, lds_bytes16_c(kPdw16Large, 1), 160 * 1024,
            kLdsBudget16Default);
    return kLdsBudget16Default;
  }
  return (size_t)v;
}
}  // namespace

struct hmme_ctx {
  int device = 0;
  int sr_max = 64;
  uint32_t lambda_q16 = 0;
  std::string err;
  bool print_errors = true;   // hmme_set_error_printing
  std::string info;
  hipStream_t stream = nullptr;   // private stream of the synchronous entry points
  // frame-path scratch (job tables, merge table, cover table) is shared by every call of the context: a call on another stream
  // than the previous one first waits (stream-side) for that one's last use
  // ONE event is recorded behind the kernels of a launch (ring of kLaunchEvents, launch_end): it marks the scratch's last use and the last
  // read of every plane the launch took (hmme_plane::read_done points into the ring).  A ring entry that has been recorded again
  // since stands for a LATER point of the same chain of launches -- every launch of a context is ordered behind the one before it
  // through the scratch -- so a waiter on a recycled entry waits longer than it had to, never too little.
  static constexpr int kLaunchEvents = 64;
  hipEvent_t launch_ev[kLaunchEvents] = {};
  unsigned launch_seq = 0;
  hipEvent_t scratch_done = nullptr;   // the ring entry of the last launch
  hipStream_t scratch_stream = nullptr;
  bool scratch_used = false;
  // per-CTU path: one device block and its pinned host mirror -- jobs, first-strip index, merge table preset, current
  // block, window -- so that a call is one upload, the kernels, one download (layout: kCall* offsets below)
  uint8_t* d_call = nullptr;
  uint8_t* h_call = nullptr;
  uint8_t* h_call_dev = nullptr;  // device-side address of h_call (mapped pinned memory)
  uint8_t* h_res = nullptr;       // 593 MVs (int16 x, y), 593 SADs, then the completion word: pinned host memory the finalize kernel writes
  uint8_t* d_res = nullptr;       // ... and its device-side address
  uint32_t call_seq = 0;
  // frame path scratch (grown on demand)
  void* d_jobs = nullptr;         // MeJob[] or MeJob16[]
  size_t jobs_bytes = 0;
  // A job table is a function of the picture size, the search range, the CTU range, the number of pairs and the predictors.  Launches
  // WITHOUT predictors (d_pred_q == null: the zero predictor of the reference's own call) of the same geometry on the same stream find
  // the table of the launch before still in place and skip the kernels that write it -- one kernel launch less per search step
  // (HMME_NO_TABLE_CACHE=1: always rebuild).  `buf` / `buf2`: the allocations the table lives in (a reallocation invalidates it)
  struct TableTag {
    bool valid = false;
    int w = 0, h = 0, bit_depth = 0, sr = 0, first = 0, count = 0, pairs = 0;
    const void* buf = nullptr; const void* buf2 = nullptr; void* stream = nullptr;
    bool same(const TableTag& o) const {
      return valid && o.valid && w == o.w && h == o.h && bit_depth == o.bit_depth && sr == o.sr && first == o.first && count == o.count && pairs == o.pairs &&
             buf == o.buf && buf2 == o.buf2 && stream == o.stream;
    }
  };
  TableTag jobs_tag;              // what d_jobs / d_first_strip hold (searches)
  void* d_frac_jobs = nullptr;    // MeJob[] of the refinement launches that read a table (+ the job counter of the job-walking mode)
  size_t frac_jobs_bytes = 0;
  TableTag frac_jobs_tag;
  int wg_slots = 512;     // search workgroups resident at once: 2 per CU (VGPRs of the search kernels, LDS of the 16-bit one)
  int* d_first_strip = nullptr;
  int first_strip_cap = 0;
  unsigned long long* d_best = nullptr;   // [jobs][593] merge table of the split / strip / segment launches
  size_t best_cap = 0;
  bool best_clean = false;   // every entry of d_best is all ones: me_finalize16_kernel resets what it decodes, so only a table that is new, has
                             // grown or was left behind by a failed launch needs the preset (merge_table)
  int16_t* d_pred = nullptr;
  int16_t* d_mv = nullptr;
  uint32_t* d_sad = nullptr;
  int out_cap = 0;
  int* d_flag = nullptr;
  bool lds_optin[8] = {false, false, false, false, false, false, false, false};
  int num_cus = 0;
  bool frac_lds_optin[3][2] = {{false, false}, {false, false}, {false, false}};   // [8-bit | u16 | u16 weighted][hadamard]
  int frac_wg_per_cu[2][2] = {{0, 0}, {0, 0}};   // [wide][hadamard] workgroups of me_frac_kernel a CU holds (runtime occupancy query, first use)
  uint8_t* d_wwin = nullptr;          // per-CTU calls with weighted prediction: the weighted copy of the staged window (the search's)
  uint16_t* d_frac_cover = nullptr;   // fractional refinement: slots covering each 8x8 / 4x4 position, same for every CTU
  int16_t* d_imv = nullptr;           // host-facing refine call: integer MVs / quarter-pel MVs / costs on the device
  int16_t* d_qmv = nullptr;
  uint32_t* d_fcost = nullptr;
  size_t refine_cap = 0;
};

struct hmme_plane {
  hmme_ctx* ctx = nullptr;
  int width = 0, height = 0;
  int bit_depth = 8;
  int bps = 1;            // bytes per sample: 1 (8-bit path) or 2 (9..12 bit)
  int pitch = 0;          // bytes per row, multiple of 256
  int rows = 0;           // height + 2 * kMarginY
  uint8_t* d_data = nullptr;
  // the picture area once more, CTU by CTU: block (cx, cy) = 64 rows x 64 samples, contiguous, raster order of CTUs, partial edge CTUs
  // completed by edge replication -- what the search kernels read the CURRENT picture from (scalar loads at immediate offsets, whatever
  // the picture's size: me_kernels.hpp me_prefetch_cur).  Written by every fill right behind the padded plane (plane_fill).
  uint8_t* d_blocks = nullptr;
  int ctus_x = 0, n_ctu = 0;
  void* d_stage = nullptr;  // device staging for uploads
  size_t stage_bytes = 0;
  hipEvent_t filled = nullptr;      // recorded after the last fill; readers on another stream wait for it
  hipStream_t fill_stream = nullptr;
  bool fill_pending = false;
  // the event of the last launch that reads the plane (an entry of the context's ring, hmme_ctx::launch_ev: not owned); a refill on
  // another stream waits for it (write after read).  mutable: reading a plane does not change what it holds
  mutable hipEvent_t read_done = nullptr;
  mutable hipStream_t read_stream = nullptr;
  mutable bool read_pending = false;
  const uint8_t* origin() const { return d_data + (size_t)kMarginY * pitch + (size_t)kMarginX * bps; }
};

namespace {

int fail(hmme_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf; else g_create_error = buf;
  if (!ctx || ctx->print_errors) fprintf(stderr, "hmme: ERROR: %s\n", buf);   // TEncOpenCL::checkError prints too (TEncOpenCL.h:93-101)
  return code;
}

#define HIP_TRY(ctx, call)                                                                              \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) return fail(ctx, HMME_ERR_DEVICE, "%s -> %s", #call, hipGetErrorString(e_));   \
  } while (0)

template <typename T>
int ensure(hmme_ctx* ctx, T** p, size_t* cap_bytes, size_t bytes, size_t grow_to = 0) {
  if (bytes <= *cap_bytes) return HMME_OK;
  // hipFree synchronises the whole device -- in the middle of a streaming pipeline that stalls the copy and download streams.  A
  // scratch that has to grow therefore grows to `grow_to` at once (the callers pass what kMaxRefs pairs of this picture size need:
  // KBs to a few MB), so a steady stream meets this path on its first launch of a picture size and never again.
  if (grow_to > bytes) bytes = grow_to;
  if (*p) hipFree(*p);
  *p = nullptr; *cap_bytes = 0;
  HIP_TRY(ctx, hipMalloc((void**)p, bytes));
  *cap_bytes = bytes;
  return HMME_OK;
}

// cross-stream ordering (see hmme.h "Streams"): scratch of the context, contents of a plane
int scratch_acquire(hmme_ctx* ctx, hipStream_t s) {
  if (ctx->scratch_used && ctx->scratch_stream != s) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->scratch_done, 0));
  return HMME_OK;
}
int launch_end(hmme_ctx* ctx, hipStream_t s) {
  hipEvent_t ev = ctx->launch_ev[ctx->launch_seq++ % hmme_ctx::kLaunchEvents];
  HIP_TRY(ctx, hipEventRecord(ev, s));
  ctx->scratch_done = ev; ctx->scratch_stream = s; ctx->scratch_used = true;
  return HMME_OK;
}
int plane_wait(hmme_ctx* ctx, const hmme_plane* pl, hipStream_t s) {
  if (pl->fill_pending && pl->fill_stream != s) HIP_TRY(ctx, hipStreamWaitEvent(s, pl->filled, 0));
  return HMME_OK;
}
// a search / refinement on `s` has been enqueued that reads `pl`.  One event holds the LAST read only, so a read on a new stream
// first makes that stream wait for the previous reader (plane_read_chain, before the launch's event is recorded): the event then
// covers both.  plane_read_mark: the launch's event (launch_end) is the plane's last read
int plane_read_chain(hmme_ctx* ctx, const hmme_plane* pl, hipStream_t s) {
  if (pl->read_pending && pl->read_stream != s) HIP_TRY(ctx, hipStreamWaitEvent(s, pl->read_done, 0));
  return HMME_OK;
}
void plane_read_mark(hmme_ctx* ctx, const hmme_plane* pl, hipStream_t s) {
  pl->read_done = ctx->scratch_done; pl->read_stream = s; pl->read_pending = true;
}
// before `s` overwrites the plane (or its staging buffer): the last fill and the last reader, if they ran on other streams
int plane_write_wait(hmme_ctx* ctx, const hmme_plane* pl, hipStream_t s) {
  if (pl->fill_pending && pl->fill_stream != s) HIP_TRY(ctx, hipStreamWaitEvent(s, pl->filled, 0));
#ifndef HMME_TEST_NO_WAR_WAIT   // tools/build_variant.sh nowar -DHMME_TEST_NO_WAR_WAIT: shows that the two-stream refill test fails without this wait
  if (pl->read_pending && pl->read_stream != s) HIP_TRY(ctx, hipStreamWaitEvent(s, pl->read_done, 0));
#endif
  return HMME_OK;
}

// ---- 8-bit path --------------------------------------------------------------------------------------
RefSet one_ref(const uint8_t* base) {
  RefSet r;
  for (int i = 0; i < hmme::kMaxRefs; ++i) r.base[i] = base;
  return r;
}

// Progress-based wave priorities of the search kernels (me_search_kernel, ME_FAIR_PRIO): the kernel whose workgroups are whole CTU
// searches always runs with them (+8 % on a single-round 1080p launch, +2 % at 2160p: the lonely ends of each CU's last workgroups);
// the launches of many small workgroups -- split tasks, window tiles, 16-bit strips -- only while the launch is a few rounds of the
// chip's workgroup slots (720p +5 %, 1080p 10-bit +5 %; config 5's 16 rounds of strips lost 0.8 % with them:
// profiles/r05d_fair_priority_ab.txt).  HMME_FAIR_PRIO=0|1 forces one for A/B runs.
int fair_prio(const hmme_ctx* ctx, int workgroups, bool whole_jobs) {
  static const int force = std::getenv("HMME_FAIR_PRIO") ? std::atoi(std::getenv("HMME_FAIR_PRIO")) : -1;
  if (force >= 0) return force ? 1 : 0;
  return (whole_jobs || workgroups <= 4 * ctx->wg_slots) ? 1 : 0;
}

int launch_search8(hmme_ctx* ctx, const RefSet& cur, int cur_ctus_x, const RefSet& ref, int ref_pitch, const MeJob* d_jobs,
                   int n_jobs, int fen, int16_t* d_mv, uint32_t* d_sad, hipStream_t stream) {
  if (n_jobs <= 0) return HMME_OK;
  const int fair = fair_prio(ctx, n_jobs, true);
  if (fen)
    hipLaunchKernelGGL((hmme::me_search_kernel<1, 0>), dim3(n_jobs), dim3(hmme::kThreads), 0, stream, cur, cur_ctus_x, ref,
                       ref_pitch, (const void*)d_jobs, ctx->lambda_q16, d_mv, d_sad, (unsigned long long*)nullptr, fair, 0);
  else
    hipLaunchKernelGGL((hmme::me_search_kernel<0, 0>), dim3(n_jobs), dim3(hmme::kThreads), 0, stream, cur, cur_ctus_x, ref,
                       ref_pitch, (const void*)d_jobs, ctx->lambda_q16, d_mv, d_sad, (unsigned long long*)nullptr, fair, 0);
  HIP_TRY(ctx, hipGetLastError());
  return HMME_OK;
}

int finalize_best(hmme_ctx* ctx, unsigned long long* d_best, const MeJob16* d_jobs, const int* d_first_strip, int n_jobs, int16_t* d_mv,
                  uint32_t* d_sad, hipStream_t stream);

// 8-bit split mode: n_jobs * n_split workgroups, each runs a slice of its CTU's tasks and merges through ctx->d_best
// the 64-bit merge table of a split / strip / tile launch: the caller's (already all ones), or ctx->d_best grown and preset here
int merge_table(hmme_ctx* ctx, int n_jobs, unsigned long long* preset, hipStream_t stream, unsigned long long** table) {
  if (preset) { *table = preset; return HMME_OK; }
  size_t cap = ctx->best_cap;
  int rc = ensure(ctx, &ctx->d_best, &cap, sizeof(unsigned long long) * HMME_NUM_CTU_PARTS * (size_t)n_jobs,
                  n_jobs > 64 ? sizeof(unsigned long long) * HMME_NUM_CTU_PARTS * (size_t)n_jobs * 2 : 0);
  if (cap != ctx->best_cap) ctx->best_clean = false;
  ctx->best_cap = cap;
  if (rc) return rc;
  if (!ctx->best_clean) HIP_TRY(ctx, hipMemsetAsync(ctx->d_best, 0xFF, ctx->best_cap, stream));
  ctx->best_clean = false;   // until the launch's finalize_best has been enqueued (it sets every entry it decodes back to all ones)
  *table = ctx->d_best;
  return HMME_OK;
}

int launch_search8_split(hmme_ctx* ctx, const RefSet& cur, int cur_ctus_x, const RefSet& ref, int ref_pitch, const MeJob16* d_jobs,
                         const int* d_first_strip, int n_jobs, int n_split, int fen, int16_t* d_mv, uint32_t* d_sad,
                         hipStream_t stream, unsigned long long* preset_best = nullptr, bool finalize = true) {
  if (n_jobs <= 0) return HMME_OK;
  unsigned long long* best = nullptr;
  int rc = merge_table(ctx, n_jobs, preset_best, stream, &best);
  if (rc) return rc;
  const int fair = fair_prio(ctx, n_jobs * n_split, false);
  if (fen)
    hipLaunchKernelGGL((hmme::me_search_kernel<1, 1>), dim3(n_jobs * n_split), dim3(hmme::kThreads), 0, stream, cur, cur_ctus_x,
                       ref, ref_pitch, (const void*)d_jobs, ctx->lambda_q16, (int16_t*)nullptr, (uint32_t*)nullptr, best, fair, 0);
  else
    hipLaunchKernelGGL((hmme::me_search_kernel<0, 1>), dim3(n_jobs * n_split), dim3(hmme::kThreads), 0, stream, cur, cur_ctus_x,
                       ref, ref_pitch, (const void*)d_jobs, ctx->lambda_q16, (int16_t*)nullptr, (uint32_t*)nullptr, best, fair, 0);
  HIP_TRY(ctx, hipGetLastError());
  return finalize ? finalize_best(ctx, best, d_jobs, d_first_strip, n_jobs, d_mv, d_sad, stream) : HMME_OK;
}

// 8-bit segment mode: the tasks of n_jobs jobs in n_wg equal segments (table: me_seg_table_*), merged through `best` (preset by the caller)
int launch_search8_segments(hmme_ctx* ctx, const RefSet& cur, int cur_ctus_x, const RefSet& ref, int ref_pitch, const void* d_table,
                            const int* d_first_strip, int n_jobs, int n_wg, int fen, int16_t* d_mv, uint32_t* d_sad, hipStream_t stream,
                            unsigned long long* best) {
  if (n_jobs <= 0 || n_wg <= 0) return HMME_OK;
  const int fair = fair_prio(ctx, n_wg, false);
  if (fen)
    hipLaunchKernelGGL((hmme::me_search_kernel<1, 2>), dim3(n_wg), dim3(hmme::kThreads), 0, stream, cur, cur_ctus_x, ref, ref_pitch, d_table,
                       ctx->lambda_q16, (int16_t*)nullptr, (uint32_t*)nullptr, best, fair, n_jobs);
  else
    hipLaunchKernelGGL((hmme::me_search_kernel<0, 2>), dim3(n_wg), dim3(hmme::kThreads), 0, stream, cur, cur_ctus_x, ref, ref_pitch, d_table,
                       ctx->lambda_q16, (int16_t*)nullptr, (uint32_t*)nullptr, best, fair, n_jobs);
  HIP_TRY(ctx, hipGetLastError());
  return finalize_best(ctx, best, hmme::me_seg_table_jobs(d_table, n_wg), d_first_strip, n_jobs, d_mv, d_sad, stream);
}

// ---- 16-bit path -------------------------------------------------------------------------------------

size_t lds_bytes16(int pdw, int strip_rows) { return lds_bytes16_c(pdw, strip_rows); }

// most candidate rows of one strip whose window rows (+ 63) fit the LDS budget
int rows_max16(int pdw) {
  int r = 1;
  while (lds_bytes16(pdw, r + 1) <= kLdsBudget16) ++r;
  return r;
}

// strips of candidate rows so that one strip's window rows fit the LDS budget
int strips_for(int pdw, int wy_max) {
  int n = 1;
  while (n < wy_max && lds_bytes16(pdw, (wy_max + n - 1) / n) > kLdsBudget16) ++n;   // bounded: one-row strips always fit (budget is validated)
  return n;
}

template <int FEN, int PDW>
int launch16_t(hmme_ctx* ctx, const RefSet& cur, int cur_ctus_x, const RefSet& ref, int ref_pitch, const MeJob16* d_jobs, int n_wg,
               size_t lds, int sh, unsigned long long* d_best, hipStream_t stream) {
  bool& attr_set = ctx->lds_optin[FEN * 4 + pdw16_index(PDW)];   // > 64 KiB of dynamic LDS: opt in once per
  if (!attr_set) {                                                            // kernel and device (= per context)
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)hmme::me_search16_kernel<FEN, PDW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((hmme::me_search16_kernel<FEN, PDW>), dim3(n_wg), dim3(hmme::kThreads16), lds, stream, cur, cur_ctus_x, ref,
                     ref_pitch, d_jobs, ctx->lambda_q16, sh, d_best, fair_prio(ctx, n_wg, false));
  HIP_TRY(ctx, hipGetLastError());
  return HMME_OK;
}

// d_jobs: n_jobs * n_strips MeJob16; results merged in ctx->d_best then decoded into d_mv / d_sad
int launch_search16(hmme_ctx* ctx, const RefSet& cur, int cur_ctus_x, const RefSet& ref, int ref_pitch, const MeJob16* d_jobs,
                    const int* d_first_strip, int n_jobs, int n_wg, int pdw, int strip_rows_max, int fen, int bit_depth,
                    int16_t* d_mv, uint32_t* d_sad, hipStream_t stream, unsigned long long* preset_best = nullptr, bool finalize = true) {
  if (n_jobs <= 0) return HMME_OK;
  unsigned long long* best = nullptr;
  int rc = merge_table(ctx, n_jobs, preset_best, stream, &best);
  if (rc) return rc;
  const size_t lds = lds_bytes16(pdw, strip_rows_max);
  const int sh = bit_depth - 8;
#define LAUNCH16(I)                                                                                                       \
  rc = fen ? launch16_t<1, kPdw16[I]>(ctx, cur, cur_ctus_x, ref, ref_pitch, d_jobs, n_wg, lds, sh, best, stream)            \
           : launch16_t<0, kPdw16[I]>(ctx, cur, cur_ctus_x, ref, ref_pitch, d_jobs, n_wg, lds, sh, best, stream)
  switch (pdw16_index(pdw)) {
    case 0: LAUNCH16(0); break;
    case 1: LAUNCH16(1); break;
    case 2: LAUNCH16(2); break;
    default: LAUNCH16(3); break;
  }
#undef LAUNCH16
  if (rc) return rc;
  return finalize ? finalize_best(ctx, best, d_jobs, d_first_strip, n_jobs, d_mv, d_sad, stream) : HMME_OK;
}

int finalize_best(hmme_ctx* ctx, unsigned long long* d_best, const MeJob16* d_jobs, const int* d_first_strip, int n_jobs, int16_t* d_mv,
                  uint32_t* d_sad, hipStream_t stream) {
  const long total = (long)n_jobs * HMME_NUM_CTU_PARTS;
  hipLaunchKernelGGL(hmme::me_finalize16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, d_best, d_jobs,
                     d_first_strip, n_jobs, ctx->lambda_q16, d_mv, d_sad);
  HIP_TRY(ctx, hipGetLastError());
  if (d_best == ctx->d_best) ctx->best_clean = true;   // the launches of a context are one chain (scratch_acquire): the next one finds the table reset
  return HMME_OK;
}

int check_frame_args(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* ref, const hmme_frame_params* fp, int* first,
                     int* count) {
  if (!ctx) return HMME_ERR_ARG;
  if (!cur || !ref || !fp) return fail(ctx, HMME_ERR_ARG, "null plane / params");
  // a launch ties the planes it reads to an event of THIS context's ring (plane_read_mark): a plane of another context would be left
  // pointing into that ring after the context is gone
  if (cur->ctx != ctx || ref->ctx != ctx) return fail(ctx, HMME_ERR_ARG, "plane belongs to another context (planes are used with the context that created them)");
  if (cur->width != ref->width || cur->height != ref->height) return fail(ctx, HMME_ERR_ARG, "cur/ref size mismatch");
  if (fp->bit_depth < 8 || fp->bit_depth > 12) return fail(ctx, HMME_ERR_UNSUPPORTED, "bit depth %d outside 8..12", fp->bit_depth);
  if (cur->bit_depth != fp->bit_depth || ref->bit_depth != fp->bit_depth)
    return fail(ctx, HMME_ERR_ARG, "planes hold %d/%d-bit samples, search asks for %d", cur->bit_depth, ref->bit_depth, fp->bit_depth);
  if (fp->search_range < 1 || fp->search_range > ctx->sr_max)
    return fail(ctx, HMME_ERR_ARG, "search range %d outside [1, %d]", fp->search_range, ctx->sr_max);
  const int n = hmme_num_ctus(cur->width, cur->height);
  *first = fp->ctu_first;
  *count = fp->ctu_count < 0 ? n - fp->ctu_first : fp->ctu_count;
  if (*first < 0 || *count < 0 || *first + *count > n) return fail(ctx, HMME_ERR_ARG, "CTU range [%d, +%d) outside 0..%d", *first, *count, n);
  return HMME_OK;
}

// Range violations are latched on the device: flag 0 by the synchronous uploads (taken right behind the fill, on its stream), flag 1
// by the asynchronous ones (taken by hmme_upload_status) -- a bad sample of an asynchronous upload can no longer fail the next
// synchronous upload of another, valid plane.  Read and clear are one atomic exchange on the stream (me_take_flag_kernel).
int take_flag(hmme_ctx* ctx, int which, hipStream_t s, int* out) {
  hipLaunchKernelGGL(hmme::me_take_flag_kernel, dim3(1), dim3(1), 0, s, ctx->d_flag + which, ctx->d_flag + 2);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_flag + 2, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return HMME_OK;
}

template <typename SrcT, typename DstT>
int plane_fill(hmme_plane* pl, const SrcT* d_src, int src_pitch_elems, hipStream_t s, bool check) {
  hmme_ctx* ctx = pl->ctx;
  int wrc = plane_write_wait(ctx, pl, s);   // the previous fill and the last reader, if on other streams
  if (wrc) return wrc;
  dim3 grid((pl->pitch / 4 + 255) / 256, pl->rows);
  hipLaunchKernelGGL((hmme::me_fill_plane_kernel<SrcT, DstT>), grid, dim3(256), 0, s, pl->d_data, pl->pitch, kMarginX, kMarginY,
                     pl->width, pl->height, d_src, src_pitch_elems, (1 << pl->bit_depth) - 1, ctx->d_flag + (check ? 0 : 1));
  HIP_TRY(ctx, hipGetLastError());
  // ... and the CTU-blocked copy the searches read the current picture from (one more pass over the staged picture: 8 MB at 2160p)
  const long blk_threads = (long)pl->n_ctu * (hmme::kBlkBytes8 * pl->bps / 4);
  hipLaunchKernelGGL((hmme::me_fill_blocks_kernel<SrcT, DstT>), dim3((unsigned)((blk_threads + 255) / 256)), dim3(256), 0, s, pl->d_blocks, pl->ctus_x,
                     pl->n_ctu, pl->width, pl->height, d_src, src_pitch_elems);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(pl->filled, s));
  pl->fill_stream = s; pl->fill_pending = true;
  if (check) {
    int flag = 0;
    const int rc = take_flag(ctx, 0, s, &flag);
    if (rc) return rc;
    if (flag) return fail(ctx, HMME_ERR_RANGE, "plane upload: sample outside [0,%d] for a %d-bit plane", (1 << pl->bit_depth) - 1, pl->bit_depth);
  }
  return HMME_OK;
}

// s == nullptr: the synchronous upload (private stream, range check, returns when the plane is filled); otherwise asynchronous on
// `s` -- the copy runs at PCIe rate only from page-locked memory (hmme_host_register) and a range violation is latched for
// hmme_upload_status
template <typename T>
int plane_upload(hmme_plane* pl, const T* origin, int stride, hipStream_t s, bool sync) {
  if (!pl) return HMME_ERR_ARG;
  hmme_ctx* ctx = pl->ctx;
  if (!origin || stride < pl->width) return fail(ctx, HMME_ERR_ARG, "plane upload: bad origin/stride");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (sync) s = ctx->stream;
  const size_t bytes = sizeof(T) * (size_t)pl->width * pl->height;
  if (pl->stage_bytes < bytes) {
    if (pl->fill_pending) HIP_TRY(ctx, hipEventSynchronize(pl->filled));   // a fill may still be reading the old staging buffer
    hipFree(pl->d_stage); pl->d_stage = nullptr; pl->stage_bytes = 0;
    HIP_TRY(ctx, hipMalloc(&pl->d_stage, bytes));
    pl->stage_bytes = bytes;
  }
  int rc = plane_write_wait(ctx, pl, s);   // the staging buffer is read by the previous fill
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy2DAsync(pl->d_stage, sizeof(T) * pl->width, origin, sizeof(T) * (size_t)stride, sizeof(T) * pl->width,
                                pl->height, hipMemcpyHostToDevice, s));
  if (pl->bps == 1) return plane_fill<T, uint8_t>(pl, (const T*)pl->d_stage, pl->width, s, sync);
  return plane_fill<T, uint16_t>(pl, (const T*)pl->d_stage, pl->width, s, sync);
}

}  // namespace

extern "C" {

int hmme_num_ctus(int width, int height) { return ((width + 63) / 64) * ((height + 63) / 64); }

int hmme_create(int device, int sr_max, unsigned flags, hmme_ctx** out) {
  (void)flags;
  if (!out) return fail(nullptr, HMME_ERR_ARG, "hmme_create: out == NULL");
  *out = nullptr;
  if (sr_max < 1 || sr_max > HMME_MAX_SEARCH_RANGE)
    return fail(nullptr, HMME_ERR_UNSUPPORTED, "hmme_create: sr_max %d outside [1, %d]", sr_max, HMME_MAX_SEARCH_RANGE);
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(nullptr, HMME_ERR_DEVICE, "hmme_create: no HIP device (this engine has no CPU fallback)");
  if (device < 0 || device >= n_dev) return fail(nullptr, HMME_ERR_ARG, "hmme_create: device %d of %d", device, n_dev);
  hipDeviceProp_t prop;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess)
    return fail(nullptr, HMME_ERR_DEVICE, "hmme_create: cannot open device %d", device);
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, HMME_ERR_DEVICE, "hmme_create: device %d is %s; the kernels are built for gfx950 only", device, prop.gcnArchName);
  hmme_ctx* ctx = new hmme_ctx;
  ctx->device = device;
  ctx->sr_max = sr_max;
  char info[256];
  // (the marketing name comes from libdrm's amdgpu.ids table, which a minimal install may lack: the string then starts with what the
  // runtime always knows -- the architecture)
  snprintf(info, sizeof info, "%s (%s), %d CUs, %.0f GiB", prop.name[0] ? prop.name : "AMD GPU (no marketing name on this host)", prop.gcnArchName,
           prop.multiProcessorCount, (double)prop.totalGlobalMem / (1 << 30));
  ctx->info = info;
  ctx->wg_slots = 2 * prop.multiProcessorCount;
  ctx->num_cus = prop.multiProcessorCount;
  const size_t win_bytes = (size_t)kWinRows * kWinPitch + 64;
#define CREATE_TRY(call)                                                                                           \
  do {                                                                                                             \
    hipError_t e_ = (call);                                                                                        \
    if (e_ != hipSuccess) { int rc = fail(nullptr, HMME_ERR_NOMEM, "hmme_create: %s -> %s", #call, hipGetErrorString(e_)); hmme_destroy(ctx); return rc; } \
  } while (0)
  CREATE_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  for (int i = 0; i < hmme_ctx::kLaunchEvents; ++i) CREATE_TRY(hipEventCreateWithFlags(&ctx->launch_ev[i], hipEventDisableTiming));
  CREATE_TRY(hipMalloc(&ctx->d_call, kCallWin + win_bytes));
  CREATE_TRY(hipHostMalloc(&ctx->h_call, kCallWin + win_bytes, hipHostMallocMapped));
  CREATE_TRY(hipHostGetDevicePointer((void**)&ctx->h_call_dev, ctx->h_call, 0));
  std::memset(ctx->h_call, 0, kCallWin);
  std::memset(ctx->h_call + kCallBest, 0xFF, 8 * HMME_NUM_CTU_PARTS);
  CREATE_TRY(hipHostMalloc(&ctx->h_res, kResBytes, hipHostMallocMapped));
  std::memset(ctx->h_res, 0, kResBytes);
  CREATE_TRY(hipHostGetDevicePointer((void**)&ctx->d_res, ctx->h_res, 0));
  CREATE_TRY(hipMalloc(&ctx->d_flag, 4 * sizeof(int)));   // [0] synchronous uploads, [1] asynchronous uploads, [2] take_flag's result
  CREATE_TRY(hipMemset(ctx->d_flag, 0, 4 * sizeof(int)));
#undef CREATE_TRY
  *out = ctx;
  return HMME_OK;
}

void hmme_destroy(hmme_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  if (ctx->stream) { hipStreamSynchronize(ctx->stream); hipStreamDestroy(ctx->stream); }
  for (int i = 0; i < hmme_ctx::kLaunchEvents; ++i)
    if (ctx->launch_ev[i]) hipEventDestroy(ctx->launch_ev[i]);
  hipFree(ctx->d_call);
  hipFree(ctx->d_jobs); hipFree(ctx->d_frac_jobs); hipFree(ctx->d_first_strip); hipFree(ctx->d_best);
  hipFree(ctx->d_pred); hipFree(ctx->d_mv); hipFree(ctx->d_sad); hipFree(ctx->d_flag);
  hipFree(ctx->d_wwin); hipFree(ctx->d_frac_cover); hipFree(ctx->d_imv); hipFree(ctx->d_qmv); hipFree(ctx->d_fcost);
  if (ctx->h_call) hipHostFree(ctx->h_call);
  if (ctx->h_res) hipHostFree(ctx->h_res);
  delete ctx;
}

const char* hmme_last_error(const hmme_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }
int hmme_set_error_printing(hmme_ctx* ctx, int on) {
  if (!ctx) return 1;
  const int was = ctx->print_errors ? 1 : 0;
  ctx->print_errors = on != 0;
  return was;
}
const char* hmme_device_info(const hmme_ctx* ctx) { return ctx ? ctx->info.c_str() : ""; }
int hmme_device_index(const hmme_ctx* ctx) { return ctx ? ctx->device : -1; }

int hmme_set_lambda(hmme_ctx* ctx, double lambda) {
  if (!ctx) return HMME_ERR_ARG;
  if (!(lambda >= 0.0)) return fail(ctx, HMME_ERR_ARG, "lambda %g", lambda);
  ctx->lambda_q16 = (uint32_t)std::floor(65536.0 * std::sqrt(lambda));   // TEncOpenCL.h:121 == TComRdCost.cpp:209
  return HMME_OK;
}
int hmme_set_lambda_q16(hmme_ctx* ctx, uint32_t q) {
  if (!ctx) return HMME_ERR_ARG;
  ctx->lambda_q16 = q;
  return HMME_OK;
}
uint32_t hmme_get_lambda_q16(const hmme_ctx* ctx) { return ctx ? ctx->lambda_q16 : 0; }

void hmme_params_ocl_compat(hmme_search_params* p, int lt_x, int lt_y, int sr) {
  // TEncOpenCL.cpp:312-313 scans [0, 2*SR] from LT and never reads RB; cl/sad.cl:374-398 prices the MV
  // against (0,0); no row sub-sampling, no bit-depth shift
  p->lt_x = lt_x; p->lt_y = lt_y; p->rb_x = lt_x + 2 * sr; p->rb_y = lt_y + 2 * sr;
  p->pred_x = 0; p->pred_y = 0; p->fen = 0; p->bit_depth = 8; p->shift_free = 1;
}

void hmme_set_search_range(int pred_x_q, int pred_y_q, int sr, int cu_x, int cu_y, int pic_w, int pic_h, int* lt_x,
                           int* lt_y, int* rb_x, int* rb_y) {
  hmme::set_search_range(pred_x_q, pred_y_q, sr, cu_x, cu_y, pic_w, pic_h, *lt_x, *lt_y, *rb_x, *rb_y);
}

// ---- slot layout: closed form of the 593-case switch of TComDataCU::getIndexBlock ------------------------
namespace {
const int kBase2NxN[4] = {588, 560, 448, 0}, kBaseNx2N[4] = {590, 568, 480, 128};      // indexed by depth (CU 64,32,16,8)
const int kBaseAMP[4] = {576, 512, 256, -1}, kBase2Nx2N[4] = {592, 584, 544, 384};

int slot_of(int part_size, int depth, int part_idx, int cx, int cy) {
  const int n = 1 << depth, r = cy * n + cx;
  if (part_idx < 0 || part_idx > 1) return -1;
  switch (part_size) {
    case 0: return part_idx == 0 ? kBase2Nx2N[depth] + r : -1;
    case 1: return kBase2NxN[depth] + cy * 2 * n + part_idx * n + cx;
    case 2: return kBaseNx2N[depth] + cy * 2 * n + 2 * cx + part_idx;
    default: break;
  }
  if (depth == 3) return -1;
  static const int k_of[4][2] = {{0, 3}, {2, 1}, {4, 7}, {6, 5}};   // 2NxnU, 2NxnD, nLx2N, nRx2N x part_idx -> AMP family
  if (part_size < 4 || part_size > 7) return -1;
  return kBaseAMP[depth] + k_of[part_size - 4][part_idx] * n * n + r;
}
}  // namespace

int hmme_slot_index(int part_size, int depth, int part_idx, int abs_z_idx) {
  if (depth < 0 || depth > 3 || abs_z_idx < 0 || abs_z_idx > 255) return -1;
  int bx = 0, by = 0;   // z-order -> raster in 4x4 units
  for (int b = 0; b < 4; ++b) { bx |= ((abs_z_idx >> (2 * b)) & 1) << b; by |= ((abs_z_idx >> (2 * b + 1)) & 1) << b; }
  const int s4 = 16 >> depth;   // CU size in 4x4 units
  if (bx % s4 || by % s4) return -1;
  return slot_of(part_size, depth, part_idx, bx / s4, by / s4);
}

// ---- the table layout of an encoder built with AMP_ENC_SPEEDUP (TypeDef.h:206, :260-261: NUM_CTU_PARTS 425; cl/sad.cl:4-138 `calcSAD`,
// TComDataCU.cpp:3393-4675).  Dead in the reference tree as shipped (the macro is 0), kept as a VIEW of the 593 tables: the same
// rectangles without the AMP shapes.  Bases per CU size 8 / 16 / 32 / 64: 2NxN 0 / 320 / 400 / 420, Nx2N 128 / 352 / 408 / 422,
// 2Nx2N 256 / 384 / 416 / 424; inside a family the order of the 593 layout -- except that the reference's table numbers the two
// parts of the 64x64 2NxN and Nx2N CUs in reverse (part 1 before part 0: TComDataCU.cpp:3396-3410), which is reproduced.
int hmme_slot_index_amp_off(int part_size, int depth, int part_idx, int abs_z_idx) {
  if (part_size < 0 || part_size > 2 || part_idx < 0 || part_idx > 1 || (part_size == 0 && part_idx)) return -1;
  const int full = hmme_slot_index(part_size, depth, part_idx, abs_z_idx);
  if (full < 0) return -1;
  static const int base593[3][4] = {{592, 584, 544, 384}, {588, 560, 448, 0}, {590, 568, 480, 128}};     // [part size][depth]
  static const int base425[3][4] = {{424, 416, 384, 256}, {420, 400, 320, 0}, {422, 408, 352, 128}};
  int in_family = full - base593[part_size][depth];
  if (depth == 0 && part_size != 0) in_family ^= 1;
  return base425[part_size][depth] + in_family;
}

int hmme_amp_off_slot(int index_amp_off) {   // which of the 593 slots holds table entry `index_amp_off` of the 425 layout; -1 out of range
  static int map[425];
  static bool built = false;
  if (!built) {   // idempotent fill: concurrent first calls write the same values
    for (int i = 0; i < 425; ++i) map[i] = -1;
    for (int depth = 0; depth < 4; ++depth) {
      const int n = 1 << depth, s4 = 16 >> depth;
      for (int cy = 0; cy < n; ++cy)
        for (int cx = 0; cx < n; ++cx) {
          int z = 0;   // raster (4x4 units) -> z-order
          for (int b = 0; b < 4; ++b) z |= (((cx * s4) >> b) & 1) << (2 * b) | (((cy * s4) >> b) & 1) << (2 * b + 1);
          for (int ps = 0; ps < 3; ++ps)
            for (int pi = 0; pi < (ps ? 2 : 1); ++pi) map[hmme_slot_index_amp_off(ps, depth, pi, z)] = hmme_slot_index(ps, depth, pi, z);
        }
    }
    built = true;
  }
  return index_amp_off >= 0 && index_amp_off < 425 ? map[index_amp_off] : -1;
}

int hmme_compact_amp_off(const int16_t* mv593, const uint32_t* sad593, int16_t* mv425, uint32_t* sad425) {
  if (!mv593 || !sad593 || !mv425 || !sad425) return HMME_ERR_ARG;
  for (int i = 0; i < 425; ++i) {
    const int s = hmme_amp_off_slot(i);
    mv425[2 * i] = mv593[2 * s]; mv425[2 * i + 1] = mv593[2 * s + 1];
    sad425[i] = sad593[s];
  }
  return HMME_OK;
}

int hmme_slot_rect(int slot, int* x, int* y, int* w, int* h) {
  if (!x || !y || !w || !h) return HMME_ERR_ARG;
  static const int part_sizes[7] = {0, 1, 2, 4, 5, 6, 7};
  for (int depth = 0; depth < 4; ++depth) {
    const int s = 64 >> depth, n = 1 << depth;
    for (int cy = 0; cy < n; ++cy)
      for (int cx = 0; cx < n; ++cx)
        for (int pi = 0; pi < 7; ++pi)
          for (int idx = 0; idx < 2; ++idx) {
            if (slot_of(part_sizes[pi], depth, idx, cx, cy) != slot) continue;
            int rx = 0, ry = 0, rw = s, rh = s;
            switch (part_sizes[pi]) {   // TComDataCU::getPartIndexAndSize
              case 1: rh = s / 2; ry = idx ? s / 2 : 0; break;
              case 2: rw = s / 2; rx = idx ? s / 2 : 0; break;
              case 4: rh = idx ? 3 * s / 4 : s / 4; ry = idx ? s / 4 : 0; break;
              case 5: rh = idx ? s / 4 : 3 * s / 4; ry = idx ? 3 * s / 4 : 0; break;
              case 6: rw = idx ? 3 * s / 4 : s / 4; rx = idx ? s / 4 : 0; break;
              case 7: rw = idx ? s / 4 : 3 * s / 4; rx = idx ? 3 * s / 4 : 0; break;
              default: break;
            }
            *x = cx * s + rx; *y = cy * s + ry; *w = rw; *h = rh;
            return HMME_OK;
          }
  }
  return HMME_ERR_ARG;
}

// ---- per-CTU drop-in ---------------------------------------------------------------------------------
namespace {
int build_frac_cover(hmme_ctx* ctx);
using frac_fn = void (*)(const RefSet, int, const RefSet, int, const MeJob*, const hmme::FracPrep, int, uint32_t*, const uint16_t*, const int16_t*, uint32_t, int, const hmme::FracWp, int16_t*, uint32_t*);
// One build per (sample width, distortion, weighting) at two waves per SIMD: the 8-bit kernels spill nothing (231 / 237 VGPRs), two
// workgroups per CU.  Round 4's three-wave build of the 8-bit kernel (168 VGPRs + 50 spilled dwords per lane: 104 MB of scratch writes
// per 2160p launch) is gone.
inline frac_fn frac_kernel(int wide, int had, int wp = 0) {
  static const frac_fn fns[2][2] = {{hmme::me_frac_kernel<0, 1, 0, 2>, hmme::me_frac_kernel<1, 1, 0, 2>}, {hmme::me_frac_kernel<0, 2, 0>, hmme::me_frac_kernel<1, 2, 0>}};
  static const frac_fn fns_wp[2] = {hmme::me_frac_kernel<0, 2, 1>, hmme::me_frac_kernel<1, 2, 1>};   // weighted calls always stage u16 samples
  if (wp) return fns_wp[had ? 1 : 0];
  return fns[wide ? 1 : 0][had ? 1 : 0];
}
// more than 64 KiB of dynamic LDS: opt in once per kernel and context
int frac_lds_optin(hmme_ctx* ctx, int wide, int had, int wp) {
  bool& done = ctx->frac_lds_optin[wp ? 2 : (wide ? 1 : 0)][had ? 1 : 0];
  if (done) return HMME_OK;
  const size_t bytes = hmme::frac_lds_bytes(wide ? 2 : 1);
  if (bytes > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute((const void*)frac_kernel(wide, had, wp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail(ctx, HMME_ERR_DEVICE, "hipFuncSetAttribute(me_frac_kernel, %zu bytes of LDS) -> %s", bytes, hipGetErrorString(e));
  }
  done = true;
  return HMME_OK;
}
const hmme::FracWp kNoWp = {0.f, 0.f, 0.f};
const hmme::FracPrep kNoPrep = {nullptr, 1u << 16, 0, 0};
// workgroups of a refinement launch: one per job, dealt from the end of the job table (me_frac_kernel).  HMME_FRAC_GRID=<n> launches n
// workgroups that take job after job from a counter instead, HMME_FRAC_GRID=-1 as many of those as the chip holds at a time (the
// runtime's occupancy figure for this kernel with its LDS block x the CUs): round 4's intermediate launch, kept for A/B runs -- once
// both orders ran last-first it was the slower one on every content (profiles/r04g_frac_grid_both_last_first.txt)
int frac_grid(hmme_ctx* ctx, int wide, int had, int jobs) {
  static const int forced = std::getenv("HMME_FRAC_GRID") ? std::atoi(std::getenv("HMME_FRAC_GRID")) : 0;
  if (forced == 0) return jobs;
  int grid = forced;
  if (forced < 0) {
    int& per_cu = ctx->frac_wg_per_cu[wide ? 1 : 0][had ? 1 : 0];
    if (per_cu == 0) {
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)frac_kernel(wide, had), hmme::frac_threads(wide ? 2 : 1),
                                                       hmme::frac_lds_bytes(wide ? 2 : 1)) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = 2;
      }
      per_cu = n;
      if (std::getenv("HMME_TRACE")) fprintf(stderr, "hmme: me_frac_kernel<%d, %d>: %d workgroups per CU, %d CUs\n", had ? 1 : 0, wide ? 2 : 1, n, ctx->num_cus);
    }
    grid = per_cu * ctx->num_cus;
  }
  return grid < jobs ? grid : jobs;
}

// host-side packing of the call block, one picture row at a time; separate reduction and narrowing loops so that the compiler
// vectorises both (the reference copies the same window sample by sample, TEncOpenCL.cpp:275-277)
inline void row_minmax(const int16_t* __restrict__ s, int n, int& lo, int& hi) {
  int16_t l = (int16_t)(lo < -32768 ? -32768 : lo), h = (int16_t)(hi > 32767 ? 32767 : hi);
  for (int x = 0; x < n; ++x) { l = s[x] < l ? s[x] : l; h = s[x] > h ? s[x] : h; }
  lo = l; hi = h;
}
inline void row_pack8(const int16_t* __restrict__ s, int n, uint8_t* __restrict__ d) {
  for (int x = 0; x < n; ++x) d[x] = (uint8_t)s[x];
}
inline void row_pack16(const int16_t* __restrict__ s, int n, int bias, uint16_t* __restrict__ d) {
  for (int x = 0; x < n; ++x) d[x] = (uint16_t)(s[x] + bias);
}

// One (CTU, reference) call: search (out_mv / out_sad), refinement of the winners or of the caller's integer MVs
// (refine_had >= 0: out_qmv / out_cost), or both.  With refinement the staged window carries a halo of kRefineHalo samples.
int ctu_call(hmme_ctx* ctx, const int16_t* ctu, int ctu_stride, const int16_t* ref0, int ref_stride, const hmme_search_params* p,
             bool do_search, const int16_t* int_mv, int refine_had, int16_t* out_mv, uint32_t* out_sad, int16_t* out_qmv, uint32_t* out_cost,
             const hmme_weight* wp = nullptr) {
  if (!ctx) return HMME_ERR_ARG;
  const bool refine = refine_had >= 0;
  if (wp && (wp->shift < 0 || wp->shift > 15)) return fail(ctx, HMME_ERR_ARG, "weighted prediction: shift %d outside 0..15", wp->shift);
  if (!ctu || !ref0 || !p || (do_search && (!out_mv || !out_sad)) || (refine && (!out_qmv || !out_cost)) || (!do_search && !int_mv))
    return fail(ctx, HMME_ERR_ARG, "hmme_search_ctu: null argument");
  const int halo = refine ? kRefineHalo : 0;
  if (p->bit_depth < 8 || p->bit_depth > 12) return fail(ctx, HMME_ERR_UNSUPPORTED, "bit depth %d outside 8..12", p->bit_depth);
  // MeJob carries the window and the predictor as int16 (what TComMv holds, TComMv.h:51-55): anything wider would address
  // the staged window at a truncated offset
  {
    const int v[6] = {p->lt_x, p->lt_y, p->rb_x, p->rb_y, p->pred_x, p->pred_y};
    for (int i = 0; i < 6; ++i)
      if (v[i] < -32768 || v[i] > 32767) return fail(ctx, HMME_ERR_ARG, "hmme_search_ctu: window / predictor component %d outside int16", v[i]);
  }
  const int maxv = (1 << p->bit_depth) - 1;
  // samples outside [0, maxv] are the bi-prediction origin 2*org - pred_other (reference TEncSearch.cpp:3702-3712,
  // TComYuv::removeHighFreq TComYuv.cpp:409-440, unclipped).  They stay exact: the 16-bit kernel runs on samples
  // biased by 2^bitDepth (|a - b| is unchanged), so pick the kernel after looking at the data.
  int lo = 32767, hi = -32768;
  for (int y = 0; y < 64; ++y) row_minmax(ctu + (long)y * ctu_stride, 64, lo, hi);
  const int wx = p->rb_x - p->lt_x + 1, wy = p->rb_y - p->lt_y + 1;
  const bool bipred_origin = lo < 0 || hi > maxv;
  if (lo < -maxv || hi > 2 * maxv)
    return fail(ctx, HMME_ERR_RANGE, "hmme_search_ctu: current-block sample outside [%d, %d] (bit depth %d)", -maxv, 2 * maxv, p->bit_depth);
  const bool wide = p->bit_depth > 8 || bipred_origin || wp;
  int bias = bipred_origin ? (1 << p->bit_depth) : 0;
  // xPatternSearchFracDIF on 2*org - pred_other (the bBi pass, TEncSearch.cpp:3798 with bBi): the refinement kernel runs on the same
  // biased u16 staging as the search; the bias passes HM's interpolation exactly and shifts its clip bounds (me_kernels.hpp, the evaluation of one work item)
  const int shift_bd = p->shift_free ? 8 : p->bit_depth;   // the kernels shift by (this - 8)
  const int sr_cap = ctx->sr_max;
  if (wx < 1 || wy < 1 || wx > 2 * sr_cap + 1 || wy > 2 * sr_cap + 1)
    return fail(ctx, HMME_ERR_ARG, "window %dx%d outside 1..%d", wx, wy, 2 * sr_cap + 1);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // pack CTU and window in pinned memory (the reference copies the same window with a scalar CPU loop,
  // TEncOpenCL.cpp:275-277); 1 byte per sample on the 8-bit path, 2 otherwise
  const int bps = wide ? 2 : 1;
  uint8_t* h_ctu = ctx->h_call + kCallCtu;
  uint8_t* h_win = ctx->h_call + kCallWin;
  const int rows = wy + 63 + 2 * halo, cols = wx + 63 + 2 * halo;
  const int16_t* src = ref0 + (long)(p->lt_y - halo) * ref_stride + (p->lt_x - halo);
  int vlo = 32767, vhi = -32768;
  if (wp)   // the window goes up as it is and is weighted on the device; block and weighted window share a bias chosen below
    for (int y = 0; y < rows; ++y) row_minmax(src + (long)y * ref_stride, cols, vlo, vhi);
  long wlo = 0, whi = 0;   // range of the weighted window
  if (wp) {
    if (vlo < 0 || vhi > maxv) return fail(ctx, HMME_ERR_RANGE, "hmme_search_ctu: reference sample outside [0,%d] for bit depth %d", maxv, p->bit_depth);
    const long a = (((long)wp->w0 * vlo + wp->round) >> wp->shift) + wp->offset, b = (((long)wp->w0 * vhi + wp->round) >> wp->shift) + wp->offset;
    wlo = std::min(a, b); whi = std::max(a, b);
    // HM keeps the weighted sample in a Pel (`const Pel pred`, TComRdCostWeightPrediction.cpp:79): beyond int16 its arithmetic wraps,
    // which nothing here reproduces -- such a call goes back to the caller
    if (wlo < -32768 || whi > 32767) return fail(ctx, HMME_ERR_UNSUPPORTED, "weighted prediction reaches %ld..%ld, beyond a Pel", wlo, whi);
    const long low = std::min<long>(wlo, lo);
    bias = low < 0 ? (int)-low : 0;
    if (std::max<long>(whi, hi) + bias > 65535) return fail(ctx, HMME_ERR_UNSUPPORTED, "weighted prediction: samples span more than 16 bits");
  }
  for (int y = 0; y < 64; ++y) {
    if (wide) row_pack16(ctu + (long)y * ctu_stride, 64, bias, (uint16_t*)h_ctu + y * 64);
    else row_pack8(ctu + (long)y * ctu_stride, 64, h_ctu + y * 64);
  }
  for (int y = 0; y < rows; ++y) {
    uint8_t* row = h_win + (size_t)y * kWinPitch;
    const int16_t* srow = src + (long)y * ref_stride;
    if (!wp) row_minmax(srow, cols, vlo, vhi);
    if (wide) row_pack16(srow, cols, wp ? 0 : bias, (uint16_t*)row);
    else row_pack8(srow, cols, row);
    std::memset(row + cols * bps, 0, 16);   // the kernels stage whole dwords past the last sample
  }
  if (vlo < 0 || vhi > maxv) return fail(ctx, HMME_ERR_RANGE, "hmme_search_ctu: reference sample outside [0,%d] for bit depth %d", maxv, p->bit_depth);
  if (wp) {   // the sums of a weighted search must fit the cost field like any other (the weighted window may be far from the block)
    const long span = std::max<long>((long)hi - wlo, whi - (long)lo);
    if (((4096 * span) >> (shift_bd - 8)) + 65535 >= (long)hmme::kInvCost16)
      return fail(ctx, HMME_ERR_UNSUPPORTED, "weighted SADs of this block could reach %ld: beyond the cost field", 4096 * span);
    // the refinement's Hadamard sums are exact in fp32 below 2^24: 64 coefficients of up to 64 * span each
    if (refine && 4096 * span >= (1L << 24))
      return fail(ctx, HMME_ERR_UNSUPPORTED, "weighted refinement: sample differences up to %ld exceed what the Hadamard sums hold exactly", span);
  }
  // unshifted sums must fit the 24-bit cost field of the 16-bit kernel's keys (me_kernels.hpp kInvCost16).  The bound is taken
  // from the samples of THIS call (both scans are made anyway): no |cur - ref| exceeds max(hi - vlo, vhi - lo), so any content whose
  // 64x64 sum cannot reach the marker is searched, whatever the nominal bit depth says (10-bit bi-prediction origins nominally
  // reach 4096 * 2046, real ones stay far below)
  if (p->shift_free && wide) {
    const long span = std::max<long>((long)hi - vlo, (long)vhi - lo);
    if (4096 * span + 65535 >= (long)hmme::kInvCost16)
      return fail(ctx, HMME_ERR_UNSUPPORTED, "shift-free SADs of this block could reach %ld (sample difference up to %ld at bit depth %d%s): beyond the cost field",
                  4096 * span, span, p->bit_depth, bipred_origin ? ", bi-prediction origin" : "");
  }
  MeJob job;
  job.ctu_x = 0; job.ctu_y = 0;
  job.lt_x = (int16_t)p->lt_x; job.lt_y = (int16_t)p->lt_y; job.rb_x = (int16_t)p->rb_x; job.rb_y = (int16_t)p->rb_y;
  job.pred_x = (int16_t)p->pred_x; job.pred_y = (int16_t)p->pred_y;
  hipStream_t s = ctx->stream;
  // one CTU alone would keep a few of the 256 CUs busy: its work is dealt to up to 64 workgroups (task ranges and window tiles on
  // the 8-bit path, strips of candidate rows on the 16-bit path) that merge through the 64-bit atomicMin table
  MeJob16* js = (MeJob16*)(ctx->h_call + kCallJobs);
  int n_wg = 0, pdw = 0, smax = 0;
  if (!do_search) {
    js[0].j = job; js[0].job = 0; js[0].y0 = js[0].y1 = 0;
    std::memcpy(ctx->h_call + kCallImv, int_mv, sizeof(int16_t) * 2 * HMME_NUM_CTU_PARTS);
  } else if (!wide) {
    // windows beyond 129 x 129 candidates: up to 2 x 2 tiles (tile (0,0) first: finalize decodes against its top-left)
    const int tiles_x = (wx + hmme::kTileStep - 1) / hmme::kTileStep, tiles_y = (wy + hmme::kTileStep - 1) / hmme::kTileStep;
    const int per_tile = kCallMaxJobs / (tiles_x * tiles_y);
    for (int ty = 0; ty < tiles_y; ++ty)
      for (int tx = 0; tx < tiles_x; ++tx) {
        MeJob sub = job;
        sub.lt_x = (int16_t)(job.lt_x + tx * hmme::kTileStep); sub.lt_y = (int16_t)(job.lt_y + ty * hmme::kTileStep);
        sub.rb_x = (int16_t)std::min<int>(job.rb_x, sub.lt_x + hmme::kTileStep - 1);
        sub.rb_y = (int16_t)std::min<int>(job.rb_y, sub.lt_y + hmme::kTileStep - 1);
        const int nt = hmme::me_num_tasks(sub.rb_x - sub.lt_x + 1, sub.rb_y - sub.lt_y + 1);
        const int parts = std::max(1, std::min(per_tile, (nt + 3) / 4));   // 4 tasks = one per wave
        for (int i = 0; i < parts; ++i, ++n_wg) {
          js[n_wg].j = sub; js[n_wg].job = 0 | tx << 30 | ty << 29;
          js[n_wg].y0 = (int16_t)((long)nt * i / parts); js[n_wg].y1 = (int16_t)((long)nt * (i + 1) / parts);
        }
      }
  } else {
    pdw = pick_pdw16(wx);
    // at least as many strips as the LDS needs, and enough of them to spread the CTU over the chip: a strip is
    // >= 4 candidate rows (each also stages the 63 rows below it)
    n_wg = std::max(strips_for(pdw, wy), std::min(kCallMaxJobs, (wy + 3) / 4));
    if (n_wg > kCallMaxJobs) return fail(ctx, HMME_ERR_UNSUPPORTED, "window needs %d strips", n_wg);
    for (int i = 0; i < n_wg; ++i) {
      js[i].j = job; js[i].job = 0;
      js[i].y0 = (int16_t)((long)wy * i / n_wg); js[i].y1 = (int16_t)((long)wy * (i + 1) / n_wg);
      if (js[i].y1 - js[i].y0 > smax) smax = js[i].y1 - js[i].y0;
    }
  }
  std::memcpy(ctx->h_call + kCallFracJob, &job, sizeof job);
  {
    const int n16 = (int)((kCallWin + (size_t)rows * kWinPitch + 64 + 15) / 16);
    hipLaunchKernelGGL(hmme::me_stage_call_kernel, dim3((n16 + 255) / 256), dim3(256), 0, s, (const uint4*)ctx->h_call_dev, (uint4*)ctx->d_call, n16);
    HIP_TRY(ctx, hipGetLastError());
    if (wp && do_search) {   // the weighted window the integer search runs on; the raw one stays where it is for the refinement's interpolation
      if (!ctx->d_wwin) HIP_TRY(ctx, hipMalloc(&ctx->d_wwin, (size_t)kWinRows * kWinPitch + 64));
      hipLaunchKernelGGL(hmme::me_weight_window_kernel, dim3((rows * cols + 255) / 256), dim3(256), 0, s, (const uint8_t*)(ctx->d_call + kCallWin), ctx->d_wwin,
                         (int)kWinPitch, rows, cols, wp->w0, wp->round, wp->shift, wp->offset + bias);
      HIP_TRY(ctx, hipGetLastError());
    }
  }
  // the kernel addresses ref(ctu + lt): bias the base so that (lt_x, lt_y) lands on the window copy's first sample
  const uint8_t* ref_base = ctx->d_call + kCallWin - (long)(p->lt_y - halo) * kWinPitch - (long)(p->lt_x - halo) * bps;
  const uint8_t* ref_base_search = wp ? ctx->d_wwin - (long)(p->lt_y - halo) * kWinPitch - (long)(p->lt_x - halo) * bps : ref_base;
  const MeJob16* d_js = (const MeJob16*)(ctx->d_call + kCallJobs);
  const int* d_first = (const int*)(ctx->d_call + kCallFirst);
  unsigned long long* d_best1 = (unsigned long long*)(ctx->d_call + kCallBest);
  // the 4.7 KB of results go straight into pinned host memory (mapped into the device's address space): no download step
  int16_t* d_mv1 = (int16_t*)(ctx->d_res + kResMv);
  uint32_t* d_sad1 = (uint32_t*)(ctx->d_res + kResSad);
  int rc = HMME_OK;
  const uint32_t seq = ++ctx->call_seq ? ctx->call_seq : ++ctx->call_seq;   // never 0, the words' initial value
  if (do_search) {
  if (!wide)
    rc = launch_search8_split(ctx, one_ref(ctx->d_call + kCallCtu), 1, one_ref(ref_base), kWinPitch, d_js, d_first, 1, n_wg, p->fen, d_mv1, d_sad1, s, d_best1, false);
  else
    rc = launch_search16(ctx, one_ref(ctx->d_call + kCallCtu), 1, one_ref(ref_base_search), kWinPitch, d_js, d_first, 1, n_wg, pdw, smax, wp ? 0 : p->fen, shift_bd,   // xGetSADw reads every row
                         d_mv1, d_sad1, s, d_best1, false);
  if (rc) return rc;
  hipLaunchKernelGGL(hmme::me_finalize1_kernel, dim3(1), dim3(640), 0, s, d_best1, d_js, ctx->lambda_q16, d_mv1, d_sad1,
                     (volatile uint32_t*)(ctx->d_res + kResDone), seq);
  HIP_TRY(ctx, hipGetLastError());
  }
  if (refine) {
    // xPatternSearchFracDIF for the 593 slots on the block and window that are staged anyway: one more workgroup-sized kernel,
    // its input the integer tables the finalize kernel just wrote (or the caller's), its output in the same pinned block
    rc = build_frac_cover(ctx);
    if (rc) return rc;
    const int16_t* d_imv = do_search ? d_mv1 : (const int16_t*)(ctx->d_call + kCallImv);
    // weighted: the interpolated prediction is weighted sample by sample (me_frac_eval0 / me_frac_eval1, FracWp); the current samples carry `bias`, the raw window none
    const hmme::FracWp fw = wp ? hmme::FracWp{std::ldexp((float)wp->w0, -wp->shift), std::ldexp((float)wp->round, -wp->shift), (float)(bias + wp->offset)} : kNoWp;
    rc = frac_lds_optin(ctx, wide ? 1 : 0, refine_had ? 1 : 0, wp ? 1 : 0);
    if (rc) return rc;
    hipLaunchKernelGGL(frac_kernel(wide ? 1 : 0, refine_had ? 1 : 0, wp ? 1 : 0), dim3(1), dim3(hmme::frac_threads(bps)), hmme::frac_lds_bytes(bps), s, one_ref(ctx->d_call + kCallCtu),
                       64 * bps, one_ref(ref_base), kWinPitch, (const MeJob*)(ctx->d_call + kCallFracJob), kNoPrep, 1, (uint32_t*)nullptr, ctx->d_frac_cover, d_imv, ctx->lambda_q16,
                       p->bit_depth | ((bipred_origin && !wp) ? 0x100 : 0), fw, (int16_t*)(ctx->d_res + kResQmv), (uint32_t*)(ctx->d_res + kResCost));
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(hmme::me_publish_kernel, dim3(1), dim3(1), 0, s, (volatile uint32_t*)(ctx->d_res + kResDone2), seq);
    HIP_TRY(ctx, hipGetLastError());
  }
  volatile uint32_t* done = (volatile uint32_t*)(ctx->h_res + (refine ? kResDone2 : kResDone));
  // the caller blocks on this call anyway (TEncSearch.cpp:3749-3758): poll the completion word for a while instead of paying the
  // interrupt wake-up of hipStreamSynchronize, then fall back to it (a faulted kernel never publishes)
  for (int spin = 0; *done != seq && spin < 200000; ++spin) __builtin_ia32_pause();
  if (*done != seq) HIP_TRY(ctx, hipStreamSynchronize(s));
  if (*done != seq) return fail(ctx, HMME_ERR_DEVICE, "hmme_search_ctu: the device did not publish results");
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  if (do_search) {
    std::memcpy(out_mv, ctx->h_res + kResMv, sizeof(int16_t) * 2 * HMME_NUM_CTU_PARTS);
    std::memcpy(out_sad, ctx->h_res + kResSad, sizeof(uint32_t) * HMME_NUM_CTU_PARTS);
  }
  if (refine) {
    std::memcpy(out_qmv, ctx->h_res + kResQmv, sizeof(int16_t) * 2 * HMME_NUM_CTU_PARTS);
    std::memcpy(out_cost, ctx->h_res + kResCost, sizeof(uint32_t) * HMME_NUM_CTU_PARTS);
  }
  return HMME_OK;
}
}  // namespace

int hmme_search_ctu(hmme_ctx* ctx, const int16_t* ctu, int ctu_stride, const int16_t* ref0, int ref_stride,
                    const hmme_search_params* p, int16_t* out_mv, uint32_t* out_sad) {
  return ctu_call(ctx, ctu, ctu_stride, ref0, ref_stride, p, true, nullptr, -1, out_mv, out_sad, nullptr, nullptr);
}

int hmme_search_ctu_w(hmme_ctx* ctx, const int16_t* ctu, int ctu_stride, const int16_t* ref0, int ref_stride,
                      const hmme_search_params* p, const hmme_weight* wp, int16_t* out_mv, uint32_t* out_sad) {
  if (ctx && !wp) return fail(ctx, HMME_ERR_ARG, "hmme_search_ctu_w: null weight");
  return ctu_call(ctx, ctu, ctu_stride, ref0, ref_stride, p, true, nullptr, -1, out_mv, out_sad, nullptr, nullptr, wp);
}

int hmme_search_refine_ctu_w(hmme_ctx* ctx, const int16_t* ctu, int ctu_stride, const int16_t* ref0, int ref_stride, const hmme_search_params* p,
                             const hmme_weight* wp, int use_hadamard, int16_t* out_mv, uint32_t* out_sad, int16_t* out_qmv, uint32_t* out_cost) {
  if (ctx && !wp) return fail(ctx, HMME_ERR_ARG, "hmme_search_refine_ctu_w: null weight");
  return ctu_call(ctx, ctu, ctu_stride, ref0, ref_stride, p, true, nullptr, use_hadamard ? 1 : 0, out_mv, out_sad, out_qmv, out_cost, wp);
}

int hmme_search_refine_ctu(hmme_ctx* ctx, const int16_t* ctu, int ctu_stride, const int16_t* ref0, int ref_stride, const hmme_search_params* p,
                           int use_hadamard, int16_t* out_mv, uint32_t* out_sad, int16_t* out_qmv, uint32_t* out_cost) {
  return ctu_call(ctx, ctu, ctu_stride, ref0, ref_stride, p, true, nullptr, use_hadamard ? 1 : 0, out_mv, out_sad, out_qmv, out_cost);
}

int hmme_refine_ctu(hmme_ctx* ctx, const int16_t* ctu, int ctu_stride, const int16_t* ref0, int ref_stride, const hmme_search_params* p,
                    const int16_t* int_mv, int use_hadamard, int16_t* out_qmv, uint32_t* out_cost) {
  return ctu_call(ctx, ctu, ctu_stride, ref0, ref_stride, p, false, int_mv, use_hadamard ? 1 : 0, nullptr, nullptr, out_qmv, out_cost);
}

int hmme_refine_ctu_w(hmme_ctx* ctx, const int16_t* ctu, int ctu_stride, const int16_t* ref0, int ref_stride, const hmme_search_params* p,
                      const hmme_weight* wp, const int16_t* int_mv, int use_hadamard, int16_t* out_qmv, uint32_t* out_cost) {
  if (ctx && !wp) return fail(ctx, HMME_ERR_ARG, "hmme_refine_ctu_w: null weight");
  return ctu_call(ctx, ctu, ctu_stride, ref0, ref_stride, p, false, int_mv, use_hadamard ? 1 : 0, nullptr, nullptr, out_qmv, out_cost, wp);
}

// ---- planes ----------------------------------------------------------------------------------------------
int hmme_plane_create_ex(hmme_ctx* ctx, int width, int height, int bit_depth, hmme_plane** out) {
  if (!ctx) return HMME_ERR_ARG;
  if (!out || width < 8 || height < 8 || width > 16384 || height > 16384) return fail(ctx, HMME_ERR_ARG, "hmme_plane_create(%d, %d)", width, height);
  if (bit_depth < 8 || bit_depth > 12) return fail(ctx, HMME_ERR_UNSUPPORTED, "bit depth %d outside 8..12", bit_depth);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hmme_plane* pl = new hmme_plane;
  pl->ctx = ctx; pl->width = width; pl->height = height;
  pl->bit_depth = bit_depth; pl->bps = bit_depth > 8 ? 2 : 1;
  pl->pitch = ((width + 2 * kMarginX) * pl->bps + 255) & ~255;
  pl->rows = height + 2 * kMarginY;
  pl->ctus_x = (width + 63) / 64; pl->n_ctu = hmme_num_ctus(width, height);
  hipError_t e = hipMalloc(&pl->d_data, (size_t)pl->pitch * (pl->rows + 1));
  if (e == hipSuccess && (e = hipMalloc(&pl->d_blocks, (size_t)pl->n_ctu * hmme::kBlkBytes8 * pl->bps)) != hipSuccess) hipFree(pl->d_data);
  if (e == hipSuccess && (e = hipEventCreateWithFlags(&pl->filled, hipEventDisableTiming)) != hipSuccess) { hipFree(pl->d_data); hipFree(pl->d_blocks); }
  if (e != hipSuccess) { delete pl; return fail(ctx, HMME_ERR_NOMEM, "plane allocation: %s", hipGetErrorString(e)); }
  *out = pl;
  return HMME_OK;
}
int hmme_plane_create(hmme_ctx* ctx, int width, int height, hmme_plane** out) { return hmme_plane_create_ex(ctx, width, height, 8, out); }

void hmme_plane_destroy(hmme_plane* pl) {
  if (!pl) return;
  hipSetDevice(pl->ctx->device);
  if (pl->fill_pending) hipEventSynchronize(pl->filled);   // a fill or a search still in flight must not outlive the buffer
  if (pl->read_pending) hipEventSynchronize(pl->read_done);
  if (pl->filled) hipEventDestroy(pl->filled);
  hipFree(pl->d_data);
  hipFree(pl->d_blocks);
  hipFree(pl->d_stage);
  delete pl;
}

int hmme_plane_width(const hmme_plane* pl) { return pl ? pl->width : 0; }
int hmme_plane_height(const hmme_plane* pl) { return pl ? pl->height : 0; }
int hmme_plane_bit_depth(const hmme_plane* pl) { return pl ? pl->bit_depth : 0; }

int hmme_plane_upload_pel(hmme_plane* pl, const int16_t* origin, int stride) { return plane_upload<int16_t>(pl, origin, stride, nullptr, true); }
int hmme_plane_upload_u8(hmme_plane* pl, const uint8_t* origin, int stride) { return plane_upload<uint8_t>(pl, origin, stride, nullptr, true); }

int hmme_plane_upload_async(hmme_plane* pl, const void* origin, int stride, int sample_bytes, void* stream) {
  if (!pl) return HMME_ERR_ARG;
  if (sample_bytes == 1) return plane_upload<uint8_t>(pl, (const uint8_t*)origin, stride, (hipStream_t)stream, false);
  // 16-bit samples (HM's Pel, or the unsigned words of a 16-bit YUV file: anything beyond the bit depth reads as negative or too large)
  if (sample_bytes == 2) return plane_upload<int16_t>(pl, (const int16_t*)origin, stride, (hipStream_t)stream, false);
  return fail(pl->ctx, HMME_ERR_ARG, "hmme_plane_upload_async: sample_bytes %d (1 or 2)", sample_bytes);
}

int hmme_upload_status(hmme_ctx* ctx, void* stream) {
  if (!ctx) return HMME_ERR_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int flag = 0;
  const int rc = take_flag(ctx, 1, (hipStream_t)stream, &flag);
  if (rc) return rc;
  if (!flag) return HMME_OK;
  return fail(ctx, HMME_ERR_RANGE, "an asynchronous plane upload carried a sample outside the range of its plane's bit depth");
}

uint64_t hmme_test_device_address(const hmme_ctx* ctx, const hmme_plane* pl) {
  if (pl) return (uint64_t)(uintptr_t)pl->origin();
  return ctx ? (uint64_t)(uintptr_t)(ctx->d_call + kCallCtu) : 0;
}

#ifdef ME_SEARCH_T_TIMELINE   // timing-only builds (tools/search16_timeline.py): the per-workgroup stamps me_search16_kernel left
int hmme_test_timeline16(void* out, size_t bytes) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(hmme::g_timeline16), bytes < sizeof(hmme::g_timeline16) ? bytes : sizeof(hmme::g_timeline16)) == hipSuccess ? HMME_OK : HMME_ERR_DEVICE;
}
#endif
static int plan_tail8(int tail, int slots, int w, int tail_knob);
static bool tail_one_launch(int head, int n_tail, int launches_knob);
int hmme_test_tail_plan(int width, int height, int search_range, int n_pairs, int slots, int* out) {
  if (!out || width < 8 || height < 8 || width > 16384 || height > 16384 || search_range < 1 || search_range > 64 || n_pairs < 1 || slots < 1) return HMME_ERR_ARG;
  const int jobs = hmme_num_ctus(width, height) * n_pairs, tail = jobs % slots, w = 2 * search_range + 1;
  const int wgs = plan_tail8(tail, slots, w, 0);
  const int head = wgs > 0 ? jobs - tail : jobs;
  out[0] = jobs; out[1] = head; out[2] = wgs; out[3] = (wgs > 0 && tail_one_launch(head, jobs - head, 0)) ? 1 : 0;
  return HMME_OK;
}

int hmme_test_frac_deal(int k, int n_pairs, int width, int height) {
  if (n_pairs < 1 || width < 8 || height < 8 || width > 16384 || height > 16384) return -1;
  const int n_ctu = hmme_num_ctus(width, height);
  if (n_ctu > 0xffff || k < 0 || k >= n_pairs * n_ctu) return -1;
  const hmme::FracPrep prep = {nullptr, (uint32_t)n_ctu << 16, (uint32_t)width | (uint32_t)height << 16, 0};
  return hmme::me_frac_deal(k, n_pairs * n_ctu, prep);
}

int hmme_abi_version(void) { return HMME_ABI_VERSION; }
#ifndef HMME_BUILD_ID
#define HMME_BUILD_ID "unknown"
#endif
const char* hmme_build_id(void) { return HMME_BUILD_ID; }

int hmme_host_register(hmme_ctx* ctx, void* buffer, size_t bytes) {
  if (!ctx) return HMME_ERR_ARG;
  if (!buffer || !bytes) return fail(ctx, HMME_ERR_ARG, "hmme_host_register: null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipHostRegister(buffer, bytes, hipHostRegisterDefault));
  return HMME_OK;
}

int hmme_host_unregister(hmme_ctx* ctx, void* buffer) {
  if (!ctx) return HMME_ERR_ARG;
  if (!buffer) return fail(ctx, HMME_ERR_ARG, "hmme_host_unregister: null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipHostUnregister(buffer));
  return HMME_OK;
}

int hmme_plane_set_device_u8(hmme_plane* pl, const void* d_src, int src_pitch, void* stream) {
  if (!pl) return HMME_ERR_ARG;
  if (!d_src || src_pitch < pl->width) return fail(pl->ctx, HMME_ERR_ARG, "hmme_plane_set_device_u8: bad source");
  HIP_TRY(pl->ctx, hipSetDevice(pl->ctx->device));
  if (pl->bps == 1) return plane_fill<uint8_t, uint8_t>(pl, (const uint8_t*)d_src, src_pitch, (hipStream_t)stream, false);
  return plane_fill<uint8_t, uint16_t>(pl, (const uint8_t*)d_src, src_pitch, (hipStream_t)stream, false);
}

// ---- frame search --------------------------------------------------------------------------------------------
// How the (CTU, reference) searches of one launch are dealt to workgroups.  The chip holds ctx->wg_slots search workgroups at a time
// and a launch runs in rounds of that many: 2 040 CTU searches are 3.98 rounds of 512, but 570 (1920 x 1200) are 1.11 -- the second
// round would keep 58 slots busy and cost as much as the first (measured before this plan existed: 2 340 GSAD/s against 3 339 at
// 2160p).  So the jobs beyond the last full round ("tail", from job tail_first on) are dealt in finer pieces:
//   8-bit:  a launch without a tail runs whole jobs (me_search_kernel<FEN, 0>).  With one, ONE launch of me_search_kernel<FEN, 2>: jobs
//           [0, tail_first) whole, one workgroup each and first in the grid; the tail's tasks -- all its jobs' lane-iterations as ONE list --
//           cut into tail_wgs equal segments, one per workgroup (a segment may end one job and begin the next); everything merged
//           through the 64-bit atomicMin table and decoded by me_finalize16_kernel.  Until round 6 every tail job was cut into the same
//           number of pieces: 240 jobs (720p) in 2 pieces each filled 480 of 512 slots with 34 or 33 lane-iterations -- 9 for the
//           slowest wave where 7.85 would do -- and the clipped windows of edge CTUs made shorter pieces still;
//   16-bit: one launch; jobs [0, tail_first) are cut into n_strips strips, tail jobs into tail_parts (>= n_strips).
// A launch of fewer jobs than slots is all tail (the small-picture split mode of round 1).  8-bit windows beyond 129 x 129 (four tiles
// per job through the split kernel) are not re-planned.
struct FramePlan {
  int jobs = 0;
  int n_strips = 1;        // 16-bit: strips of a head job; 8-bit: 4 when tiled, else 1
  int pdw = 0, strip_rows = 0;
  int tail_first = 0;      // == jobs: no tail
  int tail_parts = 1;      // 16-bit: strips of a tail job
  int tail_wgs = 0;        // 8-bit: workgroups (= segments) of the tail
  bool one_launch = true;  // 8-bit with a tail: head and tail in one segment launch (else the head whole, then the tail's segments: HMME_TAIL_LAUNCHES=2)
  bool tile8 = false;
  int n_wg16 = 0;          // 16-bit: workgroups of the launch
  size_t tail_jobs_off = 0;   // (8-bit segment launches: the table, me_seg_table_*, starts at ctx->d_jobs)
};

// pieces per tail job that finish `tail` jobs soonest: rounds of `slots` workgroups, each as long as its piece of a whole job
// (cost[k], in any unit) -- ties go to fewer pieces
static int plan_tail(int tail, int slots, int k_min, int k_max, const std::function<double(int)>& cost) {
  int best_k = k_min;
  double best = 1e30;
  for (int k = k_min; k <= k_max; ++k) {
    const double t = (double)(((long)tail * k + slots - 1) / slots) * cost(k);
    if (t < best * 0.999) { best = t; best_k = k; }
  }
  return best_k;
}

// 8-bit launches: workgroups (= segments) for `tail` jobs of full w x w windows beyond the last full round of `slots` workgroups, or 0: the tail
// runs whole in the head's launch.  (pure arithmetic: hmme_test_tail_plan exposes it to the CPU tests)
// the tail's units in `rounds` rounds of `slots` equal segments.  A round costs the lane-iterations of its slowest wave plus what
// every workgroup pays per job it touches -- window load, flush, merge: a quarter iteration fits the sweeps of
// profiles/archive/r02Q_tail_sweeps.txt -- and a segment of u units touches 1 + (u - 1) / units jobs on average.  Full windows are
// assumed (the device counts the clipped ones' units itself: me_prep_segments_kernel).  Never below one unit per workgroup: tiny
// windows get fewer workgroups than slots.
// Segments are whole units of kSegUnit = 4 tasks (one per wave), so a segment of u units costs u lane-iterations.  Against that stands
// the tail run WHOLE in the head's launch -- ceil(nt / 4) lane-iterations per wave for one more round of workgroups, which flow in
// behind the head's without a launch boundary, with the large tasks of whole jobs and without merge table, preset and decode: the
// segment launch is taken only where the model says it wins by 8 % plus half a lane-iteration (1080p's 510 jobs and the 504 left over
// at 2160p stay whole: measured 10 % and 4 % slower as segments, profiles/r06d_tail_segments.txt).
static int plan_tail8(int tail, int slots, int w, int tail_knob) {
  if (tail <= 0 || tail > hmme::kSegPrepThreads) return 0;
  const int nt = hmme::me_num_tasks(w, w), units = (nt + hmme::kSegUnit - 1) / hmme::kSegUnit;
  const long total = (long)tail * units;
  int best_wgs = 0;
  double best = (double)units + 0.25;   // the tail as whole jobs
  for (int rounds = 1; rounds <= 4; ++rounds) {
    const long wgs = std::max<long>(1, std::min<long>((long)rounds * slots, total));
    const double per_wg = (double)total / wgs;
    const double t = 1.08 * rounds * (std::ceil(per_wg) + 0.25 * (1.0 + (per_wg - 1.0) / units)) + 0.5;
    if (t < best * 0.999) { best = t; best_wgs = (int)wgs; }
    if (wgs < (long)rounds * slots) break;
  }
  if (tail_knob > 1) best_wgs = (int)std::max<long>(1, std::min<long>((long)tail * tail_knob, total));   // A/B: tail_knob workgroups per tail job
  return best_wgs;
}
// head and tail of an 8-bit launch as one segment launch?  (see prep_jobs)
static bool tail_one_launch(int head, int n_tail, int launches_knob) { return launches_knob == 1 || (launches_knob != 2 && 3 * n_tail >= head); }

// builds the device job table of a picture search against n_refs reference pictures on `s`; job index =
// ref * count + ctu.  8-bit: MeJob[head jobs] (+ MeJob16[tail jobs * parts]); 8-bit tiled / 16-bit: MeJob16[workgroups]
static int prep_jobs(hmme_ctx* ctx, const hmme_plane* cur, const hmme_frame_params* fp, const void* d_pred_q, int first, int count,
                     int n_refs, hipStream_t s, FramePlan* pl) {
  const bool wide = fp->bit_depth > 8;
  const int jobs = count * n_refs, slots = ctx->wg_slots, w = 2 * fp->search_range + 1;
  pl->jobs = jobs;
  pl->pdw = pick_pdw16(w);
  pl->n_strips = 1;
  pl->strip_rows = w;
  pl->tile8 = !wide && fp->search_range > 64;   // window beyond 129 x 129: four tile searches per CTU, merged like split tasks
  pl->tail_first = jobs; pl->tail_parts = 1;
  static const int tail_knob = std::getenv("HMME_TAIL_PARTS") ? std::atoi(std::getenv("HMME_TAIL_PARTS")) : 0;   // A/B knob: 1 = no tail plan
  const int tail = jobs % slots;
  if (wide) {   // strips of equal height for the full window (me_strip_rows16); clipped windows choose their own within n_strips
    const int rmax = rows_max16(pl->pdw), n_min = strips_for(pl->pdw, w);
    pl->strip_rows = rmax;
    const int h = hmme::me_strip_rows16(w, w, rmax, n_min + 4);   // up to four strips more than LDS alone needs
    pl->n_strips = (w + h - 1) / h;
    if (const char* e = std::getenv("HMME_STRIPS16")) {   // A/B knob: that many strips of equal height
      const int n = std::atoi(e);
      if (n >= n_min && n <= w) { pl->n_strips = n; pl->strip_rows = (w + n - 1) / n; }
    }
    // the workgroups beyond the last full round: head jobs give (jobs - tail) * n_strips of them, which need not fill whole rounds
    // either -- the tail is planned on what is left of the last head round
    const int lanes = (((w + 1) >> 1) + 2) / 3;
    auto strip_cost = [&](int k) {   // rounds of one strip of a job cut into k, plus half a round for its two window loads
      const int hk = (w + k - 1) / k;
      return 2.0 * ((((hk * lanes + 63) >> 6) + 3) >> 2) + 1.0;
    };
    if (tail && tail_knob != 1) {
      const int k_max = std::min(w / 4, 64);
      int k = tail_knob > 1 ? std::min(tail_knob, k_max) : plan_tail(tail, slots, pl->n_strips, k_max, strip_cost);
      if (k < pl->n_strips) k = pl->n_strips;
      if (k > pl->n_strips) { pl->tail_first = jobs - tail; pl->tail_parts = k; }
    }
    pl->n_wg16 = pl->tail_first * pl->n_strips + (jobs - pl->tail_first) * pl->tail_parts;
  } else if (pl->tile8) {
    pl->n_strips = 4;
  } else if (tail && tail_knob != 1) {
    const int wgs = plan_tail8(tail, slots, w, tail_knob);
    if (wgs > 0) { pl->tail_first = jobs - tail; pl->tail_wgs = wgs; }
  }
  const int head = pl->tail_first, n_tail = jobs - head;
  size_t need;
  if (wide) need = sizeof(MeJob16) * (size_t)pl->n_wg16;
  else if (pl->tile8) need = sizeof(MeJob16) * (size_t)jobs * 4;
  else {
    // One launch lets the tail's segments start on the slots the head's short jobs (clipped windows at the picture's edge) leave first, but sends
    // the head's jobs through the merge table and its decode as well: worth it where the tail is a good part of the launch (1440p, 408 tail
    // jobs behind 512: +0.4 %; 2560 x 1088, 168: +0.5 %), not for a few jobs behind a full round (1200p, 58: -3 %).  profiles/r06n_tail_one_or_two_launches.txt
    static const int launches_knob = std::getenv("HMME_TAIL_LAUNCHES") ? std::atoi(std::getenv("HMME_TAIL_LAUNCHES")) : 0;   // A/B knob: 1 | 2
    pl->one_launch = tail_one_launch(head, n_tail, launches_knob);
    pl->tail_jobs_off = pl->one_launch ? 0 : (sizeof(MeJob) * (size_t)head + 255) & ~(size_t)255;
    need = !n_tail ? sizeof(MeJob) * (size_t)head
         : pl->one_launch ? hmme::me_seg_table_bytes(head + pl->tail_wgs, jobs) : pl->tail_jobs_off + hmme::me_seg_table_bytes(pl->tail_wgs, n_tail);
  }
  size_t cap = ctx->jobs_bytes;
  const size_t per_ref = n_refs > 0 ? (need + n_refs - 1) / n_refs : need;   // what kMaxRefs pairs of this picture size would ask for
  int rc = ensure(ctx, (uint8_t**)&ctx->d_jobs, &cap, need, per_ref * hmme::kMaxRefs + 4096);
  if (cap != ctx->jobs_bytes) ctx->jobs_tag.valid = false;   // reallocated: whatever the table was, it is gone
  ctx->jobs_bytes = cap;
  if (rc) return rc;
  if (wide || pl->tile8 || n_tail) {
    size_t fcap = (size_t)ctx->first_strip_cap * sizeof(int);
    const size_t fcap0 = fcap;
    rc = ensure(ctx, &ctx->d_first_strip, &fcap, sizeof(int) * (size_t)jobs, sizeof(int) * (size_t)(jobs / (n_refs > 0 ? n_refs : 1) + 1) * hmme::kMaxRefs);
    if (fcap != fcap0) ctx->jobs_tag.valid = false;
    ctx->first_strip_cap = (int)(fcap / sizeof(int));
    if (rc) return rc;
  }
  static const bool no_cache = std::getenv("HMME_NO_TABLE_CACHE") != nullptr;
  hmme_ctx::TableTag tag;
  tag.valid = !d_pred_q && !no_cache;
  tag.w = cur->width; tag.h = cur->height; tag.bit_depth = fp->bit_depth; tag.sr = fp->search_range; tag.first = first; tag.count = count; tag.pairs = n_refs;
  tag.buf = ctx->d_jobs; tag.buf2 = (wide || pl->tile8 || n_tail) ? ctx->d_first_strip : nullptr; tag.stream = (void*)s;
  if (tag.same(ctx->jobs_tag)) return HMME_OK;   // the table of the launch before is this launch's table
  ctx->jobs_tag.valid = false;
  const dim3 block(256);
  auto grid = [](int n) { return dim3((n + 255) / 256); };
  if (pl->tile8)
    hipLaunchKernelGGL(hmme::me_prep_jobs_tile_kernel, grid(jobs), block, 0, s, (MeJob16*)ctx->d_jobs, ctx->d_first_strip,
                       (const int16_t*)d_pred_q, first, count, n_refs, cur->width, cur->height, fp->search_range);
  else if (wide)
    hipLaunchKernelGGL(hmme::me_prep_jobs16_kernel, grid(jobs), block, 0, s, (MeJob16*)ctx->d_jobs, ctx->d_first_strip,
                       (const int16_t*)d_pred_q, first, count, n_refs, cur->width, cur->height, fp->search_range, pl->n_strips, pl->strip_rows,
                       pl->tail_first, pl->tail_parts);
  else {
    if (!n_tail || !pl->one_launch) {
      if (head)
        hipLaunchKernelGGL(hmme::me_prep_jobs_kernel, grid(head), block, 0, s, (MeJob*)ctx->d_jobs, (const int16_t*)d_pred_q, first, count,
                           n_refs, cur->width, cur->height, fp->search_range, 0, head, 1, (uint32_t*)nullptr);
      if (n_tail)
        hipLaunchKernelGGL(hmme::me_prep_segments_kernel, dim3(1), dim3(hmme::kSegPrepThreads), 0, s, (void*)((uint8_t*)ctx->d_jobs + pl->tail_jobs_off), ctx->d_first_strip,
                           (const int16_t*)d_pred_q, first, count, n_refs, cur->width, cur->height, fp->search_range, pl->tail_wgs, head, n_tail, 0);
    } else {   // one segment table for the whole launch: the head's jobs whole, then the tail's segments
      if (head)
        hipLaunchKernelGGL(hmme::me_prep_whole_segments_kernel, grid(head), block, 0, s, ctx->d_jobs, ctx->d_first_strip, (const int16_t*)d_pred_q, first, count,
                           n_refs, cur->width, cur->height, fp->search_range, head + pl->tail_wgs, head);
      hipLaunchKernelGGL(hmme::me_prep_segments_kernel, dim3(1), dim3(hmme::kSegPrepThreads), 0, s, ctx->d_jobs, ctx->d_first_strip,
                         (const int16_t*)d_pred_q, first, count, n_refs, cur->width, cur->height, fp->search_range, head + pl->tail_wgs, head, n_tail, head);
    }
  }
  HIP_TRY(ctx, hipGetLastError());
  ctx->jobs_tag = tag;
  return HMME_OK;
}

// curs: the CTU-blocked copies of the current pictures (hmme_plane::d_blocks), cur_ctus_x their blocks per block row
static int run_search(hmme_ctx* ctx, const RefSet& curs, int cur_ctus_x, const RefSet& refs, int ref_pitch, const hmme_frame_params* fp, const FramePlan& pl,
                      int16_t* d_mv, uint32_t* d_sad, hipStream_t s) {
  if (fp->bit_depth > 8)
    return launch_search16(ctx, curs, cur_ctus_x, refs, ref_pitch, (const MeJob16*)ctx->d_jobs, ctx->d_first_strip, pl.jobs, pl.n_wg16,
                           pl.pdw, pl.strip_rows, fp->fen, fp->bit_depth, d_mv, d_sad, s);
  if (pl.tile8)
    return launch_search8_split(ctx, curs, cur_ctus_x, refs, ref_pitch, (const MeJob16*)ctx->d_jobs, ctx->d_first_strip, pl.jobs, 4,
                                fp->fen, d_mv, d_sad, s);
  const int head = pl.tail_first, n_tail = pl.jobs - head;
  if (!n_tail) return launch_search8(ctx, curs, cur_ctus_x, refs, ref_pitch, (const MeJob*)ctx->d_jobs, head, fp->fen, d_mv, d_sad, s);
  unsigned long long* best = nullptr;
  int rc = merge_table(ctx, pl.one_launch ? pl.jobs : n_tail, nullptr, s, &best);   // (preset, if it has to be, before the head runs, not between the two)
  if (rc) return rc;
  if (pl.one_launch)
    return launch_search8_segments(ctx, curs, cur_ctus_x, refs, ref_pitch, ctx->d_jobs, ctx->d_first_strip, pl.jobs, head + pl.tail_wgs, fp->fen, d_mv, d_sad, s, best);
  rc = launch_search8(ctx, curs, cur_ctus_x, refs, ref_pitch, (const MeJob*)ctx->d_jobs, head, fp->fen, d_mv, d_sad, s);
  if (rc) return rc;
  return launch_search8_segments(ctx, curs, cur_ctus_x, refs, ref_pitch, (const uint8_t*)ctx->d_jobs + pl.tail_jobs_off, ctx->d_first_strip, n_tail, pl.tail_wgs,
                                 fp->fen, d_mv + (size_t)head * 2 * HMME_NUM_CTU_PARTS, d_sad + (size_t)head * HMME_NUM_CTU_PARTS, s, best);
}

// n_pairs (current, reference) picture pairs of one size in one launch: validates, orders the streams, fills the two plane sets
namespace {
struct PairLaunch {
  RefSet curs, refs;
  RefSet cur_blocks;   // the current pictures' CTU-blocked copies: what the search kernels read (the refinement reads the padded planes)
  int first = 0, count = 0;
};
int pairs_begin(hmme_ctx* ctx, const hmme_plane* const* curs, const hmme_plane* const* refs, int n_pairs, const hmme_frame_params* fp,
                hipStream_t s, PairLaunch* pl) {
  if (!curs || !refs || n_pairs < 1 || n_pairs > hmme::kMaxRefs) return fail(ctx, HMME_ERR_ARG, "%d picture pairs outside 1..%d", n_pairs, hmme::kMaxRefs);
  pl->curs = one_ref(nullptr); pl->refs = one_ref(nullptr); pl->cur_blocks = one_ref(nullptr);
  for (int r = 0; r < n_pairs; ++r) {
    int rc = check_frame_args(ctx, curs[r], refs[r], fp, &pl->first, &pl->count);
    if (rc) return rc;
    if (refs[r]->pitch != refs[0]->pitch || curs[r]->pitch != curs[0]->pitch || curs[r]->width != curs[0]->width || curs[r]->height != curs[0]->height)
      return fail(ctx, HMME_ERR_ARG, "the planes of one launch differ in size");
    pl->curs.base[r] = curs[r]->origin();
    pl->cur_blocks.base[r] = curs[r]->d_blocks;
    pl->refs.base[r] = refs[r]->origin();
  }
  if (pl->count == 0) return HMME_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = scratch_acquire(ctx, s);
  for (int r = 0; r < n_pairs && rc == HMME_OK; ++r) {
    rc = plane_wait(ctx, curs[r], s);
    if (rc == HMME_OK) rc = plane_wait(ctx, refs[r], s);
  }
  return rc;
}
// after the kernels are enqueued (or an enqueue failed): the planes have a reader on `s`, the scratch a user
int pairs_end(hmme_ctx* ctx, const hmme_plane* const* curs, const hmme_plane* const* refs, int n_pairs, hipStream_t s, int rc) {
  for (int r = 0; r < n_pairs; ++r) {
    int r2 = (r == 0 || curs[r] != curs[r - 1]) ? plane_read_chain(ctx, curs[r], s) : HMME_OK;
    if (r2 == HMME_OK) r2 = plane_read_chain(ctx, refs[r], s);
    if (rc == HMME_OK) rc = r2;
  }
  // one event for the whole launch (it was one per plane and one for the scratch: 3 .. 33 packets between two kernels of a stream)
  const int r3 = launch_end(ctx, s);
  if (r3 == HMME_OK)
    for (int r = 0; r < n_pairs; ++r) { plane_read_mark(ctx, curs[r], s); plane_read_mark(ctx, refs[r], s); }
  return rc ? rc : r3;
}
}  // namespace

int hmme_search_pairs_device(hmme_ctx* ctx, const hmme_plane* const* curs, const hmme_plane* const* refs, int n_pairs,
                             const hmme_frame_params* fp, const void* d_pred_q, void* d_out_mv, void* d_out_sad, void* stream) {
  if (!ctx) return HMME_ERR_ARG;
  if (!d_out_mv || !d_out_sad) return fail(ctx, HMME_ERR_ARG, "null output buffer");
  hipStream_t s = (hipStream_t)stream;
  PairLaunch pl;
  int rc = pairs_begin(ctx, curs, refs, n_pairs, fp, s, &pl);
  if (rc || pl.count == 0) return rc;
  FramePlan plan;
  rc = prep_jobs(ctx, curs[0], fp, d_pred_q, pl.first, pl.count, n_pairs, s, &plan);
  if (rc == HMME_OK) rc = run_search(ctx, pl.cur_blocks, curs[0]->ctus_x, pl.refs, refs[0]->pitch, fp, plan, (int16_t*)d_out_mv, (uint32_t*)d_out_sad, s);
  return pairs_end(ctx, curs, refs, n_pairs, s, rc);
}

int hmme_search_frame_multi_device(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* const* refs, int n_refs,
                                   const hmme_frame_params* fp, const void* d_pred_q, void* d_out_mv, void* d_out_sad, void* stream) {
  if (!ctx) return HMME_ERR_ARG;
  if (!refs || n_refs < 1 || n_refs > hmme::kMaxRefs) return fail(ctx, HMME_ERR_ARG, "n_refs %d outside 1..%d", n_refs, hmme::kMaxRefs);
  const hmme_plane* curs[hmme::kMaxRefs];
  for (int r = 0; r < n_refs; ++r) curs[r] = cur;
  return hmme_search_pairs_device(ctx, curs, refs, n_refs, fp, d_pred_q, d_out_mv, d_out_sad, stream);
}

int hmme_search_frame_device(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* ref, const hmme_frame_params* fp,
                             const void* d_pred_q, void* d_out_mv, void* d_out_sad, void* stream) {
  return hmme_search_frame_multi_device(ctx, cur, &ref, 1, fp, d_pred_q, d_out_mv, d_out_sad, stream);
}

// host-facing frame calls: device staging for `need` (CTU, reference) result tables and their predictors
static int ensure_frame_buffers(hmme_ctx* ctx, size_t need) {
  if ((size_t)ctx->out_cap >= need) return HMME_OK;
  hipFree(ctx->d_mv); hipFree(ctx->d_sad); hipFree(ctx->d_pred);
  ctx->d_mv = nullptr; ctx->d_sad = nullptr; ctx->d_pred = nullptr; ctx->out_cap = 0;
  HIP_TRY(ctx, hipMalloc(&ctx->d_mv, sizeof(int16_t) * 2 * HMME_NUM_CTU_PARTS * need));
  HIP_TRY(ctx, hipMalloc(&ctx->d_sad, sizeof(uint32_t) * HMME_NUM_CTU_PARTS * need));
  HIP_TRY(ctx, hipMalloc(&ctx->d_pred, sizeof(int16_t) * 2 * need));
  ctx->out_cap = (int)need;
  return HMME_OK;
}

int hmme_search_frame_multi(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* const* refs, int n_refs,
                            const hmme_frame_params* fp, const int16_t* pred_q, int16_t* out_mv, uint32_t* out_sad) {
  if (!ctx) return HMME_ERR_ARG;
  if (!refs || n_refs < 1 || n_refs > hmme::kMaxRefs) return fail(ctx, HMME_ERR_ARG, "n_refs %d outside 1..%d", n_refs, hmme::kMaxRefs);
  int first, count;
  int rc = check_frame_args(ctx, cur, refs[0], fp, &first, &count);
  if (rc) return rc;
  if (!out_mv || !out_sad) return fail(ctx, HMME_ERR_ARG, "null output buffer");
  if (count == 0) return HMME_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int n_ctu = hmme_num_ctus(cur->width, cur->height);
  const size_t need = (size_t)n_ctu * n_refs;
  rc = ensure_frame_buffers(ctx, need);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  if (pred_q) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_pred, pred_q, sizeof(int16_t) * 2 * need, hipMemcpyHostToDevice, s));
  rc = hmme_search_frame_multi_device(ctx, cur, refs, n_refs, fp, pred_q ? ctx->d_pred : nullptr, ctx->d_mv, ctx->d_sad, s);
  if (rc) return rc;
  const size_t res = (size_t)count * n_refs;
  HIP_TRY(ctx, hipMemcpyAsync(out_mv, ctx->d_mv, sizeof(int16_t) * 2 * HMME_NUM_CTU_PARTS * res, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(out_sad, ctx->d_sad, sizeof(uint32_t) * HMME_NUM_CTU_PARTS * res, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return HMME_OK;
}

int hmme_search_frame(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* ref, const hmme_frame_params* fp,
                      const int16_t* pred_q, int16_t* out_mv, uint32_t* out_sad) {
  return hmme_search_frame_multi(ctx, cur, &ref, 1, fp, pred_q, out_mv, out_sad);
}

// ---- fractional-pel refinement -------------------------------------------------------------------------------
namespace {
// cover table of me_frac_kernel: for each of the 64 8x8 positions of the CTU the 18 kind-8 slots (width and height
// multiples of 8: 8x8 Hadamard blocks, xGetHADs) that contain it, then for each of the 256 4x4 positions the 6 other
// slots that contain it; ascending slot ids
int build_frac_cover(hmme_ctx* ctx) {
  if (ctx->d_frac_cover) return HMME_OK;
  std::vector<uint16_t> cover;
  for (int kind8 = 1; kind8 >= 0; --kind8) {
    const int step = kind8 ? 8 : 4, per = kind8 ? hmme::kFracCover8 : hmme::kFracCover4;
    for (int py = 0; py < 64; py += step)
      for (int px = 0; px < 64; px += step) {
        int n = 0;
        for (int slot = 0; slot < HMME_NUM_CTU_PARTS; ++slot) {
          int x, y, w, h;
          hmme_slot_rect(slot, &x, &y, &w, &h);
          if (((w % 8 == 0) && (h % 8 == 0)) != (kind8 == 1)) continue;
          if (px >= x && px < x + w && py >= y && py < y + h) { cover.push_back((uint16_t)slot); ++n; }
        }
        if (n != per) return fail(ctx, HMME_ERR_DEVICE, "internal: %d slots cover a %dx%d position, expected %d", n, step, step, per);
      }
  }
  HIP_TRY(ctx, hipMalloc(&ctx->d_frac_cover, sizeof(uint16_t) * cover.size()));
  HIP_TRY(ctx, hipMemcpy(ctx->d_frac_cover, cover.data(), sizeof(uint16_t) * cover.size(), hipMemcpyHostToDevice));
  return HMME_OK;
}
}  // namespace

int hmme_refine_pairs_device(hmme_ctx* ctx, const hmme_plane* const* curs, const hmme_plane* const* refs, int n_pairs,
                             const hmme_frame_params* fp, const void* d_pred_q, const void* d_int_mv, int use_hadamard,
                             void* d_out_qmv, void* d_out_cost, void* stream) {
  if (!ctx) return HMME_ERR_ARG;
  if (!d_int_mv || !d_out_qmv || !d_out_cost) return fail(ctx, HMME_ERR_ARG, "null buffer");
  hipStream_t s = (hipStream_t)stream;
  PairLaunch pl;
  int rc = pairs_begin(ctx, curs, refs, n_pairs, fp, s, &pl);
  if (rc || pl.count == 0) return rc;
  const int jobs = pl.count * n_pairs;
  rc = build_frac_cover(ctx);
  if (rc == HMME_OK) {
    size_t cap = ctx->frac_jobs_bytes;
    rc = ensure(ctx, (uint8_t**)&ctx->d_frac_jobs, &cap, sizeof(MeJob) * (size_t)jobs + 64,   // + the launch's job counter behind the table
                sizeof(MeJob) * (size_t)pl.count * hmme::kMaxRefs + 4096);
    if (cap != ctx->frac_jobs_bytes) ctx->frac_jobs_tag.valid = false;
    ctx->frac_jobs_bytes = cap;
  }
  if (rc == HMME_OK) {
    uint32_t* counter = (uint32_t*)((uint8_t*)ctx->d_frac_jobs + ((sizeof(MeJob) * (size_t)jobs + 15) & ~(size_t)15));
    const int had = use_hadamard ? 1 : 0, wide = curs[0]->bps == 2 ? 1 : 0;
    const int grid = frac_grid(ctx, wide, had, jobs);
    // one workgroup per job (the default) on the two-wave builds: every workgroup derives its job itself (FracPrep) -- no job table, no
    // launch in front of this one (1080p: 0.095 -> 0.090 ms).  A table is read by the
    // job-walking launch of HMME_FRAC_GRID: its prep kernel is also what resets the job counter
    const hmme::FracPrep prep = {(const int16_t*)d_pred_q, (uint32_t)pl.first | (uint32_t)pl.count << 16, (uint32_t)curs[0]->width | (uint32_t)curs[0]->height << 16, fp->search_range};
    static const bool table = std::getenv("HMME_FRAC_JOB_TABLE") != nullptr;   // A/B: the job table and its kernel as before
    const bool walk = grid < jobs;
    const bool packable = pl.count <= 0xffff && pl.first <= 0xffff;   // FracPrep packs the CTU range into 16 + 16 bits (a 16384 x 16384 picture has 65 536 CTUs)
    const bool need_table = walk || table || !packable;
    static const bool no_cache = std::getenv("HMME_NO_TABLE_CACHE") != nullptr;
    hmme_ctx::TableTag tag;
    tag.valid = !d_pred_q && !no_cache && !walk;   // (the job-walking mode's prep kernel also resets the job counter: every launch)
    tag.w = curs[0]->width; tag.h = curs[0]->height; tag.bit_depth = fp->bit_depth; tag.sr = fp->search_range; tag.first = pl.first; tag.count = pl.count;
    tag.pairs = n_pairs; tag.buf = ctx->d_frac_jobs; tag.stream = (void*)s;
    const bool have_table = need_table && tag.same(ctx->frac_jobs_tag);
    if (need_table && !have_table) ctx->frac_jobs_tag = tag;
    if (need_table && !have_table)
      hipLaunchKernelGGL(hmme::me_prep_jobs_kernel, dim3((jobs + 255) / 256), dim3(256), 0, s, (MeJob*)ctx->d_frac_jobs, (const int16_t*)d_pred_q,
                         pl.first, pl.count, n_pairs, curs[0]->width, curs[0]->height, fp->search_range, 0, jobs, 0, counter);
    rc = frac_lds_optin(ctx, wide, had, 0);
    if (rc != HMME_OK) return pairs_end(ctx, curs, refs, n_pairs, s, rc);
    hipLaunchKernelGGL(frac_kernel(wide, had, 0), dim3(grid), dim3(hmme::frac_threads(wide ? 2 : 1)), hmme::frac_lds_bytes(wide ? 2 : 1), s, pl.curs,
                       curs[0]->pitch, pl.refs, refs[0]->pitch, need_table ? (const MeJob*)ctx->d_frac_jobs : (const MeJob*)nullptr, prep, jobs, walk ? counter : (uint32_t*)nullptr, ctx->d_frac_cover,
                       (const int16_t*)d_int_mv, ctx->lambda_q16,
                       fp->bit_depth, kNoWp, (int16_t*)d_out_qmv, (uint32_t*)d_out_cost);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = fail(ctx, HMME_ERR_DEVICE, "refinement launch -> %s", hipGetErrorString(e));
  }
  return pairs_end(ctx, curs, refs, n_pairs, s, rc);
}

int hmme_refine_frame_multi_device(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* const* refs, int n_refs,
                                   const hmme_frame_params* fp, const void* d_pred_q, const void* d_int_mv, int use_hadamard,
                                   void* d_out_qmv, void* d_out_cost, void* stream) {
  if (!ctx) return HMME_ERR_ARG;
  if (!refs || n_refs < 1 || n_refs > hmme::kMaxRefs) return fail(ctx, HMME_ERR_ARG, "n_refs %d outside 1..%d", n_refs, hmme::kMaxRefs);
  const hmme_plane* curs[hmme::kMaxRefs];
  for (int r = 0; r < n_refs; ++r) curs[r] = cur;
  return hmme_refine_pairs_device(ctx, curs, refs, n_refs, fp, d_pred_q, d_int_mv, use_hadamard, d_out_qmv, d_out_cost, stream);
}

int hmme_refine_frame(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* ref, const hmme_frame_params* fp, const int16_t* pred_q,
                      const int16_t* int_mv, int use_hadamard, int16_t* out_qmv, uint32_t* out_cost) {
  int first, count;
  int rc = check_frame_args(ctx, cur, ref, fp, &first, &count);
  if (rc) return rc;
  if (!int_mv || !out_qmv || !out_cost) return fail(ctx, HMME_ERR_ARG, "null buffer");
  if (count == 0) return HMME_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int n_ctu = hmme_num_ctus(cur->width, cur->height);
  const size_t slots = (size_t)HMME_NUM_CTU_PARTS * n_ctu;
  if (ctx->refine_cap < slots) {
    hipFree(ctx->d_imv); hipFree(ctx->d_qmv); hipFree(ctx->d_fcost);
    ctx->d_imv = nullptr; ctx->d_qmv = nullptr; ctx->d_fcost = nullptr; ctx->refine_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_imv, sizeof(int16_t) * 2 * slots));
    HIP_TRY(ctx, hipMalloc(&ctx->d_qmv, sizeof(int16_t) * 2 * slots));
    HIP_TRY(ctx, hipMalloc(&ctx->d_fcost, sizeof(uint32_t) * slots));
    ctx->refine_cap = slots;
  }
  rc = ensure_frame_buffers(ctx, (size_t)n_ctu);   // the predictors travel in the search path's staging buffer
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const size_t res = (size_t)HMME_NUM_CTU_PARTS * count;
  if (pred_q) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_pred, pred_q, sizeof(int16_t) * 2 * (size_t)n_ctu, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_imv, int_mv, sizeof(int16_t) * 2 * res, hipMemcpyHostToDevice, s));
  rc = hmme_refine_frame_multi_device(ctx, cur, &ref, 1, fp, pred_q ? ctx->d_pred : nullptr, ctx->d_imv, use_hadamard, ctx->d_qmv,
                                      ctx->d_fcost, s);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(out_qmv, ctx->d_qmv, sizeof(int16_t) * 2 * res, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(out_cost, ctx->d_fcost, sizeof(uint32_t) * res, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return HMME_OK;
}

int hmme_test_time_search_kernel(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* ref, const hmme_frame_params* fp,
                            const void* d_pred_q, void* d_out_mv, void* d_out_sad, void* stream, int reps, float* avg_ms) {
  int first, count;
  int rc = check_frame_args(ctx, cur, ref, fp, &first, &count);
  if (rc) return rc;
  if (!avg_ms || reps < 1) return fail(ctx, HMME_ERR_ARG, "hmme_test_time_search_kernel: bad reps/avg_ms");
  hipStream_t s = (hipStream_t)stream;
  PairLaunch pl;
  rc = pairs_begin(ctx, &cur, &ref, 1, fp, s, &pl);
  if (rc || pl.count == 0) return rc;
  // job table once (it is not part of the timed kernel), then `reps` launches of the search kernel(s) alone
  FramePlan plan;
  rc = prep_jobs(ctx, cur, fp, d_pred_q, first, count, 1, s, &plan);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipSuccess;
  float ms = 0.f;
  if (rc == HMME_OK && (e = hipEventCreate(&e0)) == hipSuccess && (e = hipEventCreate(&e1)) == hipSuccess) {
    e = hipEventRecord(e0, s);
    for (int i = 0; i < reps && rc == HMME_OK && e == hipSuccess; ++i)
      rc = run_search(ctx, pl.cur_blocks, cur->ctus_x, pl.refs, ref->pitch, fp, plan, (int16_t*)d_out_mv, (uint32_t*)d_out_sad, s);
    if (rc == HMME_OK && e == hipSuccess) e = hipEventRecord(e1, s);
    if (rc == HMME_OK && e == hipSuccess) e = hipEventSynchronize(e1);
    if (rc == HMME_OK && e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  }
  if (e0) hipEventDestroy(e0);
  if (e1) hipEventDestroy(e1);
  if (rc == HMME_OK && e != hipSuccess) rc = fail(ctx, HMME_ERR_DEVICE, "timing the search kernel: %s", hipGetErrorString(e));
  if (rc == HMME_OK) *avg_ms = ms / reps;
  return pairs_end(ctx, &cur, &ref, 1, s, rc);
}

}  // extern "C"
