/*
 * hmme.h -- C ABI of the MI355X-native integer motion-estimation engine for HM 16.4.
 *
 * This is the drop-in boundary: plain C types, no exceptions, no torch/HIP types in the
 * signatures (streams and device buffers travel as void*).  It replaces what the reference's
 * OpenCL add-on does behind TEncOpenCL (all citations relative to /root/reference/):
 *
 *   hmme_create            <- TEncOpenCL::findDevice + compileKernelSource + createBuffers
 *                             (source/Lib/TLibEncoder/TEncOpenCL.cpp:69, :139, :195; called from
 *                             TEncTop::xInitOpenCL, TEncTop.cpp:1116-1162)
 *   hmme_set_lambda*       <- TEncOpenCL::setLambda (TEncOpenCL.h:121; TEncSlice.cpp:150)
 *   hmme_search_ctu        <- TEncOpenCL::calcMotionVectors + getX/getY/getRuiCost
 *                             (TEncOpenCL.cpp:240-362, TEncOpenCL.h:117-119; caller
 *                             TEncSearch::xMotionEstimation, TEncSearch.cpp:3743-3765)
 *   hmme_refine_ctu,       <- TEncSearch::xPatternSearchFracDIF (TEncSearch.cpp:4294-4331) for the 593 slots of that CTU,
 *   hmme_search_refine_ctu    the per-PU call at TEncSearch.cpp:3798 turned into a table lookup like the integer search
 *   hmme_search_frame*     <- the same search batched over every CTU of a picture (the
 *                             reference has no batched form: it launches 2*(2SR+1)^2 kernels per
 *                             CTU from the host, TEncOpenCL.cpp:312-333)
 *   hmme_search_pairs_device, <- the same for up to 16 (current, reference) picture pairs of a GOP in one launch
 *   hmme_refine_pairs_device     (cfg/encoder_randomaccess_main.cfg:28-31, cfg/encoder_lowdelay_P_main.cfg:24-27)
 *   hmme_plane_*           <- the padded reference plane calcMotionVectors reads
 *                             (TComPicYuv, TLibCommon/TComPicYuv.cpp:91-92, 214-262); hmme_plane_upload_* take what
 *                             TVideoIOYuv::read delivers (TVideoIOYuv.cpp:247, :680: 8-bit or 16-bit little-endian samples)
 *   hmme_last_error        <- TEncOpenCL::checkError (TEncOpenCL.h:93-101)
 *   hmme_destroy           <- TEncOpenCL::~TEncOpenCL (TEncOpenCL.cpp:38-66)
 *
 * Results use the reference's slot order (TComDataCU::getIndexBlock, TComDataCU.cpp:3379-3391):
 * out_mv is laid out exactly like TComMv[NUM_CTU_PARTS] ({Short hor, Short ver}, integer pels)
 * and out_sad like Distortion[NUM_CTU_PARTS], so one memcpy fills
 * TEncSearch::allMotionVectors[list][refIdx] / allRuiCost[list][refIdx] (TEncSearch.h:114-115).
 *
 * Arithmetic is HM's CPU arithmetic (TEncSearch::xPatternSearch, TEncSearch.cpp:3835-3897):
 * predictor-relative MV cost, window LT..RB, strict '<' in raster order, optional FEN row
 * sub-sampling.  hmme_params_ocl_compat() selects what cl/sad.cl does instead (pred (0,0),
 * window LT..LT+2*SR, all rows).
 *
 * Threading: a context is not thread-safe; use one per host thread / per GPU.
 * Streams: a context owns scratch (job tables, merge tables, staging) that every frame call reuses, and planes are filled
 * asynchronously by hmme_plane_set_device_u8.  The library orders these itself: a *_device call issued on another stream than the
 * context's previous frame call first waits -- on the device, with hipStreamWaitEvent, never blocking the host -- for that call's
 * last use of the scratch, every search / refinement waits for the last fill of each plane it reads if that fill ran on
 * another stream, and every fill / upload of a plane waits for the last search / refinement that read it on another stream (so a
 * ring of planes can be refilled on a copy stream while the compute stream is still searching older contents: tools/me_sequence.py
 * --stream).  So calls of one context may be spread over streams; they serialise where they share scratch.  Output buffers
 * are the caller's: reading d_out_* on another stream than the one passed in needs the caller's own event.  The synchronous
 * host-facing calls run on a private non-blocking stream of the context and return when done.
 * Every function returns HMME_OK (0) or a negative HMME_ERR_* code; nothing ever falls back
 * to a CPU implementation.
 */
#ifndef HMME_H
#define HMME_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct of this header changes layout or an entry point changes meaning; hmme_abi_version() returns the value the
 * library was built with.  TEncOpenCL::findDevice and hmme/api.py refuse a library whose version differs from the header they were
 * written against (hmme_search_params grew by `shift_free` in version 2: a caller built against version 1 would have the library
 * read 4 bytes past its struct).  3: hmme_search_pairs_device / hmme_refine_pairs_device, asynchronous uploads, bi-prediction
 * origins in the refinement calls.  4: hmme_search_ctu_w (explicit weighted prediction); hmme_time_search_kernel and
 * hmme_debug_device_address left this header (include/hmme_test.h).  5: hmme_set_error_printing.  6: hmme_set_error_printing returns the
 * previous setting; frame calls refuse planes of another context. */
#define HMME_ABI_VERSION 6
int hmme_abi_version(void);
/* identifies the kernel sources + build flags the library was compiled from (bench.py ties committed counter summaries to it) */
const char* hmme_build_id(void);

#define HMME_NUM_CTU_PARTS 593 /* TLibCommon/TypeDef.h:263 */
#define HMME_CTU_SIZE 64
#define HMME_MAX_SEARCH_RANGE 128 /* any bit depth, frame and per-CTU calls: windows up to 257 x 257 candidates (8-bit windows beyond
                                      129 x 129 run as 2 x 2 tiles) */

enum {
  HMME_OK = 0,
  HMME_ERR_ARG = -1,     /* invalid argument (null pointer, window larger than sr_max, window / predictor beyond int16, ...) */
  HMME_ERR_DEVICE = -2,  /* no usable gfx950 device / HIP runtime error */
  HMME_ERR_RANGE = -3,   /* sample outside the range of the bit depth (bi-prediction origins of hmme_search_ctu excepted) */
  HMME_ERR_NOMEM = -4,
  HMME_ERR_UNSUPPORTED = -5
};

typedef struct hmme_ctx hmme_ctx;
typedef struct hmme_plane hmme_plane;

/* one (CTU, reference picture) search; integer-pel window, quarter-pel predictor */
typedef struct hmme_search_params {
  int lt_x, lt_y;      /* cMvSrchRngLT after xSetSearchRange (TEncSearch.cpp:3814-3830) */
  int rb_x, rb_y;      /* cMvSrchRngRB, inclusive */
  int pred_x, pred_y;  /* m_pcRdCost->setPredictor(*pcMvPred), quarter pels (TEncSearch.cpp:3737) */
  int fen;             /* m_pcEncCfg->getUseFastEnc() (TEncSearch.cpp:3853-3859) */
  int bit_depth;       /* 8 (packed-byte path) or 9..12 (16-bit path): SAD >> (bitDepth-8), TComRdCost.cpp:520-521 */
  int shift_free;      /* 1: no >> (bitDepth-8) on the SAD -- what cl/sad.cl computes for any Pel width;
                          bit_depth then only states the sample range.  A call is refused (HMME_ERR_UNSUPPORTED) when the samples it
                          was handed could produce a 64x64 sum + MV cost >= 8 000 000, i.e. when the largest |cur - ref| its two
                          blocks admit exceeds 1 937: never at <= 10 bit, never at 9 bit with bi-prediction origins */
} hmme_search_params;

/* one whole-picture search: window derived per CTU from the predictor exactly like
 * xSetSearchRange + TComDataCU::clipMv (TComDataCU.cpp:2907-2920) */
typedef struct hmme_frame_params {
  int search_range;  /* SearchRange (cfg/encoder_lowdelay_P_main.cfg:33) */
  int fen;           /* FEN (cfg/encoder_lowdelay_P_main.cfg:34) */
  int bit_depth;
  int ctu_first;     /* first CTU (raster order) and number of CTUs to search; count < 0 = to the end */
  int ctu_count;
} hmme_frame_params;

/* ---- context ------------------------------------------------------------------------- */
int hmme_create(int device, int sr_max, unsigned flags, hmme_ctx** out);
void hmme_destroy(hmme_ctx* ctx);
const char* hmme_last_error(const hmme_ctx* ctx); /* ctx may be NULL: error of the calling thread's last failed hmme_create */
/* A failed call prints its message on stderr (as TEncOpenCL::checkError does, TEncOpenCL.h:93-101) and keeps it for hmme_last_error.
 * on = 0 keeps it only -- for a caller that PROBES with a call it expects to be refused (TEncOpenCL's reference-mode call tries the
 * sample width it has latched and widens it on HMME_ERR_RANGE, instead of scanning every window for its largest sample first).
 * Returns the previous setting (1 / 0; 1 for a NULL context), so that a probe restores what its caller had chosen. */
int hmme_set_error_printing(hmme_ctx* ctx, int on);
const char* hmme_device_info(const hmme_ctx* ctx);
int hmme_device_index(const hmme_ctx* ctx);   /* the HIP device the context lives on (host code that makes its own HIP calls beside the library's) */
int hmme_set_lambda(hmme_ctx* ctx, double lambda);         /* m_lambda = floor(65536*sqrt(lambda)) */
int hmme_set_lambda_q16(hmme_ctx* ctx, uint32_t lambda_q16);
uint32_t hmme_get_lambda_q16(const hmme_ctx* ctx);

/* the parameter set that reproduces the reference GPU path's choices */
void hmme_params_ocl_compat(hmme_search_params* p, int lt_x, int lt_y, int search_range);
/* xSetSearchRange + clipMv on the host (exported so callers and tests can derive LT/RB) */
void hmme_set_search_range(int pred_x_q, int pred_y_q, int search_range, int cu_x, int cu_y, int pic_w,
                           int pic_h, int* lt_x, int* lt_y, int* rb_x, int* rb_y);

/* ---- slot layout (TComDataCU::getIndexBlock, TComDataCU.cpp:3379-3391 + case table :4676-6461) ------ */
/* slot 0..592 of a PU: part_size = HM PartSize enum (0 2Nx2N, 1 2NxN, 2 Nx2N, 4 2NxnU, 5 2NxnD, 6 nLx2N, 7 nRx2N),
 * depth 0..3 (CU size 64 >> depth), part_idx 0/1, abs_z_idx = z-order address of the CU in 4x4 units.
 * Returns -1 for combinations the reference does not tabulate (NxN, AMP at 8x8). */
int hmme_slot_index(int part_size, int depth, int part_idx, int abs_z_idx);
/* rectangle of a slot inside the 64x64 CTU; returns 0 or HMME_ERR_ARG */
int hmme_slot_rect(int slot, int* x, int* y, int* w, int* h);
/* The 425-entry table layout of an encoder built with AMP_ENC_SPEEDUP (TypeDef.h:206, :260-261; TComDataCU.cpp:3393-4675; the
 * reference's `calcSAD` kernel, cl/sad.cl:4-138) -- the macro is 0 in the reference tree as shipped, so this is a view for such a
 * build, not a second search: the same rectangles as the 593 layout without the AMP shapes.  hmme_slot_index_amp_off = that build's
 * getIndexBlock (-1 for what it does not tabulate); hmme_amp_off_slot maps an entry of the 425 layout to the slot of the 593
 * layout that holds the same rectangle; hmme_compact_amp_off turns a call's 593 results into the 425 tables. */
int hmme_slot_index_amp_off(int part_size, int depth, int part_idx, int abs_z_idx);
int hmme_amp_off_slot(int index_amp_off);
int hmme_compact_amp_off(const int16_t* mv593, const uint32_t* sad593, int16_t* mv425, uint32_t* sad425);

/* ---- per-CTU drop-in (host buffers, HM `Pel` = int16) --------------------------------- */
/* ctu: 64x64 current block (TEncSearch.cpp:3747); ref_at_ctu_origin: reference plane at the CTU
 * origin inside its padded buffer, as handed to calcMotionVectors.  Synchronous.  Current-block samples may be the
 * bi-prediction origin 2*org - pred (TEncSearch.cpp:3702-3712), i.e. lie in [-maxv, 2*maxv].
 * out_mv: int16[593][2] (hor, ver), out_sad: uint32[593] (pure SAD at the arg-min = ruiCost). */
int hmme_search_ctu(hmme_ctx* ctx, const int16_t* ctu, int ctu_stride, const int16_t* ref_at_ctu_origin,
                    int ref_stride, const hmme_search_params* p, int16_t* out_mv, uint32_t* out_sad);

/* The same search in a slice with explicit weighted prediction (TEncSearch::setWpScalingDistParam, TEncSearch.cpp:3740, :5594-5635:
 * m_cDistParam.bApplyWeight + wpCur): every candidate is priced by TComRdCostWeightPrediction::xGetSADw
 * (TComRdCostWeightPrediction.cpp:55-90), |org - (((w0 * ref + round) >> shift) + offset)| summed over EVERY row (p->fen is not
 * consulted: each xGetSAD* hands over to xGetSADw before it looks at iSubShift, TComRdCost.cpp:467-469), the prediction unclipped,
 * the whole-block sum >> (bitDepth-8).  `wp` = the luma WPScalingParam of the reference picture (w, offset, shift, round; TComSlice.h:1178-
 * 1190).  HMME_ERR_UNSUPPORTED when a weighted sample of the window would leave int16 (HM keeps it in a Pel and wraps; not
 * reproduced) or the sums could exceed the engine's cost field: the caller then searches on the CPU.  Integer search only. */
typedef struct hmme_weight { int w0, offset, shift, round; } hmme_weight;
int hmme_search_ctu_w(hmme_ctx* ctx, const int16_t* ctu, int ctu_stride, const int16_t* ref_at_ctu_origin, int ref_stride,
                      const hmme_search_params* p, const hmme_weight* wp, int16_t* out_mv, uint32_t* out_sad);
/* ... and with xPatternSearchFracDIF of the 593 winners in the same call, as hmme_search_refine_ctu: in such a slice the refinement's
 * distortion is xGetHADsw / xGetSADw (TComRdCostWeightPrediction.cpp:407-470) -- the interpolated, clipped prediction weighted sample by
 * sample before the difference is taken.  Additionally refused (HMME_ERR_UNSUPPORTED; hmme_search_ctu_w still serves the call) when
 * the weighted sample differences of the block could exceed 4095 (the Hadamard sums are kept exactly). */
int hmme_search_refine_ctu_w(hmme_ctx* ctx, const int16_t* ctu, int ctu_stride, const int16_t* ref_at_ctu_origin, int ref_stride,
                             const hmme_search_params* p, const hmme_weight* wp, int use_hadamard, int16_t* out_mv, uint32_t* out_sad,
                             int16_t* out_qmv, uint32_t* out_cost);
int hmme_refine_ctu_w(hmme_ctx* ctx, const int16_t* ctu, int ctu_stride, const int16_t* ref_at_ctu_origin, int ref_stride,
                      const hmme_search_params* p, const hmme_weight* wp, const int16_t* int_mv, int use_hadamard, int16_t* out_qmv,
                      uint32_t* out_cost);   /* the weighted refinement alone, at the caller's integer MVs (as hmme_refine_ctu) */

/* The step after the search, for the same CTU: TEncSearch::xPatternSearchFracDIF (TEncSearch.cpp:4294-4331, called per PU at :3798)
 * for all 593 slots -- half- then quarter-pel refinement around each slot's integer MV, HM's 8-tap interpolation, Hadamard
 * (use_hadamard, HadamardME) or SAD distortion, MV cost against p->pred.  out_qmv: int16[593][2] quarter-pel MVs
 * (int << 2) + (half << 1) + quarter; out_cost: uint32[593], the ruiCost xPatternSearchFracDIF returns (distortion + MV cost).
 * hmme_search_refine_ctu = hmme_search_ctu + the refinement of its winners in ONE call (block and window are staged once);
 * hmme_refine_ctu refines the caller's integer MVs int_mv[593][2] (entries outside the window LT..RB are clamped to it).
 * Unlike hmme_search_ctu these read the reference 4 samples (+ up to 3 for alignment) beyond the window (64 + 2*SR)^2 on every
 * side -- the interpolation filter's support, which xPatternSearchFracDIF reads as well (HM planes carry an 80-sample margin).
 * Bi-prediction origins (current-block samples in [-maxv, 2*maxv]: the bBi pass, TEncSearch.cpp:3702-3712, :3798) are refined
 * like any other block: the interpolated reference is clipped to the sample range, the origin is not.
 * shift_free: hmme_search_refine_ctu's integer leg honours it (out_sad unshifted); the refinement never does -- out_cost is always
 * HM's ((distortion >> (bitDepth-8)) + MV cost), so with shift_free = 1 at more than 8 bits the two outputs are on different
 * scales.  HM-arithmetic callers leave it 0. */
int hmme_search_refine_ctu(hmme_ctx* ctx, const int16_t* ctu, int ctu_stride, const int16_t* ref_at_ctu_origin, int ref_stride,
                           const hmme_search_params* p, int use_hadamard, int16_t* out_mv, uint32_t* out_sad, int16_t* out_qmv,
                           uint32_t* out_cost);
int hmme_refine_ctu(hmme_ctx* ctx, const int16_t* ctu, int ctu_stride, const int16_t* ref_at_ctu_origin, int ref_stride,
                    const hmme_search_params* p, const int16_t* int_mv, int use_hadamard, int16_t* out_qmv, uint32_t* out_cost);

/* ---- frame path ------------------------------------------------------------------------ */
/* device-resident luma plane with edge-replicated margins; 8-bit planes store bytes, 9..12-bit planes u16.  Every plane also holds its picture
 * area once more CTU by CTU (64 x 64 blocks, contiguous: what a search reads the CURRENT picture from), so a plane costs about
 * (W + 256) x (H + 160) + W x H samples of device memory -- 17.9 MB for an 8-bit 2160p picture.
 * A plane belongs to the context that created it: every frame call refuses planes of another context (HMME_ERR_ARG), and a
 * context's planes are destroyed BEFORE the context (hmme_plane_destroy reads its context). */
int hmme_plane_create(hmme_ctx* ctx, int width, int height, hmme_plane** out);   /* 8-bit */
int hmme_plane_create_ex(hmme_ctx* ctx, int width, int height, int bit_depth, hmme_plane** out);
void hmme_plane_destroy(hmme_plane* plane);
/* upload the width x height picture area of an HM plane (origin = sample (0,0)); borders are
 * re-extended on the device like TComPicYuv::extendPicBorder.  Samples outside [0, 2^bitDepth) are
 * rejected with HMME_ERR_RANGE */
int hmme_plane_upload_pel(hmme_plane* plane, const int16_t* origin, int stride);
int hmme_plane_upload_u8(hmme_plane* plane, const uint8_t* origin, int stride);
/* The same upload, asynchronous on `stream` (hipStream_t): returns once the copy and the border extension are enqueued.  origin:
 * samples of sample_bytes 1 (u8) or 2 (HM's Pel / the little-endian words of a 16-bit YUV file, TVideoIOYuv.cpp:247); any plane
 * bit depth.  The host buffer must stay untouched until the copy has run (the caller's event on `stream`) and should be page-locked
 * (hmme_host_register), otherwise the runtime stages it and the call blocks.  Ordering against searches that still read the
 * plane's previous contents on another stream is the library's (see "Streams").  A sample outside the plane's range cannot be
 * reported by this call: it is latched and returned -- once -- by the next hmme_upload_status (a latch of its own: synchronous
 * uploads of other planes neither see nor clear it). */
int hmme_plane_upload_async(hmme_plane* plane, const void* origin, int stride, int sample_bytes, void* stream);
/* waits for `stream`; HMME_ERR_RANGE if an upload since the last check carried an out-of-range sample */
int hmme_upload_status(hmme_ctx* ctx, void* stream);
/* Optional: page-lock a long-lived host buffer (e.g. the TComPicYuv planes of the decoded picture buffer,
 * TComPicYuv.cpp:80-133) so that uploads from it run at PCIe rate instead of through the runtime's pageable staging.
 * The buffer must stay allocated until hmme_host_unregister; uploads work with or without registration. */
int hmme_host_register(hmme_ctx* ctx, void* buffer, size_t bytes);
int hmme_host_unregister(hmme_ctx* ctx, void* buffer);
/* device-side producers (e.g. a torch tensor): copy a width x height u8 image that already
 * lives in device memory, then extend borders; asynchronous on `stream` (hipStream_t) */
int hmme_plane_set_device_u8(hmme_plane* plane, const void* d_src, int src_pitch, void* stream);
int hmme_plane_width(const hmme_plane* plane);
int hmme_plane_height(const hmme_plane* plane);
int hmme_plane_bit_depth(const hmme_plane* plane);

/* number of CTUs (partial edge CTUs included) of a width x height picture */
int hmme_num_ctus(int width, int height);

/* synchronous, host results.  pred_q: int16[n_ctu][2] quarter-pel predictors indexed by CTU
 * raster address, or NULL for (0,0).  out_mv: int16[count][593][2], out_sad: uint32[count][593] */
int hmme_search_frame(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* ref, const hmme_frame_params* fp,
                      const int16_t* pred_q, int16_t* out_mv, uint32_t* out_sad);
/* asynchronous on `stream`, everything device-resident (d_pred_q may be NULL) */
int hmme_search_frame_device(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* ref,
                             const hmme_frame_params* fp, const void* d_pred_q, void* d_out_mv, void* d_out_sad,
                             void* stream);

/* several reference pictures of one current picture in ONE launch (HM's low-delay P configuration searches 4 per
 * picture, cfg/encoder_lowdelay_P_main.cfg:24-27).  refs: n_refs (<= 16) planes of the picture size.
 * pred_q: int16[n_refs][n_ctu][2] or NULL; out_mv: int16[n_refs][count][593][2]; out_sad: uint32[n_refs][count][593] */
int hmme_search_frame_multi(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* const* refs, int n_refs,
                            const hmme_frame_params* fp, const int16_t* pred_q, int16_t* out_mv, uint32_t* out_sad);
int hmme_search_frame_multi_device(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* const* refs, int n_refs,
                                   const hmme_frame_params* fp, const void* d_pred_q, void* d_out_mv, void* d_out_sad,
                                   void* stream);

/* Several (current, reference) picture PAIRS of one size in one launch (<= 16): an open-loop pass over a sequence (BASELINE config 4:
 * the pairs of cfg/encoder_randomaccess_main.cfg:28-31) searches small pictures several pairs at a time -- a single 1080p pair is
 * 510 workgroups, less than one round of the chip's 512 slots.  hmme_search_frame_multi_device is the case curs[i] == cur.
 * pred_q: int16[n_pairs][n_ctu][2] or NULL; out_mv: int16[n_pairs][count][593][2]; out_sad: uint32[n_pairs][count][593] */
int hmme_search_pairs_device(hmme_ctx* ctx, const hmme_plane* const* curs, const hmme_plane* const* refs, int n_pairs,
                             const hmme_frame_params* fp, const void* d_pred_q, void* d_out_mv, void* d_out_sad, void* stream);

/* ---- fractional-pel refinement: the step after the integer search ------------------------------------------
 * TEncSearch::xPatternSearchFracDIF (TEncSearch.cpp:4294-4331) for every slot of every CTU: half- then quarter-pel
 * refinement around the slot's integer MV with HM's 8-tap interpolation, Hadamard (HadamardME = 1, xGetHADs) or SAD
 * distortion plus the MV cost.  int_mv: int16[n_refs][count][593][2] as produced by hmme_search_frame*.
 * out_qmv: quarter-pel MV (int << 2) + (half << 1) + quarter; out_cost: distortion + MV cost of the winner (the
 * ruiCost xPatternSearchFracDIF returns).  8..12-bit planes, any search range the search accepts.  Integer MVs outside
 * the CTU's search window (never produced by hmme_search_frame*) are clamped to it first. */
int hmme_refine_frame(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* ref, const hmme_frame_params* fp,
                      const int16_t* pred_q, const int16_t* int_mv, int use_hadamard, int16_t* out_qmv, uint32_t* out_cost);
int hmme_refine_frame_multi_device(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* const* refs, int n_refs,
                                   const hmme_frame_params* fp, const void* d_pred_q, const void* d_int_mv, int use_hadamard,
                                   void* d_out_qmv, void* d_out_cost, void* stream);

int hmme_refine_pairs_device(hmme_ctx* ctx, const hmme_plane* const* curs, const hmme_plane* const* refs, int n_pairs,
                             const hmme_frame_params* fp, const void* d_pred_q, const void* d_int_mv, int use_hadamard,
                             void* d_out_qmv, void* d_out_cost, void* stream);

/* ---- environment (diagnostics and A/B measurements; none of these changes a result) ---------
 *   HMME_TRACE=1          one stderr line per context about launch geometry the library derives at run time (with HMME_FRAC_GRID=-1:
 *                         workgroups of the refinement kernel per CU); the TEncOpenCL host module prints its call summary in its destructor
 *   HMME_FRAC_GRID=<n>    workgroups of a refinement launch: default (0) = one per job; n > 0 = exactly n, each taking job after
 *                         job from a counter; -1 = as many of those as the chip holds at a time
 *   HMME_FRAC_JOB_TABLE=1 whole-picture refinement launches: job table written by a kernel in front of the launch (as before round 4's
 *                         end) instead of every workgroup deriving its job itself
 *   HMME_NO_TABLE_CACHE=1 launches without predictors: rebuild the job table every time (default: a launch of the same geometry on the
 *                         same stream as the one before it reuses the table that is still in place)
 *   HMME_FAIR_PRIO=<0|1>  search kernels: wave priorities that fall with a wave's progress off / on whatever the launch size (default: on
 *                         for whole-CTU workgroups, and for split / strip launches of up to four rounds of workgroups)
 *   HMME_TAIL_PARTS=<n>   the jobs beyond a launch's last full round of workgroups ("tail"): 1 = no tail plan (they run whole, like the others);
 *                         n > 1 = n workgroups per tail job (8-bit: equal segments of the tail's task list; 16-bit: n strips per tail job);
 *                         default = the planner's choice (DESIGN.md 4.1 "tails")
 *   HMME_TAIL_LAUNCHES=<1|2> 8-bit launches with a tail: head and tail in one segment launch / the head's whole jobs, then the tail's segments
 *                         (default: one launch where the tail is at least a third of the head)
 *   HMME_STRIPS16=<n>     16-bit search kernel: that many equal window strips per job instead of the planner's number
 *   HMME_LDS_BUDGET16=<b> 16-bit search kernel: LDS bytes a strip's window rows may take (clamped to what the kernel can address)
 * Measurement and test entry points live in include/hmme_test.h, not here. */

#ifdef __cplusplus
}
#endif
#endif
