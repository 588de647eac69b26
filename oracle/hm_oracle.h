/*
 * hm_oracle.h -- CPU restatement of the HM 16.4 integer motion-estimation path that the
 * reference's TEncOpenCL add-on accelerates.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link, load or call it.  The shipped engine
 * (hm-opencl_amd/csrc) never does; it fails loudly when the HIP library is missing.
 *
 * Parity is PINNED: every function here is checked (tests/test_oracle_vs_ref.py) against
 *   (a) oracle/_ref/libhmref.so -- the reference's own TEncSearch::xPatternSearch /
 *       xTZSearch / TComRdCost / TComDataCU::getIndexBlock compiled from /root/reference, and
 *   (b) the golden vectors under tests/golden/ that (a) generated (tests/golden/gen_golden.py).
 *
 * All file:line citations are relative to /root/reference/source/Lib/.
 */
#ifndef HM_ORACLE_H
#define HM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMO_NUM_CTU_PARTS 593 /* TLibCommon/TypeDef.h:263 */
#define HMO_CTU 64

typedef int16_t hmo_pel; /* TLibCommon/TypeDef.h:706  (Pel = Short) */

typedef struct hmo_rect { int x, y, w, h; } hmo_rect;

/* search parameters of one (CTU, reference picture) search */
typedef struct hmo_params {
  int lt_x, lt_y;       /* integer-pel window, top-left  (TEncSearch.cpp:3814-3830) */
  int rb_x, rb_y;       /* integer-pel window, bottom-right (inclusive) */
  int pred_x, pred_y;   /* AMVP predictor in quarter pels (TComRdCost::setPredictor, TEncSearch.cpp:3737) */
  uint32_t lambda_q16;  /* m_uiCost = floor(65536*sqrt(lambda)) (TComRdCost.cpp:209, TEncOpenCL.h:121) */
  int fen;              /* getUseFastEnc(): rows > 8 use every 2nd row (TEncSearch.cpp:3853-3859) */
  int bit_depth;        /* SAD >> (bitDepth-8)  (TComRdCost.cpp:520-521) */
} hmo_params;

/* explicit weighted prediction of the reference picture (WPScalingParam of the luma component, TComSlice.h:1178-1190): the
 * integer search of a slice with weighted prediction prices |org - (((w0 * ref + round) >> shift) + offset)| (xGetSADw) */
typedef struct hmo_wp { int w0, offset, shift, round; } hmo_wp;

/* ---- MV-bit cost ------------------------------------------------------------------- */
uint32_t hmo_component_bits(int val);                        /* TComRdCost.cpp:278-292 */
uint32_t hmo_lambda_q16(double lambda);                      /* TComRdCost.cpp:209 */
/* getCost(x,y) with cost scale `scale` (2 during integer ME): TComRdCost.h:172-189 */
uint32_t hmo_mv_cost(uint32_t lambda_q16, int x, int y, int pred_x, int pred_y, int scale);

/* ---- SAD (xGetSAD4..64 / 12 / 24 / 48 all reduce to this): TComRdCost.cpp:465-964 -- */
uint32_t hmo_sad(const hmo_pel* org, int org_stride, const hmo_pel* cur, int cur_stride,
                 int w, int h, int sub_shift, int bit_depth);

/* ---- slot layout: TComDataCU.cpp:3379-3391 + case table :4676-6461 ----------------- */
/* rectangle (inside the 64x64 CTU) of slot 0..592 */
int hmo_slot_rect(int slot, hmo_rect* r);
/* the key getIndexBlock() switches on, and the slot it maps to (-1 when not tabulated) */
int32_t hmo_index_key(int part_size, int depth, int part_idx, int abs_z_idx, int cu_h, int cu_w);
int hmo_index_block(int part_size, int depth, int part_idx, int abs_z_idx, int cu_size);

/* ---- search window: TEncSearch.cpp:3814-3830 + TComDataCU::clipMv :2907-2920 ------- */
void hmo_clip_mv(int* mvx_q, int* mvy_q, int cu_x, int cu_y, int pic_w, int pic_h, int max_cu);
void hmo_set_search_range(int pred_x_q, int pred_y_q, int sr, int cu_x, int cu_y, int pic_w,
                          int pic_h, int max_cu, int* lt_x, int* lt_y, int* rb_x, int* rb_y);

/* ---- exhaustive search for ONE PU: literal restatement of xPatternSearch
 *      (TEncSearch.cpp:3835-3897).  `ref` points at the PU's own origin in the reference
 *      plane (displacement 0). */
void hmo_pattern_search(const hmo_pel* org, int org_stride, int w, int h, const hmo_pel* ref,
                        int ref_stride, const hmo_params* p, int* mvx, int* mvy, uint32_t* sad);

/* ---- exhaustive search for all 593 slots of one CTU (what calcMotionVectors returns,
 *      TEncOpenCL.cpp:240-362, with the CPU's arithmetic -- SURVEY 8a quirk 4 defines parity
 *      for slots != 592 as "xPatternSearch on that rectangle with the CTU's window and
 *      predictor").  `ctu` is the 64x64 current block, `ref` the reference plane at the CTU
 *      origin.  Fast formulation (4x4 block sums + integral image); bit-identical to calling
 *      hmo_pattern_search per slot (tests/test_oracle.py checks that). */
void hmo_search_ctu(const hmo_pel* ctu, int ctu_stride, const hmo_pel* ref, int ref_stride,
                    const hmo_params* p, int32_t* out_x, int32_t* out_y, uint32_t* out_sad,
                    uint32_t* out_cost);

/* reference-GPU-compatible preset: pred=(0,0), window LT..LT+2*sr,
 * no FEN, no bit-depth shift */
void hmo_ocl_compat_params(hmo_params* p, int lt_x, int lt_y, int sr, uint32_t lambda_q16);

/* ---- the same searches in a slice with explicit weighted prediction (m_cDistParam.bApplyWeight, TEncSearch.cpp:3740, :5594-5635) ---- */
/* TComRdCostWeightPrediction::xGetSADw (TComRdCostWeightPrediction.cpp:55-90): every row (the FEN sub-sampling is never reached:
 * each xGetSAD* returns xGetSADw first, TComRdCost.cpp:467-469), prediction not clipped, whole-block sum >> (bitDepth-8) */
uint32_t hmo_sad_w(const hmo_pel* org, int org_stride, const hmo_pel* cur, int cur_stride, int w, int h, int bit_depth, const hmo_wp* wp);
/* xPatternSearch with bApplyWeight: p->fen is ignored (see hmo_sad_w) */
void hmo_pattern_search_w(const hmo_pel* org, int org_stride, int w, int h, const hmo_pel* ref, int ref_stride, const hmo_params* p,
                          const hmo_wp* wp, int* mvx, int* mvy, uint32_t* sad);
/* xPatternSearchFracDIF in a slice with weighted prediction (xGetHADsw / xGetSADw on the weighted interpolated prediction) */
void hmo_frac_refine_w(const hmo_pel* org, int org_stride, int w, int h, const hmo_pel* ref, int ref_stride, int int_x, int int_y, int pred_x,
                       int pred_y, uint32_t lambda_q16, int use_had, int bit_depth, const hmo_wp* wp, int* half_x, int* half_y, int* qter_x,
                       int* qter_y, uint32_t* cost);
void hmo_search_ctu_w(const hmo_pel* ctu, int ctu_stride, const hmo_pel* ref, int ref_stride, const hmo_params* p, const hmo_wp* wp,
                      int32_t* out_x, int32_t* out_y, uint32_t* out_sad, uint32_t* out_cost);

/* ---- TZ search for one PU: xTZSearch + helpers (TEncSearch.cpp:3935-4136, :340-808,
 *      configuration :305-321).  Returns the number of SAD probes made.
 *      int_mv_2nx2n: NULL or pointer to {x,y} integer MV predictor (pIntegerMv2Nx2NPred). */
typedef struct hmo_tz_ctx {
  int sr;                 /* m_iSearchRange */
  int cu_x, cu_y;         /* CU position in the picture (for clipMv) */
  int pic_w, pic_h, max_cu;
} hmo_tz_ctx;
long hmo_tz_search(const hmo_pel* org, int org_stride, int w, int h, const hmo_pel* ref,
                   int ref_stride, const hmo_params* p, const hmo_tz_ctx* tz,
                   const int* int_mv_2nx2n, int start_x_q, int start_y_q, int* mvx, int* mvy,
                   uint32_t* sad);

/* ---- frame helpers (the steps either side of the path) ----------------------------- */
/* TComPicYuv::extendPicBorder (TComPicYuv.cpp:214-262): edge-replicate `margin` samples */
void hmo_extend_border(hmo_pel* pic_origin, int stride, int w, int h, int margin_x, int margin_y);

/* whole-frame exhaustive search: every CTU (partial edge CTUs included, computed on the
 * padded plane), per-CTU predictors, window from hmo_set_search_range.
 * cur / ref point at sample (0,0) of padded planes.  out arrays are [n_ctu][593].
 * n_threads <= 0 -> 1.  Returns number of CTUs. */
int hmo_search_frame(const hmo_pel* cur, int cur_stride, const hmo_pel* ref, int ref_stride,
                     int pic_w, int pic_h, int sr, const int16_t* pred_q /* [n_ctu][2] or NULL */,
                     uint32_t lambda_q16, int fen, int bit_depth, int ctu_first, int ctu_count,
                     int n_threads, int32_t* out_x, int32_t* out_y, uint32_t* out_sad);

/* ---- fractional-pel refinement (the step after the integer search) -------------
 * canonical luma prediction block at a quarter-pel displacement (what TEncSearch's m_filteredBlock tables
 * hold), xGetHADs, and xPatternSearchFracDIF for one PU */
void hmo_pred_block_qpel(const hmo_pel* ref, int ref_stride, int w, int h, int qx, int qy, int bit_depth,
                         hmo_pel* dst, int dst_stride);
uint32_t hmo_had(const hmo_pel* org, int org_stride, const hmo_pel* cur, int cur_stride, int w, int h, int bit_depth);
void hmo_frac_refine(const hmo_pel* org, int org_stride, int w, int h, const hmo_pel* ref, int ref_stride, int int_x,
                     int int_y, int pred_x, int pred_y, uint32_t lambda_q16, int use_had, int bit_depth, int* half_x,
                     int* half_y, int* qter_x, int* qter_y, uint32_t* cost);

/* hmo_frac_refine for every slot of every CTU of a picture range, threaded over CTUs (the checker of hmme_refine_frame* at full
 * picture sizes).  int_mv: [ctu_count][593][2]; out_qmv: [ctu_count][593][2] quarter-pel MVs; out_cost: [ctu_count][593] */
int hmo_refine_frame(const hmo_pel* cur, int cur_stride, const hmo_pel* ref, int ref_stride, int pic_w, int pic_h, const int16_t* pred_q,
                     uint32_t lambda_q16, int use_had, int bit_depth, int ctu_first, int ctu_count, int n_threads, const int16_t* int_mv,
                     int16_t* out_qmv, uint32_t* out_cost);

/* CPU baseline leg of bench.py: xTZSearch for the 64x64 PU (all_slots=0) or for all 593 PU
 * rectangles (all_slots=1) of every CTU in [ctu_first, ctu_first+ctu_count), threaded over CTUs.
 * probes = SAD evaluations; sad4x4 = the same work in 4x4-block-SAD equivalents (w*h/16 per probe,
 * halved where FEN sub-sampling applies).  out_* may be NULL. */
int hmo_tz_frame(const hmo_pel* cur, int cur_stride, const hmo_pel* ref, int ref_stride, int pic_w, int pic_h,
                 int sr, const int16_t* pred_q, uint32_t lambda_q16, int fen, int bit_depth, int ctu_first,
                 int ctu_count, int n_threads, int all_slots, int32_t* out_x, int32_t* out_y, uint32_t* out_sad,
                 long* probes, double* sad4x4);

#ifdef __cplusplus
}
#endif
#endif
