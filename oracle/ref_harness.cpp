// ref_harness.cpp -- thin C entry points onto the REFERENCE's own code, compiled from the
// sources where they lie under /root/reference (see oracle/Makefile).  TEST INFRASTRUCTURE:
// it validates oracle/hm_oracle.c and generates tests/golden/*.  Nothing under
// hm-opencl_amd/ links it.  Output goes to oracle/_ref/ only (git-ignored, travels with gpurun).
//
// The reference keeps the functions we need protected/private; the harness opens them with
// the usual test-only macro trick (the reference sources themselves are untouched).
#define protected public
#define private public
#include "TLibCommon/TComRom.h"
#include "TLibCommon/TComRdCost.h"
#include "TLibCommon/TComPattern.h"
#include "TLibCommon/TComDataCU.h"
#include "TLibCommon/TComSlice.h"
#include "TLibEncoder/TEncCfg.h"
#include "TLibEncoder/TEncSearch.h"
#undef protected
#undef private

#include <stdint.h>
#include <string.h>
#include <new>

namespace {

struct Rig {
  TEncSearch* search;
  TEncCfg* cfg;
  TComRdCost* rd;
  TComDataCU* cu;
  TComSlice* slice;
  TComSPS* sps;
  Rig() {
    initROM();
    // heap objects, never destroyed: ~TEncSearch dereferences members init() would have set
    search = new TEncSearch;
    cfg = new TEncCfg;
    rd = new TComRdCost;
    cu = new TComDataCU;
    slice = new TComSlice;
    sps = new TComSPS;
    rd->init();
    search->m_pcEncCfg = cfg;
    search->m_pcRdCost = rd;
    slice->setSPS(sps);
    cu->m_pcSlice = slice;
    cu->m_pePartSize = new Char[4]();
    cu->m_puhDepth = new UChar[4]();
    cu->m_puhWidth = new UChar[4]();
    cu->m_puhHeight = new UChar[4]();
  }
};

Rig& rig() {
  static Rig* r = new Rig;
  return *r;
}

void setup_cost(double lambda, int pred_x, int pred_y, int bit_depth) {
  Rig& r = rig();
  BitDepths bd;
  for (int i = 0; i < MAX_NUM_CHANNEL_TYPE; ++i) bd.recon[i] = bit_depth;
  r.rd->setLambda(lambda, bd);
  r.rd->getMotionCost(true, 0, false);  // TEncSearch.cpp:3735
  TComMv pred((Short)pred_x, (Short)pred_y);
  r.rd->setPredictor(pred);             // :3737
  r.rd->setCostScale(2);                // :3738
}

void setup_cu(int cu_x, int cu_y, int pic_w, int pic_h, int max_cu) {
  Rig& r = rig();
  r.sps->setPicWidthInLumaSamples(pic_w);
  r.sps->setPicHeightInLumaSamples(pic_h);
  r.sps->setMaxCUWidth(max_cu);
  r.sps->setMaxCUHeight(max_cu);
  r.cu->m_uiCUPelX = cu_x;
  r.cu->m_uiCUPelY = cu_y;
}

}  // namespace

extern "C" {

// TComRdCost::xGetComponentBits
uint32_t ref_component_bits(int v) { return rig().rd->xGetComponentBits(v); }

// m_uiLambdaMotionSAD[0] after TComRdCost::setLambda
uint32_t ref_lambda_q16(double lambda) {
  setup_cost(lambda, 0, 0, 8);
  return rig().rd->m_uiCost;
}

// TComRdCost::getCost(x, y) with cost scale 2
uint32_t ref_mv_cost(double lambda, int x, int y, int pred_x, int pred_y) {
  setup_cost(lambda, pred_x, pred_y, 8);
  return rig().rd->getCost(x, y);
}

// the SAD function TComRdCost::setDistParam(pattern, ...) selects for this width
uint32_t ref_sad(int16_t* org, int org_stride, int16_t* cur, int cur_stride, int w, int h, int sub_shift,
                 int bit_depth) {
  Rig& r = rig();
  TComPattern pat;
  pat.initPattern(org, w, h, org_stride, bit_depth);
  DistParam dp;
  dp.bApplyWeight = false;
  r.rd->setDistParam(&pat, cur, cur_stride, dp);
  dp.iSubShift = sub_shift;
  dp.bitDepth = bit_depth;
  return dp.DistFunc(&dp);
}

// TComDataCU::getIndexBlock
int ref_index_block(int part_size, int depth, int part_idx, int abs_z_idx, int cu_h, int cu_w) {
  Rig& r = rig();
  r.cu->m_pePartSize[0] = (Char)part_size;
  r.cu->m_puhDepth[0] = (UChar)depth;
  r.cu->m_absZIdxInCtu = abs_z_idx;
  r.cu->m_puhHeight[0] = (UChar)cu_h;
  r.cu->m_puhWidth[0] = (UChar)cu_w;
  return r.cu->getIndexBlock(part_idx);
}

// TEncSearch::xSetSearchRange (+ TComDataCU::clipMv)
void ref_set_search_range(int pred_x_q, int pred_y_q, int sr, int cu_x, int cu_y, int pic_w, int pic_h,
                          int max_cu, int* lt_x, int* lt_y, int* rb_x, int* rb_y) {
  Rig& r = rig();
  setup_cu(cu_x, cu_y, pic_w, pic_h, max_cu);
  TComMv pred((Short)pred_x_q, (Short)pred_y_q), lt, rb;
  r.search->xSetSearchRange(r.cu, pred, sr, lt, rb);
  *lt_x = lt.getHor(); *lt_y = lt.getVer(); *rb_x = rb.getHor(); *rb_y = rb.getVer();
}

// TEncSearch::xPatternSearch for one PU.  `ref_at_pu` = reference plane at the PU origin.
void ref_pattern_search(int16_t* org, int org_stride, int w, int h, int16_t* ref_at_pu, int ref_stride, int lt_x,
                        int lt_y, int rb_x, int rb_y, int pred_x, int pred_y, double lambda, int fen,
                        int bit_depth, int* mvx, int* mvy, uint32_t* sad) {
  Rig& r = rig();
  setup_cost(lambda, pred_x, pred_y, bit_depth);
  r.cfg->setUseFastEnc(fen != 0);
  r.cfg->setFastSearch(0);
  TComPattern pat;
  pat.initPattern(org, w, h, org_stride, bit_depth);
  TComMv lt((Short)lt_x, (Short)lt_y), rb((Short)rb_x, (Short)rb_y), mv;
  Distortion d = 0;
  r.search->xPatternSearch(&pat, ref_at_pu, ref_stride, &lt, &rb, mv, d);
  *mvx = mv.getHor(); *mvy = mv.getVer(); *sad = d;
}

// the same with explicit weighted prediction switched on for the search (what setWpScalingDistParam, TEncSearch.cpp:5594-5635, leaves
// in m_cDistParam for a slice with weighted prediction): bApplyWeight + the luma WPScalingParam of the reference picture.
// Every candidate's DistFunc is then TComRdCostWeightPrediction::xGetSADw.
void ref_pattern_search_w(int16_t* org, int org_stride, int w, int h, int16_t* ref_at_pu, int ref_stride, int lt_x,
                          int lt_y, int rb_x, int rb_y, int pred_x, int pred_y, double lambda, int fen,
                          int bit_depth, int w0, int offset, int shift, int round, int* mvx, int* mvy, uint32_t* sad) {
  Rig& r = rig();
  setup_cost(lambda, pred_x, pred_y, bit_depth);
  r.cfg->setUseFastEnc(fen != 0);
  r.cfg->setFastSearch(0);
  static WPScalingParam wp[MAX_NUM_COMPONENT];
  memset(wp, 0, sizeof wp);
  wp[COMPONENT_Y].w = w0; wp[COMPONENT_Y].offset = offset; wp[COMPONENT_Y].shift = shift; wp[COMPONENT_Y].round = round;
  r.search->m_cDistParam.bApplyWeight = true;
  r.search->m_cDistParam.wpCur = wp;
  TComPattern pat;
  pat.initPattern(org, w, h, org_stride, bit_depth);
  TComMv lt((Short)lt_x, (Short)lt_y), rb((Short)rb_x, (Short)rb_y), mv;
  Distortion d = 0;
  r.search->xPatternSearch(&pat, ref_at_pu, ref_stride, &lt, &rb, mv, d);
  r.search->m_cDistParam.bApplyWeight = false;
  r.search->m_cDistParam.wpCur = NULL;
  *mvx = mv.getHor(); *mvy = mv.getVer(); *sad = d;
}

// TComRdCostWeightPrediction::xGetSADw through the DistFunc TComRdCost::setDistParam selects (which hands over to it)
uint32_t ref_sad_w(int16_t* org, int org_stride, int16_t* cur, int cur_stride, int w, int h, int sub_shift, int bit_depth, int w0, int offset,
                   int shift, int round) {
  Rig& r = rig();
  TComPattern pat;
  pat.initPattern(org, w, h, org_stride, bit_depth);
  WPScalingParam wp[MAX_NUM_COMPONENT];
  memset(wp, 0, sizeof wp);
  wp[COMPONENT_Y].w = w0; wp[COMPONENT_Y].offset = offset; wp[COMPONENT_Y].shift = shift; wp[COMPONENT_Y].round = round;
  DistParam dp;
  r.rd->setDistParam(&pat, cur, cur_stride, dp);
  dp.bApplyWeight = true;
  dp.wpCur = wp;
  dp.compIdx = COMPONENT_Y;
  dp.iSubShift = sub_shift;     // set as the FEN path sets it: xGetSADw never looks at it
  dp.bitDepth = bit_depth;
  return dp.DistFunc(&dp);
}

// TEncSearch::xTZSearch for one PU (FastSearch=1).  start_q = *pcMvPred (TEncSearch.cpp:3778).
// has_int_mv: pass pIntegerMv2Nx2NPred.  Window LT/RB as computed by the caller.
void ref_tz_search(int16_t* org, int org_stride, int w, int h, int16_t* ref_at_pu, int ref_stride, int lt_x,
                   int lt_y, int rb_x, int rb_y, int pred_x, int pred_y, double lambda, int fen, int bit_depth,
                   int sr, int cu_x, int cu_y, int pic_w, int pic_h, int max_cu, int has_int_mv, int int_mv_x,
                   int int_mv_y, int* mvx, int* mvy, uint32_t* sad) {
  Rig& r = rig();
  setup_cost(lambda, pred_x, pred_y, bit_depth);
  setup_cu(cu_x, cu_y, pic_w, pic_h, max_cu);
  r.cfg->setUseFastEnc(fen != 0);
  r.cfg->setFastSearch(1);
  r.search->m_iSearchRange = sr;
  r.search->m_iFastSearch = 1;
  TComPattern pat;
  pat.initPattern(org, w, h, org_stride, bit_depth);
  TComMv lt((Short)lt_x, (Short)lt_y), rb((Short)rb_x, (Short)rb_y);
  TComMv mv((Short)pred_x, (Short)pred_y);
  TComMv imv((Short)int_mv_x, (Short)int_mv_y);
  Distortion d = 0;
  r.search->xTZSearch(r.cu, &pat, ref_at_pu, ref_stride, &lt, &rb, mv, d, has_int_mv ? &imv : 0);
  *mvx = mv.getHor(); *mvy = mv.getVer(); *sad = d;
}

// The reference's fast search over every PU shape of CTUs [ctu_first, ctu_first + ctu_count) of a picture pair: xTZSearch (what HM runs
// with FastSearch=1, its default) for the 64x64 2Nx2N PU first, then the other 592 shapes seeded with its result like
// m_integerMv2Nx2N (TEncSearch.cpp:3780-3789); predictor (0,0), window from xSetSearchRange.  rects: int32[593][4] (x, y, w, h) in slot
// order.  The CPU baseline bench.py times on the GPU node's host (kind "reference", one core: HM is single-threaded); the same loop
// as oracle/hm_oracle.c tz_worker, whose results must be identical.  Returns the number of PU searches.
long ref_tz_frame(int16_t* cur, int cur_stride, int16_t* ref, int ref_stride, int pic_w, int pic_h, int sr, double lambda, int fen,
                  int bit_depth, int ctu_first, int ctu_count, const int32_t* rects, int32_t* out_x, int32_t* out_y, uint32_t* out_sad) {
  const int ctus_x = (pic_w + 63) / 64;
  long n_searches = 0;
  for (int i = 0; i < ctu_count; ++i) {
    const int ctu = ctu_first + i, cu_x = (ctu % ctus_x) * 64, cu_y = (ctu / ctus_x) * 64;
    int ltx, lty, rbx, rby;
    ref_set_search_range(0, 0, sr, cu_x, cu_y, pic_w, pic_h, 64, &ltx, &lty, &rbx, &rby);
    int imv_x = 0, imv_y = 0;
    for (int n = 0; n < 593; ++n) {
      const int s = n == 0 ? 592 : n - 1;
      const int32_t* r = rects + 4 * s;
      int mx, my;
      uint32_t sad;
      ref_tz_search(cur + (long)(cu_y + r[1]) * cur_stride + cu_x + r[0], cur_stride, r[2], r[3],
                    ref + (long)(cu_y + r[1]) * ref_stride + cu_x + r[0], ref_stride, ltx, lty, rbx, rby, 0, 0, lambda, fen, bit_depth, sr,
                    cu_x, cu_y, pic_w, pic_h, 64, n == 0 ? 0 : 1, imv_x, imv_y, &mx, &my, &sad);
      if (n == 0) { imv_x = mx; imv_y = my; }
      if (out_x) { out_x[(long)i * 593 + s] = mx; out_y[(long)i * 593 + s] = my; out_sad[(long)i * 593 + s] = sad; }
      ++n_searches;
    }
  }
  return n_searches;
}

// TEncSearch::xPatternSearchFracDIF for one PU, set up like xMotionEstimation does (TEncSearch.cpp:3792-3798):
// getMotionCost(true, 0, ...), cost scale 1; the function itself switches to scale 0 for the quarter stage.
static bool g_frac_bipred = false;
void ref_frac_refine(int16_t* org, int org_stride, int w, int h, int16_t* ref_at_pu, int ref_stride, int int_x, int int_y,
                     int pred_x, int pred_y, double lambda, int use_had, int bit_depth, int* half_x, int* half_y,
                     int* qter_x, int* qter_y, uint32_t* cost) {
  Rig& r = rig();
  static bool buffers = false;
  if (!buffers) { r.search->initTempBuff(CHROMA_420); buffers = true; }
  setup_cost(lambda, pred_x, pred_y, bit_depth);
  r.rd->setCostScale(1);
  r.cfg->setUseHADME(use_had != 0);
  TComPattern pat;
  pat.initPattern(org, w, h, org_stride, bit_depth);
  TComMv mv((Short)int_x, (Short)int_y), half, qter;
  Distortion d = 0;
  r.search->xPatternSearchFracDIF(false, &pat, ref_at_pu, ref_stride, &mv, half, qter, d, g_frac_bipred);
  *half_x = half.getHor(); *half_y = half.getVer(); *qter_x = qter.getHor(); *qter_y = qter.getVer(); *cost = d;
}

// xPatternSearchFracDIF in a slice with explicit weighted prediction: m_cDistParam carries bApplyWeight + wpCur from
// setWpScalingDistParam (TEncSearch.cpp:3740) into the refinement (:3798), whose distortion functions are then xGetHADsw / xGetSADw
void ref_frac_refine_w(int16_t* org, int org_stride, int w, int h, int16_t* ref_at_pu, int ref_stride, int int_x, int int_y,
                       int pred_x, int pred_y, double lambda, int use_had, int bit_depth, int w0, int offset, int shift, int round,
                       int* half_x, int* half_y, int* qter_x, int* qter_y, uint32_t* cost) {
  Rig& r = rig();
  static WPScalingParam wp[MAX_NUM_COMPONENT];
  memset(wp, 0, sizeof wp);
  wp[COMPONENT_Y].w = w0; wp[COMPONENT_Y].offset = offset; wp[COMPONENT_Y].shift = shift; wp[COMPONENT_Y].round = round;
  r.search->m_cDistParam.bApplyWeight = true;
  r.search->m_cDistParam.wpCur = wp;
  ref_frac_refine(org, org_stride, w, h, ref_at_pu, ref_stride, int_x, int_y, pred_x, pred_y, lambda, use_had, bit_depth, half_x, half_y,
                  qter_x, qter_y, cost);
  r.search->m_cDistParam.bApplyWeight = false;
  r.search->m_cDistParam.wpCur = NULL;
}

// the bBi call of TEncSearch::xMotionEstimation (TEncSearch.cpp:3798 with bBi = true): `org` is the bi-prediction origin
// 2*org - pred_other (TEncSearch.cpp:3702-3712, TComYuv::removeHighFreq: unclipped, so samples lie in [-maxv, 2*maxv])
void ref_frac_refine_bi(int16_t* org, int org_stride, int w, int h, int16_t* ref_at_pu, int ref_stride, int int_x, int int_y,
                        int pred_x, int pred_y, double lambda, int use_had, int bit_depth, int* half_x, int* half_y,
                        int* qter_x, int* qter_y, uint32_t* cost) {
  g_frac_bipred = true;
  ref_frac_refine(org, org_stride, w, h, ref_at_pu, ref_stride, int_x, int_y, pred_x, pred_y, lambda, use_had, bit_depth, half_x, half_y,
                  qter_x, qter_y, cost);
  g_frac_bipred = false;
}

// TComPicYuv::create + extendPicBorder (TComPicYuv.cpp:80-133, :214-262): a w x h luma picture goes in, the whole padded buffer
// (margin maxCU + 16 on every side) comes out.  out must hold (*out_stride) * (h + 2 * margin) samples; returns the margin.
int ref_extend_border(const int16_t* img, int img_stride, int w, int h, int max_cu, int16_t* out, int out_capacity, int* out_stride) {
  initROM();
  TComPicYuv pic;
  pic.create(w, h, CHROMA_400, max_cu, max_cu, 4, true);
  const int stride = pic.getStride(COMPONENT_Y), total_h = pic.getTotalHeight(COMPONENT_Y), margin = pic.getMarginX(COMPONENT_Y);
  *out_stride = stride;
  if (stride * total_h > out_capacity) { pic.destroy(); return -1; }
  Pel* org = pic.getAddr(COMPONENT_Y);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) org[y * stride + x] = img[y * img_stride + x];
  pic.extendPicBorder();
  const Pel* buf = pic.getBuf(COMPONENT_Y);
  for (int i = 0; i < stride * total_h; ++i) out[i] = buf[i];
  pic.destroy();
  return margin;
}

}  // extern "C"
