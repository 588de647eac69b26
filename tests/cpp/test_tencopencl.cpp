// Drives the TEncOpenCL-shaped host module the way TEncTop::xInitOpenCL and
// TEncSearch::xMotionEstimation do (reference TEncTop.cpp:1129-1145, TEncSearch.cpp:3743-3765) and
// checks the 593 results against the CPU oracle (test infrastructure).  Exit code 0 = all equal.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../hm-opencl_amd/host/TEncOpenCL.h"
#include "../../oracle/hm_oracle.h"

static unsigned rng_state = 12345u;
static unsigned rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main() {
  const int SR = 8, M = 80, W = 192, H = 128, stride = W + 2 * M;
  std::vector<Pel> cur((H + 2 * M) * stride), ref((H + 2 * M) * stride);
  for (size_t i = 0; i < cur.size(); ++i) { cur[i] = (Pel)(rnd() & 255); ref[i] = (Pel)(rnd() & 255); }

  TEncOpenCL me;
  if (!me.findDevice(0)) { fprintf(stderr, "findDevice failed\n"); return 2; }
  if (!me.compileKernelSource("cl/sad.cl", "calcSAD_AMP")) { fprintf(stderr, "compileKernelSource failed\n"); return 2; }
  if (!me.createBuffers(64, 64, SR)) { fprintf(stderr, "createBuffers failed\n"); return 2; }
  me.setEnabled(true);
  printf("device: %s\n", me.getDeviceInfo());
  const double lambda = 57.9;
  me.setLambda(lambda);

  // these stand for TEncSearch::allMotionVectors / allRuiCost [list][refIdx] (TEncSearch.h:114-115)
  static TComMv allMotionVectors[NUM_CTU_PARTS];
  static Distortion allRuiCost[NUM_CTU_PARTS];
  int failures = 0;
  for (int mode = 0; mode < 2; ++mode) {
    for (int ctu = 0; ctu < 6; ++ctu) {
      const int cu_x = (ctu % 3) * 64, cu_y = (ctu / 3) * 64;
      Pel* piCtu = &cur[(M + cu_y) * stride + M + cu_x];
      Pel* piRefY = &ref[(M + cu_y) * stride + M + cu_x];
      const int pred_x = mode ? (int)(rnd() % 41) - 20 : 0, pred_y = mode ? (int)(rnd() % 41) - 20 : 0;
      int ltx, lty, rbx, rby;
      hmo_set_search_range(pred_x, pred_y, SR, cu_x, cu_y, W, H, 64, &ltx, &lty, &rbx, &rby);
      TComMv lt((Short)ltx, (Short)lty), rb((Short)rbx, (Short)rby);
      hmo_params p;
      if (mode == 0) {   // exactly the reference call sequence: no additive setters
        me.setCostMode(TEncOpenCL::ME_MODE_OCL_COMPAT);
        hmo_ocl_compat_params(&p, ltx, lty, SR, hmo_lambda_q16(lambda));
      } else {
        me.setCostMode(TEncOpenCL::ME_MODE_HM);
        me.setPredictor(TComMv((Short)pred_x, (Short)pred_y));
        me.setSearchRangeRB(rb);
        me.setFastEnc(true);
        p.lt_x = ltx; p.lt_y = lty; p.rb_x = rbx; p.rb_y = rby; p.pred_x = pred_x; p.pred_y = pred_y;
        p.lambda_q16 = hmo_lambda_q16(lambda); p.fen = 1; p.bit_depth = 8;
      }
      me.calcMotionVectors(piCtu, piRefY, stride, stride, SR, &lt);
      if (!me.lastCallOk()) { fprintf(stderr, "calcMotionVectors failed\n"); return 3; }
      Int* xTemp = me.getX();
      Int* yTemp = me.getY();
      Distortion* ruiCostTemp = me.getRuiCost();
      for (int i = 0; i < NUM_CTU_PARTS; i++) {   // TEncSearch.cpp:3760-3764
        allRuiCost[i] = ruiCostTemp[i];
        allMotionVectors[i].set((Short)xTemp[i], (Short)yTemp[i]);
      }
      if (std::memcmp(allMotionVectors, me.getMvs(), sizeof allMotionVectors) != 0) { fprintf(stderr, "getMvs layout mismatch\n"); ++failures; }
      int32_t ox[HMO_NUM_CTU_PARTS], oy[HMO_NUM_CTU_PARTS];
      uint32_t osad[HMO_NUM_CTU_PARTS];
      hmo_search_ctu(piCtu, stride, piRefY, stride, &p, ox, oy, osad, NULL);
      for (int i = 0; i < NUM_CTU_PARTS; i++)
        if (allMotionVectors[i].getHor() != ox[i] || allMotionVectors[i].getVer() != oy[i] || allRuiCost[i] != osad[i]) {
          if (failures < 10)
            fprintf(stderr, "mode %d ctu %d slot %d: got (%d,%d,%u) want (%d,%d,%u)\n", mode, ctu, i, allMotionVectors[i].getHor(),
                    allMotionVectors[i].getVer(), allRuiCost[i], ox[i], oy[i], osad[i]);
          ++failures;
        }
    }
  }
  // ---- bi-prediction tables: a bBi call (TEncSearch.cpp:3221: origin 2*org - pred_other, unclipped,
  // TComYuv.cpp:409-440; SearchRange = BipredSearchRange = 4) must leave the uni-prediction tables alone
  {
    const int cu_x = 64, cu_y = 0, BSR = 4;
    Pel* piCtu = &cur[(M + cu_y) * stride + M + cu_x];
    Pel* piRefY = &ref[(M + cu_y) * stride + M + cu_x];
    me.setCostMode(TEncOpenCL::ME_MODE_HM);
    me.setFastEnc(true);
    int ltx, lty, rbx, rby;
    hmo_set_search_range(12, -8, SR, cu_x, cu_y, W, H, 64, &ltx, &lty, &rbx, &rby);
    TComMv lt((Short)ltx, (Short)lty);
    me.setPredictor(TComMv(12, -8));
    me.setSearchRangeRB(TComMv((Short)rbx, (Short)rby));
    me.calcMotionVectors(piCtu, piRefY, stride, stride, SR, &lt);
    static TComMv uniMv[NUM_CTU_PARTS];
    static Distortion uniCost[NUM_CTU_PARTS];
    std::memcpy(uniMv, me.getMvs(), sizeof uniMv);
    std::memcpy(uniCost, me.getRuiCost(), sizeof uniCost);
    std::vector<Pel> bi(64 * 64);
    for (int y = 0; y < 64; ++y)
      for (int x = 0; x < 64; ++x) bi[y * 64 + x] = (Pel)(2 * piCtu[y * stride + x] - (Pel)(rnd() & 255));   // in [-255, 510]
    const int cx = uniMv[592].getHor() << 2, cy = uniMv[592].getVer() << 2;   // xSetSearchRange(pcCU, rcMv, ...), TEncSearch.cpp:3725
    hmo_set_search_range(cx, cy, BSR, cu_x, cu_y, W, H, 64, &ltx, &lty, &rbx, &rby);
    TComMv ltb((Short)ltx, (Short)lty);
    me.setBiPred(true);
    me.setPredictor(TComMv(-3, 5));
    me.setSearchRangeRB(TComMv((Short)rbx, (Short)rby));
    me.calcMotionVectors(&bi[0], piRefY, stride, 64, BSR, &ltb);
    if (!me.lastCallOk()) { fprintf(stderr, "bi-pred call failed\n"); ++failures; }
    hmo_params p;
    p.lt_x = ltx; p.lt_y = lty; p.rb_x = rbx; p.rb_y = rby; p.pred_x = -3; p.pred_y = 5;
    p.lambda_q16 = hmo_lambda_q16(lambda); p.fen = 1; p.bit_depth = 8;
    int32_t ox[HMO_NUM_CTU_PARTS], oy[HMO_NUM_CTU_PARTS];
    uint32_t osad[HMO_NUM_CTU_PARTS];
    hmo_search_ctu(&bi[0], 64, piRefY, stride, &p, ox, oy, osad, NULL);
    for (int i = 0; i < NUM_CTU_PARTS; i++)
      if (me.getMvs()[i].getHor() != ox[i] || me.getMvs()[i].getVer() != oy[i] || me.getRuiCost()[i] != osad[i] || me.getX()[i] != ox[i]) {
        if (failures < 10) fprintf(stderr, "bi-pred slot %d differs from the oracle\n", i);
        ++failures;
      }
    // the bBi pass also refines that origin (xPatternSearchFracDIF(..., bBi), TEncSearch.cpp:3798): with setRefine the same call
    // fills the refinement tables, flagged as bi-prediction tables
    me.setRefine(true, true);
    me.calcMotionVectors(&bi[0], piRefY, stride, 64, BSR, &ltb);
    if (!me.lastCallOk() || !me.fracOk() || !me.fracWasBi()) { fprintf(stderr, "bi-pred search + refine call failed\n"); ++failures; }
    for (int slot = 0; slot < NUM_CTU_PARTS; slot += 37) {
      hmo_rect r;
      hmo_slot_rect(slot, &r);
      const TComMv imv = me.getMvs()[slot];
      if (imv.getHor() != ox[slot] || imv.getVer() != oy[slot]) { fprintf(stderr, "bi-pred search+refine: integer slot %d differs\n", slot); ++failures; }
      int hx, hy, qx, qy;
      uint32_t cost;
      hmo_frac_refine(&bi[r.y * 64 + r.x], 64, r.w, r.h, piRefY + r.y * stride + r.x, stride, imv.getHor(), imv.getVer(), -3, 5, hmo_lambda_q16(lambda), 1, 8,
                      &hx, &hy, &qx, &qy, &cost);
      const int wx = 4 * imv.getHor() + 2 * hx + qx, wy = 4 * imv.getVer() + 2 * hy + qy;
      if (me.getQMvs()[slot].getHor() != wx || me.getQMvs()[slot].getVer() != wy || me.getFracCost()[slot] != cost) {
        if (failures < 10) fprintf(stderr, "bi-pred refinement slot %d: got (%d,%d,%u) want (%d,%d,%u)\n", slot, me.getQMvs()[slot].getHor(),
                                   me.getQMvs()[slot].getVer(), me.getFracCost()[slot], wx, wy, cost);
        ++failures;
      }
    }
    me.setRefine(false);
    me.setBiPred(false);
    if (std::memcmp(uniMv, me.getMvs(), sizeof uniMv) != 0 || std::memcmp(uniCost, me.getRuiCost(), sizeof uniCost) != 0 ||
        std::memcmp(uniMv, me.getMvs(false), sizeof uniMv) != 0) {
      fprintf(stderr, "the bi-prediction call clobbered the uni-prediction tables\n");
      ++failures;
    }
    if (std::memcmp(me.getMvs(true), me.getMvs(false), sizeof uniMv) == 0) { fprintf(stderr, "bi tables equal uni tables?\n"); ++failures; }
  }
  // ---- refinement tables: setRefine makes a uni-prediction ME_MODE_HM call also run xPatternSearchFracDIF for
  // the 593 slots; storeFrac keeps them per [list][refIdx] with the MV cost taken out
  {
    const int cu_x = 64, cu_y = 64;
    Pel* piCtu = &cur[(M + cu_y) * stride + M + cu_x];
    Pel* piRefY = &ref[(M + cu_y) * stride + M + cu_x];
    me.setCostMode(TEncOpenCL::ME_MODE_HM);
    me.setFastEnc(true);
    me.setRefine(true, true);
    int ltx, lty, rbx, rby;
    hmo_set_search_range(-9, 14, SR, cu_x, cu_y, W, H, 64, &ltx, &lty, &rbx, &rby);
    TComMv lt((Short)ltx, (Short)lty);
    me.setPredictor(TComMv(-9, 14));
    me.setSearchRangeRB(TComMv((Short)rbx, (Short)rby));
    me.calcMotionVectors(piCtu, piRefY, stride, stride, SR, &lt);
    if (!me.lastCallOk() || !me.fracOk()) { fprintf(stderr, "search + refine call failed\n"); ++failures; }
    me.markTables(1, 2, 40, 9);
    me.storeFrac(1, 2);
    const bool stored = me.fracStored(1, 2, 40, 9);
    if (!stored || me.fracStored(1, 2, 40, 8) || me.fracStored(0, 2, 40, 9)) { fprintf(stderr, "frac table tags\n"); ++failures; }
    // tables nobody stored read as poison (never as unallocated memory): [0][2] was not stored, and neither is anything out of range
    if (me.getFracCostStored(0, 2, 5) != 0xFFFFFFFFu || me.getFracDist(0, 2, 5) != 0xFFFFFFFFu || me.getFracMv(0, 2, 5).getHor() != 0 ||
        me.getFracCostStored(1, 2, 593) != 0xFFFFFFFFu || me.getFracCostStored(2, 2, 0) != 0xFFFFFFFFu) {
      fprintf(stderr, "unstored refinement tables do not read as poison\n"); ++failures;
    }
    const uint32_t lq = hmo_lambda_q16(lambda);
    for (int slot = 0; stored && slot < NUM_CTU_PARTS; slot += 16) {   // the stored tables are read only when fracStored() says so
      hmo_rect r;
      hmo_slot_rect(slot, &r);
      const TComMv imv = me.getMvs()[slot];
      int hx, hy, qx, qy;
      uint32_t cost;
      hmo_frac_refine(piCtu + r.y * stride + r.x, stride, r.w, r.h, piRefY + r.y * stride + r.x, stride, imv.getHor(), imv.getVer(), -9, 14, lq, 1, 8,
                      &hx, &hy, &qx, &qy, &cost);
      const int wx = 4 * imv.getHor() + 2 * hx + qx, wy = 4 * imv.getVer() + 2 * hy + qy;
      if (me.getQMvs()[slot].getHor() != wx || me.getQMvs()[slot].getVer() != wy || me.getFracCost()[slot] != cost) {
        if (failures < 10) fprintf(stderr, "refinement slot %d: got (%d,%d,%u) want (%d,%d,%u)\n", slot, me.getQMvs()[slot].getHor(), me.getQMvs()[slot].getVer(),
                                   me.getFracCost()[slot], wx, wy, cost);
        ++failures;
      }
      const uint32_t mvc = hmo_mv_cost(lq, wx, wy, -9, 14, 0);
      if (me.getFracDist(1, 2, slot) + mvc != cost || me.getFracCostStored(1, 2, slot) != cost || me.getFracMv(1, 2, slot).getHor() != wx) {
        fprintf(stderr, "stored refinement slot %d\n", slot); ++failures;
      }
    }
    me.setRefine(false);
  }
  // ---- explicit weighted prediction: setWeight -> hmme_search_ctu_w == the oracle's weighted search (itself pinned by the reference's
  // xPatternSearch with bApplyWeight, tests/golden/wp.npz); refinement tables are not produced for weighted calls
  {
    const int cu_x = 64, cu_y = 0;
    std::vector<Pel> faded(64 * 64);
    Pel* piRefY = &ref[(M + cu_y) * stride + M + cu_x];
    for (int y = 0; y < 64; ++y)
      for (int x = 0; x < 64; ++x) {
        const int v = ((52 * piRefY[(y + 2) * stride + x - 3] + 32) >> 6) + 17 + (int)(rnd() % 3) - 1;
        faded[y * 64 + x] = (Pel)(v < 0 ? 0 : (v > 255 ? 255 : v));
      }
    me.setCostMode(TEncOpenCL::ME_MODE_HM);
    me.setFastEnc(true);
    me.setRefine(true, true);
    me.setWeight(52, 17, 6, 32);
    int ltx, lty, rbx, rby;
    hmo_set_search_range(5, -3, SR, cu_x, cu_y, W, H, 64, &ltx, &lty, &rbx, &rby);
    TComMv lt((Short)ltx, (Short)lty);
    me.setPredictor(TComMv(5, -3));
    me.setSearchRangeRB(TComMv((Short)rbx, (Short)rby));
    me.calcMotionVectors(&faded[0], piRefY, stride, 64, SR, &lt);
    if (!me.lastCallOk() || !me.fracOk()) { fprintf(stderr, "weighted call: ok %d, fracOk %d\n", (int)me.lastCallOk(), (int)me.fracOk()); ++failures; }
    hmo_params p;
    p.lt_x = ltx; p.lt_y = lty; p.rb_x = rbx; p.rb_y = rby; p.pred_x = 5; p.pred_y = -3; p.lambda_q16 = hmo_lambda_q16(lambda); p.fen = 1; p.bit_depth = 8;
    const hmo_wp wp = {52, 17, 6, 32};
    int32_t ox[HMO_NUM_CTU_PARTS], oy[HMO_NUM_CTU_PARTS];
    uint32_t osad[HMO_NUM_CTU_PARTS];
    hmo_search_ctu_w(&faded[0], 64, piRefY, stride, &p, &wp, ox, oy, osad, NULL);
    int bad = 0;
    for (int i = 0; i < NUM_CTU_PARTS; i++) bad += me.getMvs()[i].getHor() != ox[i] || me.getMvs()[i].getVer() != oy[i] || me.getRuiCost()[i] != osad[i];
    if (bad) { fprintf(stderr, "weighted prediction: %d slots differ from the oracle\n", bad); failures += bad; }
    if (ox[592] != -3 || oy[592] != 2) { fprintf(stderr, "weighted prediction: the fade's displacement was not found (%d,%d)\n", ox[592], oy[592]); ++failures; }
    for (int slot = 0; slot < NUM_CTU_PARTS; slot += 13) {   // the refinement tables of the same call: xGetHADsw on the weighted interpolated prediction
      hmo_rect r;
      hmo_slot_rect(slot, &r);
      int hx, hy, qx, qy;
      uint32_t cost;
      hmo_frac_refine_w(&faded[r.y * 64 + r.x], 64, r.w, r.h, piRefY + r.y * stride + r.x, stride, ox[slot], oy[slot], 5, -3, p.lambda_q16, 1, 8, &wp,
                        &hx, &hy, &qx, &qy, &cost);
      if (me.getQMvs()[slot].getHor() != 4 * ox[slot] + 2 * hx + qx || me.getQMvs()[slot].getVer() != 4 * oy[slot] + 2 * hy + qy || me.getFracCost()[slot] != cost) {
        if (failures < 10) fprintf(stderr, "weighted refinement slot %d differs from the oracle\n", slot);
        ++failures;
      }
    }
    me.clearWeight();
    me.setRefine(false);
  }
  // ---- picture-edge CTU: 192x128 has none, so pretend the picture ends inside CTU (2,1): 40 x 24 valid
  {
    const int cu_x = 128, cu_y = 64, vw = 40, vh = 24, pw = cu_x + vw, ph = cu_y + vh;
    Pel* piCtu = &cur[(M + cu_y) * stride + M + cu_x];
    Pel* piRefY = &ref[(M + cu_y) * stride + M + cu_x];
    me.calcMotionVectorsEdge(piCtu, stride, vw, vh, piRefY, stride, SR, TComMv(-6, 9), cu_x, cu_y, pw, ph);
    if (!me.lastCallOk()) { fprintf(stderr, "edge call failed\n"); ++failures; }
    std::vector<Pel> blk(64 * 64);
    for (int y = 0; y < 64; ++y)
      for (int x = 0; x < 64; ++x) blk[y * 64 + x] = piCtu[(y < vh ? y : vh - 1) * stride + (x < vw ? x : vw - 1)];
    hmo_params p;
    hmo_set_search_range(-6, 9, SR, cu_x, cu_y, pw, ph, 64, &p.lt_x, &p.lt_y, &p.rb_x, &p.rb_y);
    p.pred_x = -6; p.pred_y = 9; p.lambda_q16 = hmo_lambda_q16(lambda); p.fen = 1; p.bit_depth = 8;
    int32_t ox[HMO_NUM_CTU_PARTS], oy[HMO_NUM_CTU_PARTS];
    uint32_t osad[HMO_NUM_CTU_PARTS];
    hmo_search_ctu(&blk[0], 64, piRefY, stride, &p, ox, oy, osad, NULL);
    for (int i = 0; i < NUM_CTU_PARTS; i++)
      if (me.getMvs()[i].getHor() != ox[i] || me.getMvs()[i].getVer() != oy[i] || me.getRuiCost()[i] != osad[i]) {
        if (failures < 10) fprintf(stderr, "edge CTU slot %d differs from the oracle\n", i);
        ++failures;
      }
    if (me.tablesValidFor(0, 1, 7, 5)) { fprintf(stderr, "tables valid before they were marked\n"); ++failures; }
    me.markTables(0, 1, 7, 5);
    if (!me.tablesValidFor(0, 1, 7, 5) || me.tablesValidFor(0, 1, 7, 6) || me.tablesValidFor(0, 1, 8, 5) || me.tablesValidFor(1, 1, 7, 5)) {
      fprintf(stderr, "table tags\n"); ++failures;
    }
  }
  // ---- unmodified reference call sites on 10-bit content: nothing tells the class the bit depth; it takes the sample width
  // from the window and, like cl/sad.cl, does not shift the SAD (the oracle's compat preset: bit depth 8 = no shift)
  {
    std::vector<Pel> c10(cur.size()), r10(ref.size());
    for (size_t i = 0; i < c10.size(); ++i) { c10[i] = (Pel)(rnd() & 1023); r10[i] = (Pel)(rnd() & 1023); }
    TEncOpenCL me10;
    if (!me10.findDevice(0) || !me10.compileKernelSource("cl/sad.cl", "calcSAD_AMP") || !me10.createBuffers(64, 64, SR)) return 2;
    me10.setLambda(lambda);
    const int cu_x = 64, cu_y = 64;
    TComMv lt((Short)-SR, (Short)(-SR + 1));
    me10.calcMotionVectors(&c10[(M + cu_y) * stride + M + cu_x], &r10[(M + cu_y) * stride + M + cu_x], stride, stride, SR, &lt);
    if (!me10.lastCallOk()) { fprintf(stderr, "10-bit compat call failed\n"); ++failures; }
    hmo_params p;
    hmo_ocl_compat_params(&p, -SR, -SR + 1, SR, hmo_lambda_q16(lambda));
    int32_t ox[HMO_NUM_CTU_PARTS], oy[HMO_NUM_CTU_PARTS];
    uint32_t osad[HMO_NUM_CTU_PARTS];
    hmo_search_ctu(&c10[(M + cu_y) * stride + M + cu_x], stride, &r10[(M + cu_y) * stride + M + cu_x], stride, &p, ox, oy, osad, NULL);
    int bad = 0;
    for (int i = 0; i < NUM_CTU_PARTS; i++)
      bad += me10.getX()[i] != ox[i] || me10.getY()[i] != oy[i] || me10.getRuiCost()[i] != osad[i];
    if (bad) { fprintf(stderr, "10-bit compat mode: %d slots differ from the oracle\n", bad); failures += bad; }
    if (osad[592] < 300000) { fprintf(stderr, "10-bit sums look shifted (%u)\n", osad[592]); ++failures; }
    // a dark reference window (max <= 255) under a bright 10-bit block (> 2 * 255), as on a fade or a scene cut, on a FRESH object:
    // the width must come from the block too, or the call is rejected as out of range and the encoder gets poisoned tables
    TEncOpenCL meDark;
    if (!meDark.findDevice(0) || !meDark.compileKernelSource("cl/sad.cl", "calcSAD_AMP") || !meDark.createBuffers(64, 64, SR)) return 2;
    meDark.setLambda(lambda);
    std::vector<Pel> dark(ref.size());
    for (size_t i = 0; i < dark.size(); ++i) dark[i] = (Pel)(rnd() & 127);
    Pel* cb = &c10[(M + cu_y) * stride + M + cu_x];
    cb[5 * stride + 7] = 1023;
    meDark.calcMotionVectors(cb, &dark[(M + cu_y) * stride + M + cu_x], stride, stride, SR, &lt);
    if (!meDark.lastCallOk()) { fprintf(stderr, "dark window / bright block: call rejected\n"); ++failures; }
    if (meDark.getInferredBitDepth() != 10) { fprintf(stderr, "dark window / bright block: inferred %d bits\n", meDark.getInferredBitDepth()); ++failures; }
    hmo_search_ctu(cb, stride, &dark[(M + cu_y) * stride + M + cu_x], stride, &p, ox, oy, osad, NULL);
    bad = 0;
    for (int i = 0; i < NUM_CTU_PARTS; i++) bad += meDark.getX()[i] != ox[i] || meDark.getY()[i] != oy[i] || meDark.getRuiCost()[i] != osad[i];
    if (bad) { fprintf(stderr, "dark window / bright block: %d slots differ from the oracle\n", bad); failures += bad; }
    // ... and the width stays latched: a dark block over the same dark window afterwards still runs as 10-bit content, same results
    std::vector<Pel> dcur(64 * 64);
    for (size_t i = 0; i < dcur.size(); ++i) dcur[i] = (Pel)(rnd() & 127);
    meDark.calcMotionVectors(&dcur[0], &dark[(M + cu_y) * stride + M + cu_x], stride, 64, SR, &lt);
    if (!meDark.lastCallOk() || meDark.getInferredBitDepth() != 10) { fprintf(stderr, "width not latched (%d)\n", meDark.getInferredBitDepth()); ++failures; }
    hmo_search_ctu(&dcur[0], 64, &dark[(M + cu_y) * stride + M + cu_x], stride, &p, ox, oy, osad, NULL);
    bad = 0;
    for (int i = 0; i < NUM_CTU_PARTS; i++) bad += meDark.getX()[i] != ox[i] || meDark.getY()[i] != oy[i] || meDark.getRuiCost()[i] != osad[i];
    if (bad) { fprintf(stderr, "latched width: %d slots differ from the oracle\n", bad); failures += bad; }
    // an 8-bit bi-prediction origin (samples in [-255, 510]) through the unmodified call sites must NOT widen the estimate
    TEncOpenCL me8;
    if (!me8.findDevice(0) || !me8.compileKernelSource("cl/sad.cl", "calcSAD_AMP") || !me8.createBuffers(64, 64, SR)) return 2;
    me8.setLambda(lambda);
    std::vector<Pel> bi8(64 * 64);
    for (size_t i = 0; i < bi8.size(); ++i) bi8[i] = (Pel)(2 * (Pel)(rnd() & 255) - (Pel)(rnd() & 255));
    bi8[0] = 510; bi8[1] = -255;
    me8.calcMotionVectors(&bi8[0], &ref[(M + cu_y) * stride + M + cu_x], stride, 64, SR, &lt);
    if (!me8.lastCallOk() || me8.getInferredBitDepth() != 8) { fprintf(stderr, "8-bit bi-prediction origin: ok %d, inferred %d bits\n", (int)me8.lastCallOk(), me8.getInferredBitDepth()); ++failures; }
  }
  // error behaviour: a sample outside [0,255] is reported, results flagged not-ok, no crash
  cur[(M)*stride + M] = 9999;   // beyond any bit depth the class could derive
  TComMv lt0(-8, -8);
  me.setCostMode(TEncOpenCL::ME_MODE_OCL_COMPAT);
  me.calcMotionVectors(&cur[M * stride + M], &ref[M * stride + M], stride, stride, SR, &lt0);
  if (me.lastCallOk()) { fprintf(stderr, "out-of-range sample was not rejected\n"); ++failures; }
  // ... and the tables are poisoned, not left at the previous CTU's values: the reference caller copies them unchecked
  for (int i = 0; i < NUM_CTU_PARTS; i++)
    if (me.getX()[i] != 0 || me.getY()[i] != 0 || me.getRuiCost()[i] != 0xFFFFFFFFu || me.getMvs()[i].getHor() != 0) {
      fprintf(stderr, "slot %d not poisoned after a failed call\n", i); ++failures; break;
    }
  // ... and neither are refinement tables: a failed search + refine call, stored for a [list][refIdx] that held good tables before
  // (stored above for [1][2]), leaves that entry unreadable -- the getters hand out poison (gpurun_out/r03p: they used to index
  // tables that were never allocated when the very first call failed)
  me.setCostMode(TEncOpenCL::ME_MODE_HM);
  me.setRefine(true, true);
  me.setSearchRangeRB(TComMv(8, 8));
  me.calcMotionVectors(&cur[M * stride + M], &ref[M * stride + M], stride, stride, SR, &lt0);
  if (me.lastCallOk() || me.fracOk()) { fprintf(stderr, "failed search + refine call reported ok\n"); ++failures; }
  me.markTables(1, 2, 41, 0);
  me.storeFrac(1, 2);
  if (me.fracStored(1, 2, 41, 0) || me.getFracCostStored(1, 2, 592) != 0xFFFFFFFFu || me.getFracDist(1, 2, 0) != 0xFFFFFFFFu ||
      me.getFracMv(1, 2, 100).getHor() != 0) { fprintf(stderr, "refinement tables of a failed call are readable\n"); ++failures; }
  {
    TEncOpenCL fresh;   // the very first call of an object fails: nothing was ever allocated
    if (!fresh.findDevice(0) || !fresh.createBuffers(64, 64, SR)) return 2;
    fresh.setCostMode(TEncOpenCL::ME_MODE_HM);
    fresh.setRefine(true, true);
    fresh.setSearchRangeRB(TComMv(8, 8));
    fresh.calcMotionVectors(&cur[M * stride + M], &ref[M * stride + M], stride, stride, SR, &lt0);
    fresh.markTables(0, 0, 1, 0);
    fresh.storeFrac(0, 0);
    if (fresh.lastCallOk() || fresh.fracStored(0, 0, 1, 0) || fresh.getFracCostStored(0, 0, 592) != 0xFFFFFFFFu || fresh.getFracMv(0, 0, 0).getVer() != 0) {
      fprintf(stderr, "fresh object, failed first call: refinement getters\n"); ++failures;
    }
  }
  printf("%s (%d mismatches)\n", failures ? "FAIL" : "PASS", failures);
  return failures ? 1 : 0;
}
