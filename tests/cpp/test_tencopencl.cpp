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
  // error behaviour: a sample outside [0,255] is reported, results flagged not-ok, no crash
  cur[(M)*stride + M] = 999;
  TComMv lt0(-8, -8);
  me.setCostMode(TEncOpenCL::ME_MODE_OCL_COMPAT);
  me.calcMotionVectors(&cur[M * stride + M], &ref[M * stride + M], stride, stride, SR, &lt0);
  if (me.lastCallOk()) { fprintf(stderr, "out-of-range sample was not rejected\n"); ++failures; }
  printf("%s (%d mismatches)\n", failures ? "FAIL" : "PASS", failures);
  return failures ? 1 : 0;
}
