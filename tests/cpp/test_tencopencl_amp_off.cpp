// The TEncOpenCL class compiled for an encoder with AMP_ENC_SPEEDUP (TypeDef.h:206, :260-261: NUM_CTU_PARTS 425, getIndexBlock's second
// table TComDataCU.cpp:3393-4675): its tables are 425 entries in THAT layout, filled from the engine's 593 results.  Built with
// -DNUM_CTU_PARTS=425 (class source compiled in, not the 593-entry libhmme_host.so), run on the GPU, checked against the oracle.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../hm-opencl_amd/host/TEncOpenCL.h"
#include "../../include/hmme.h"
#include "../../oracle/hm_oracle.h"

static unsigned rng_state = 4242u;
static unsigned rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main() {
  if (NUM_CTU_PARTS != 425) { fprintf(stderr, "build with -DNUM_CTU_PARTS=425\n"); return 2; }
  const int SR = 8, M = 80, W = 128, H = 64, stride = W + 2 * M;
  std::vector<Pel> cur((H + 2 * M) * stride), ref((H + 2 * M) * stride);
  for (size_t i = 0; i < cur.size(); ++i) { cur[i] = (Pel)(rnd() & 255); ref[i] = (Pel)(rnd() & 255); }
  TEncOpenCL me;
  if (!me.findDevice(0) || !me.compileKernelSource("cl/sad.cl", "calcSAD") || !me.createBuffers(64, 64, SR)) { fprintf(stderr, "no device\n"); return 2; }
  const double lambda = 57.9;
  me.setLambda(lambda);
  me.setCostMode(TEncOpenCL::ME_MODE_HM);
  me.setFastEnc(true);
  me.setRefine(true, true);
  int failures = 0;
  const int cu_x = 64, cu_y = 0;
  Pel* piCtu = &cur[(M + cu_y) * stride + M + cu_x];
  Pel* piRefY = &ref[(M + cu_y) * stride + M + cu_x];
  hmo_params p;
  hmo_set_search_range(7, -2, SR, cu_x, cu_y, W, H, 64, &p.lt_x, &p.lt_y, &p.rb_x, &p.rb_y);
  p.pred_x = 7; p.pred_y = -2; p.lambda_q16 = hmo_lambda_q16(lambda); p.fen = 1; p.bit_depth = 8;
  TComMv lt((Short)p.lt_x, (Short)p.lt_y);
  me.setPredictor(TComMv(7, -2));
  me.setSearchRangeRB(TComMv((Short)p.rb_x, (Short)p.rb_y));
  me.calcMotionVectors(piCtu, piRefY, stride, stride, SR, &lt);
  if (!me.lastCallOk() || !me.fracOk()) { fprintf(stderr, "call failed\n"); return 1; }
  int32_t ox[HMO_NUM_CTU_PARTS], oy[HMO_NUM_CTU_PARTS];
  uint32_t osad[HMO_NUM_CTU_PARTS];
  hmo_search_ctu(piCtu, stride, piRefY, stride, &p, ox, oy, osad, NULL);
  for (int i = 0; i < NUM_CTU_PARTS; ++i) {
    const int s = hmme_amp_off_slot(i);
    hmo_rect r;
    hmo_slot_rect(s, &r);
    int x, y, w, h;
    if (!TEncOpenCL::slotRect(i, x, y, w, h) || x != r.x || y != r.y || w != r.w || h != r.h) { fprintf(stderr, "entry %d: rectangle\n", i); ++failures; }
    if (me.getX()[i] != ox[s] || me.getY()[i] != oy[s] || me.getRuiCost()[i] != osad[s] || me.getMvs()[i].getHor() != ox[s]) {
      if (failures < 10) fprintf(stderr, "entry %d (slot %d) differs from the oracle\n", i, s);
      ++failures;
    }
    if (i % 29 == 0) {   // refinement tables follow the same layout
      int hx, hy, qx, qy;
      uint32_t cost;
      hmo_frac_refine(piCtu + r.y * stride + r.x, stride, r.w, r.h, piRefY + r.y * stride + r.x, stride, ox[s], oy[s], 7, -2, p.lambda_q16, 1, 8,
                      &hx, &hy, &qx, &qy, &cost);
      if (me.getQMvs()[i].getHor() != 4 * ox[s] + 2 * hx + qx || me.getQMvs()[i].getVer() != 4 * oy[s] + 2 * hy + qy || me.getFracCost()[i] != cost) {
        fprintf(stderr, "entry %d: refinement\n", i); ++failures;
      }
    }
  }
  // the 64x64 2NxN / Nx2N parts sit in reverse order in this layout (TComDataCU.cpp:3396-3410)
  if (hmme_slot_index_amp_off(1, 0, 1, 0) != 420 || hmme_slot_index_amp_off(2, 0, 0, 0) != 423 || hmme_slot_index_amp_off(0, 0, 0, 0) != 424) { fprintf(stderr, "64x64 order\n"); ++failures; }
  printf("%s (%d mismatches)\n", failures ? "FAIL" : "PASS", failures);
  return failures ? 1 : 0;
}
