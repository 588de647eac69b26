// Sanitizer run of the CPU-side code ("-fsanitize=address on host code"): the oracle (test infrastructure) and the
// TEncOpenCL host module + the GPU-free entry points of the C ABI, built with -fsanitize=address,undefined by
// tests/test_abi_cpu.py.  No GPU is needed: without one hmme_create fails and the class must degrade exactly as the
// reference's caller expects (createBuffers -> false; a stray calcMotionVectors poisons the tables).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../hm-opencl_amd/host/TEncOpenCL.h"
#include "../../include/hmme.h"
#include "../../oracle/hm_oracle.h"

static unsigned rng_state = 777u;
static unsigned rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "asan_driver: check failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
  // ---- oracle: exact-size buffers so that any over-read lands in a redzone
  const int SR = 5, M = 80, W = 136, H = 72, stride = W + 2 * M;
  std::vector<hmo_pel> cur((H + 2 * M) * stride), ref((H + 2 * M) * stride);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) { cur[(M + y) * stride + M + x] = (hmo_pel)(rnd() & 1023); ref[(M + y) * stride + M + x] = (hmo_pel)(rnd() & 1023); }
  hmo_extend_border(&cur[M * stride + M], stride, W, H, M, M);
  hmo_extend_border(&ref[M * stride + M], stride, W, H, M, M);
  const int n_ctu = 3 * 2;
  std::vector<int16_t> pred(2 * n_ctu);
  for (size_t i = 0; i < pred.size(); ++i) pred[i] = (int16_t)((int)(rnd() % 65) - 32);
  std::vector<int32_t> ox(n_ctu * HMO_NUM_CTU_PARTS), oy(n_ctu * HMO_NUM_CTU_PARTS);
  std::vector<uint32_t> os(n_ctu * HMO_NUM_CTU_PARTS);
  const uint32_t lq = hmo_lambda_q16(57.9);
  CHECK(hmo_search_frame(&cur[M * stride + M], stride, &ref[M * stride + M], stride, W, H, SR, pred.data(), lq, 1, 10, 0, n_ctu, 3, ox.data(),
                         oy.data(), os.data()) == n_ctu);
  // literal per-PU search == the fast all-slot form, for a few slots of the last (partial) CTU
  for (int slot = 0; slot < HMO_NUM_CTU_PARTS; slot += 37) {
    hmo_rect r;
    CHECK(hmo_slot_rect(slot, &r) == 0);
    const int ctu = n_ctu - 1, cx = (ctu % 3) * 64, cy = (ctu / 3) * 64;
    hmo_params p;
    hmo_set_search_range(pred[2 * ctu], pred[2 * ctu + 1], SR, cx, cy, W, H, 64, &p.lt_x, &p.lt_y, &p.rb_x, &p.rb_y);
    p.pred_x = pred[2 * ctu]; p.pred_y = pred[2 * ctu + 1]; p.lambda_q16 = lq; p.fen = 1; p.bit_depth = 10;
    int mx, my;
    uint32_t sad;
    hmo_pattern_search(&cur[(M + cy + r.y) * stride + M + cx + r.x], stride, r.w, r.h, &ref[(M + cy + r.y) * stride + M + cx + r.x], stride, &p,
                       &mx, &my, &sad);
    CHECK(mx == ox[ctu * HMO_NUM_CTU_PARTS + slot] && my == oy[ctu * HMO_NUM_CTU_PARTS + slot] && sad == os[ctu * HMO_NUM_CTU_PARTS + slot]);
    int hx, hy, qx, qy;
    uint32_t cost;
    hmo_frac_refine(&cur[(M + cy + r.y) * stride + M + cx + r.x], stride, r.w, r.h, &ref[(M + cy + r.y) * stride + M + cx + r.x], stride, mx, my,
                    p.pred_x, p.pred_y, lq, slot & 1, 10, &hx, &hy, &qx, &qy, &cost);
    CHECK(hx >= -1 && hx <= 1 && qy >= -1 && qy <= 1);
  }
  long probes = 0;
  double s4 = 0;
  CHECK(hmo_tz_frame(&cur[M * stride + M], stride, &ref[M * stride + M], stride, W, H, 64, pred.data(), lq, 1, 10, 0, n_ctu, 2, 1, ox.data(),
                     oy.data(), os.data(), &probes, &s4) == n_ctu);
  CHECK(probes > 0);
  for (int key = 0; key < 64; ++key) (void)hmo_index_block(key & 7, (key >> 3) & 3, key & 1, (key * 7) & 255, 64 >> ((key >> 3) & 3));

  // ---- C ABI entry points that need no GPU
  int x, y, w, h, seen = 0;
  for (int s = 0; s < HMME_NUM_CTU_PARTS; ++s) { CHECK(hmme_slot_rect(s, &x, &y, &w, &h) == HMME_OK); seen += (x + w <= 64 && y + h <= 64); }
  CHECK(seen == HMME_NUM_CTU_PARTS && hmme_slot_rect(HMME_NUM_CTU_PARTS, &x, &y, &w, &h) == HMME_ERR_ARG);
  for (int ps = 0; ps < 8; ++ps)
    for (int d = 0; d < 4; ++d)
      for (int z = 0; z < 256; z += 3) {
        const int s = hmme_slot_index(ps, d, z & 1, z);
        CHECK(s >= -1 && s < HMME_NUM_CTU_PARTS);
      }
  int ltx, lty, rbx, rby, a, b, c, d2;
  hmme_set_search_range(-300, 77, 64, 128, 64, W, H, &ltx, &lty, &rbx, &rby);
  hmo_set_search_range(-300, 77, 64, 128, 64, W, H, 64, &a, &b, &c, &d2);
  CHECK(ltx == a && lty == b && rbx == c && rby == d2);
  hmme_search_params sp;
  hmme_params_ocl_compat(&sp, -8, -8, 8);
  CHECK(sp.rb_x == 8 && sp.shift_free == 1 && hmme_num_ctus(1920, 1080) == 510);

  // ---- the host class where no MI355X exists (this container): every step fails cleanly, tables are poisoned
  hmme_ctx* ctx = 0;
  const int rc = hmme_create(0, 64, 0, &ctx);
  if (rc == HMME_OK) {   // a GPU box: still fine, run one real call through the class below
    hmme_destroy(ctx);
  } else {
    CHECK(ctx == 0 && std::strlen(hmme_last_error(0)) > 0);
  }
  TEncOpenCL me;
  CHECK(me.findDevice(0));
  const bool have_gpu = me.createBuffers(64, 64, SR);
  CHECK(have_gpu == (rc == HMME_OK));
  CHECK(!me.createBuffers(32, 32, SR) || !have_gpu);   // only 64x64 CTUs, like cl/sad.cl
  me.setLambda(57.9);
  TComMv lt((Short)-SR, (Short)-SR);
  std::vector<Pel> c8(cur.size()), r8(ref.size());
  for (size_t i = 0; i < c8.size(); ++i) { c8[i] = (Pel)(cur[i] >> 2); r8[i] = (Pel)(ref[i] >> 2); }
  me.calcMotionVectors(&c8[M * stride + M], &r8[M * stride + M], stride, stride, SR, &lt);
  CHECK(me.lastCallOk() == have_gpu);
  if (!have_gpu)
    for (int i = 0; i < NUM_CTU_PARTS; ++i) CHECK(me.getX()[i] == 0 && me.getY()[i] == 0 && me.getRuiCost()[i] == 0xFFFFFFFFu);
  me.calcMotionVectorsEdge(&c8[M * stride + M + 128], stride, 8, 8, &r8[M * stride + M + 128], stride, SR, TComMv(3, -2), 128, 0, W, H);
  CHECK(me.lastCallOk() == have_gpu && me.numCalls() == 2);
  // ---- the refinement tables on the path where the engine reported an error (gpurun_out/r03p: the class's stored-table getters
  // indexed tables that storeFrac never allocated).  A TEncOpenCL whose createBuffers FAILED (device -1 never exists, with or
  // without a GPU in the box): calcMotionVectors -> storeFrac -> every getter must hand out the poison values, in range or not
  {
    TEncOpenCL bad;
    bad.setDeviceId(-1);
    CHECK(!bad.findDevice(-1));
    CHECK(!bad.createBuffers(64, 64, SR));
    bad.setCostMode(TEncOpenCL::ME_MODE_HM);
    bad.setFastEnc(true);
    bad.setRefine(true, true);
    bad.setPredictor(TComMv(1, -1));
    bad.setSearchRangeRB(TComMv((Short)SR, (Short)SR));
    bad.calcMotionVectors(&c8[M * stride + M], &r8[M * stride + M], stride, stride, SR, &lt);
    CHECK(!bad.lastCallOk() && !bad.fracOk() && bad.numFailed() == 1);
    bad.markTables(1, 2, 40, 9);
    bad.storeFrac(1, 2);
    bad.storeFrac(-1, 99);                                       // out-of-range indices are ignored
    CHECK(bad.tablesValidFor(1, 2, 40, 9) && !bad.fracStored(1, 2, 40, 9) && !bad.fracStored(-1, 99, 40, 9));
    const int lists[] = {-1, 0, 1, 2}, refs[] = {-1, 0, 2, 32, 33}, slots[] = {-1, 0, 592, 593};
    for (int l : lists)
      for (int r : refs)
        for (int s2 : slots) {
          CHECK(bad.getFracMv(l, r, s2).getHor() == 0 && bad.getFracMv(l, r, s2).getVer() == 0);
          CHECK(bad.getFracDist(l, r, s2) == 0xFFFFFFFFu && bad.getFracCostStored(l, r, s2) == 0xFFFFFFFFu);
        }
    for (int i = 0; i < NUM_CTU_PARTS; ++i)
      CHECK(bad.getMvs()[i].getHor() == 0 && bad.getRuiCost()[i] == 0xFFFFFFFFu && bad.getMvs(true)[i].getVer() == 0);
  }
  if (have_gpu) {   // a successful store, then a failed call for the same [list][refIdx]: the stored tables are invalidated, not kept
    me.setCostMode(TEncOpenCL::ME_MODE_HM);
    me.setFastEnc(true);
    me.setRefine(true, true);
    me.setPredictor(TComMv(0, 0));
    me.setSearchRangeRB(TComMv((Short)SR, (Short)SR));
    me.calcMotionVectors(&c8[M * stride + M], &r8[M * stride + M], stride, stride, SR, &lt);
    CHECK(me.lastCallOk() && me.fracOk());
    me.markTables(0, 0, 1, 0);
    me.storeFrac(0, 0);
    CHECK(me.fracStored(0, 0, 1, 0) && me.getFracCostStored(0, 0, 592) == me.getFracCost()[592]);
    c8[M * stride + M] = 9999;                                   // out of any range: the next call fails
    me.calcMotionVectors(&c8[M * stride + M], &r8[M * stride + M], stride, stride, SR, &lt);
    CHECK(!me.lastCallOk() && !me.fracOk());
    me.markTables(0, 0, 1, 1);
    me.storeFrac(0, 0);
    CHECK(!me.fracStored(0, 0, 1, 1) && me.getFracCostStored(0, 0, 592) == 0xFFFFFFFFu && me.getFracMv(0, 0, 592).getHor() == 0);
  }
  printf("asan_driver: PASS (%s)\n", have_gpu ? "with GPU" : "no GPU: failure paths");
  return 0;
}
