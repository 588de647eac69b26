// Latency of TEncOpenCL::calcMotionVectors as the encoder sees it (the class over the C ABI: hm-opencl_amd/host/TEncOpenCL.cpp), beside the
// C ABI call it wraps -- reference call sequence (ME_MODE_OCL_COMPAT: the bit depth is inferred from the samples of every call) and the
// patched encoder's (ME_MODE_HM), with and without the refinement tables.
// Build: g++ -O2 -o tools/class_latency tools/class_latency.cpp -Iinclude -Lhm-opencl_amd/host -lhmme_host -Lhm-opencl_amd/csrc -lhmme \
//        -Wl,-rpath,$PWD/hm-opencl_amd/host -Wl,-rpath,$PWD/hm-opencl_amd/csrc
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../hm-opencl_amd/host/TEncOpenCL.h"
#include "../include/hmme.h"

static double ms_since(std::chrono::steady_clock::time_point t0, int n) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
  const int n = 300;
  printf("{");
  bool first = true;
  for (int sr : {64, 8}) {
    const int M = 80 + sr, W = 192, H = 192, stride = W + 2 * M;
    std::vector<Pel> cur((size_t)(H + 2 * M) * stride), ref((size_t)(H + 2 * M) * stride);
    srand(3);
    for (auto& v : cur) v = (Pel)(rand() & 255);
    for (auto& v : ref) v = (Pel)(rand() & 255);
    Pel* piCtu = &cur[(size_t)(M + 64) * stride + M + 64];
    Pel* piRef = &ref[(size_t)(M + 64) * stride + M + 64];
    TEncOpenCL me;
    if (!me.findDevice(0) || !me.compileKernelSource("cl/sad.cl", "calcSAD_AMP") || !me.createBuffers(64, 64, sr)) { fprintf(stderr, "init failed\n"); return 2; }
    me.setEnabled(true);
    me.setLambda(57.9);
    TComMv lt((Short)-sr, (Short)-sr), rb((Short)sr, (Short)sr);
    // the C ABI call the class wraps, on the same block and window
    {
      hmme_ctx* ctx = nullptr;
      if (hmme_create(0, 128, 0, &ctx) != HMME_OK) return 2;
      hmme_set_lambda(ctx, 57.9);
      std::vector<int16_t> mv(2 * HMME_NUM_CTU_PARTS);
      std::vector<uint32_t> sad(HMME_NUM_CTU_PARTS);
      for (int compat = 0; compat < 2; ++compat) {   // the parameters each class mode hands down: HM's (FEN, predictor-relative cost) and the reference kernel's
        hmme_search_params p = {-sr, -sr, sr, sr, 0, 0, 1, 8};
        if (compat) { hmme_params_ocl_compat(&p, -sr, -sr, sr); p.bit_depth = 8; }
        for (int i = 0; i < 10; ++i) hmme_search_ctu(ctx, piCtu, stride, piRef, stride, &p, mv.data(), sad.data());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) hmme_search_ctu(ctx, piCtu, stride, piRef, stride, &p, mv.data(), sad.data());
        printf("%s\"sr%d_c_abi_%s_params_ms\": %.4f", first ? "" : ", ", sr, compat ? "compat" : "hm", ms_since(t0, n));
        first = false;
      }
      hmme_destroy(ctx);
    }
    for (int mode = 0; mode < 3; ++mode) {
      if (mode == 0) me.setCostMode(TEncOpenCL::ME_MODE_OCL_COMPAT);
      else {
        me.setCostMode(TEncOpenCL::ME_MODE_HM);
        me.setPredictor(TComMv(0, 0)); me.setSearchRangeRB(rb); me.setFastEnc(true); me.setBitDepth(8);
        me.setRefine(mode == 2, true);
      }
      for (int i = 0; i < 10; ++i) me.calcMotionVectors(piCtu, piRef, stride, stride, sr, &lt);
      if (!me.lastCallOk()) { fprintf(stderr, "calcMotionVectors failed\n"); return 3; }
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < n; ++i) me.calcMotionVectors(piCtu, piRef, stride, stride, sr, &lt);
      printf(", \"sr%d_class_%s_ms\": %.4f", sr, mode == 0 ? "compat" : mode == 1 ? "hm" : "hm_with_refinement", ms_since(t0, n));
    }
  }
  printf("}\n");
  return 0;
}
