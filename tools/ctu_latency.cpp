// Latency of hmme_search_ctu from C++ (what TEncOpenCL::calcMotionVectors pays per call), without Python in the loop.
// Build: g++ -O2 -o ctu_latency_cpp tools/ctu_latency.cpp -Iinclude -Lhm-opencl_amd/csrc -lhmme -Wl,-rpath,$PWD/hm-opencl_amd/csrc
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hmme.h"

int main() {
  hmme_ctx* ctx = nullptr;
  if (hmme_create(0, 128, 0, &ctx) != HMME_OK) { fprintf(stderr, "create: %s\n", hmme_last_error(nullptr)); return 1; }
  hmme_set_lambda(ctx, 57.9);
  struct Case { const char* name; int sr, bd; } cases[] = {{"8bit_sr64", 64, 8}, {"8bit_sr8", 8, 8}, {"8bit_sr128", 128, 8}, {"10bit_sr64", 64, 10}, {"10bit_sr128", 128, 10}};
  printf("{");
  for (const Case& c : cases) {
    const int side = 64 + 2 * c.sr + 16, maxv = (1 << c.bd) - 1;
    std::vector<int16_t> cur(64 * 64), ref((size_t)side * side);
    srand(1);
    for (auto& v : cur) v = (int16_t)(rand() % (maxv + 1));
    for (auto& v : ref) v = (int16_t)(rand() % (maxv + 1));
    hmme_search_params p = {-c.sr, -c.sr, c.sr, c.sr, 5, -3, 1, c.bd};
    std::vector<int16_t> mv(2 * HMME_NUM_CTU_PARTS);
    std::vector<uint32_t> sad(HMME_NUM_CTU_PARTS);
    const int16_t* r0 = ref.data() + (size_t)(c.sr + 8) * side + (c.sr + 8);   // 8 samples of margin: window + refinement halo
    for (int i = 0; i < 5; ++i)
      if (hmme_search_ctu(ctx, cur.data(), 64, r0, side, &p, mv.data(), sad.data()) != HMME_OK) { fprintf(stderr, "%s\n", hmme_last_error(ctx)); return 1; }
    const int n = 200;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hmme_search_ctu(ctx, cur.data(), 64, r0, side, &p, mv.data(), sad.data());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n;
    printf("%s\"%s_ms_per_call\": %.4f", &c == cases ? "" : ", ", c.name, ms);
    // the same call with the refinement of the winners riding along (hmme_search_refine_ctu): on unrelated content nearly every slot
    // has its own MV (nothing shared between slots: the refinement's worst case), on a displaced copy of the reference one MV
    std::vector<int16_t> qmv(2 * HMME_NUM_CTU_PARTS);
    std::vector<uint32_t> cost(HMME_NUM_CTU_PARTS);
    std::vector<int16_t> moved(64 * 64);
    for (int y = 0; y < 64; ++y)
      for (int x = 0; x < 64; ++x) moved[y * 64 + x] = r0[(size_t)(y + 3) * side + x - 5];
    const int16_t* blocks[2] = {cur.data(), moved.data()};
    const char* tags[2] = {"search_refine_unrelated", "search_refine_coherent"};
    for (int k = 0; k < 2; ++k) {
      for (int i = 0; i < 3; ++i)
        if (hmme_search_refine_ctu(ctx, blocks[k], 64, r0, side, &p, 1, mv.data(), sad.data(), qmv.data(), cost.data()) != HMME_OK) { fprintf(stderr, "%s\n", hmme_last_error(ctx)); return 1; }
      const auto t1 = std::chrono::steady_clock::now();
      for (int i = 0; i < n; ++i) hmme_search_refine_ctu(ctx, blocks[k], 64, r0, side, &p, 1, mv.data(), sad.data(), qmv.data(), cost.data());
      const double ms2 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count() / n;
      printf(", \"%s_%s_ms_per_call\": %.4f", c.name, tags[k], ms2);
    }
  }
  // explicit weighted prediction (a fade): the window is weighted once on the device, the 16-bit kernel searches the weighted copy,
  // the refinement weights its interpolated prediction (hmme_search_ctu_w / hmme_search_refine_ctu_w)
  {
    const int sr = 64, side = 64 + 2 * sr + 16;
    std::vector<int16_t> cur(64 * 64), ref((size_t)side * side);
    srand(2);
    for (auto& v : ref) v = (int16_t)(rand() % 256);
    const int16_t* r0 = ref.data() + (size_t)(sr + 8) * side + (sr + 8);
    const hmme_weight w = {45, 9, 6, 32};
    for (int y = 0; y < 64; ++y)
      for (int x = 0; x < 64; ++x) {
        const int v = ((w.w0 * r0[(size_t)(y + 3) * side + x - 5] + w.round) >> w.shift) + w.offset;
        cur[y * 64 + x] = (int16_t)(v < 0 ? 0 : v > 255 ? 255 : v);
      }
    hmme_search_params p = {-sr, -sr, sr, sr, 5, -3, 1, 8};
    std::vector<int16_t> mv(2 * HMME_NUM_CTU_PARTS), qmv(2 * HMME_NUM_CTU_PARTS);
    std::vector<uint32_t> sad(HMME_NUM_CTU_PARTS), cost(HMME_NUM_CTU_PARTS);
    const int n = 200;
    for (int i = 0; i < 5; ++i)
      if (hmme_search_ctu_w(ctx, cur.data(), 64, r0, side, &p, &w, mv.data(), sad.data()) != HMME_OK) { fprintf(stderr, "%s\n", hmme_last_error(ctx)); return 1; }
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hmme_search_ctu_w(ctx, cur.data(), 64, r0, side, &p, &w, mv.data(), sad.data());
    printf(", \"8bit_sr64_weighted_ms_per_call\": %.4f", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n);
    for (int i = 0; i < 3; ++i)
      if (hmme_search_refine_ctu_w(ctx, cur.data(), 64, r0, side, &p, &w, 1, mv.data(), sad.data(), qmv.data(), cost.data()) != HMME_OK) { fprintf(stderr, "%s\n", hmme_last_error(ctx)); return 1; }
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hmme_search_refine_ctu_w(ctx, cur.data(), 64, r0, side, &p, &w, 1, mv.data(), sad.data(), qmv.data(), cost.data());
    printf(", \"8bit_sr64_weighted_search_refine_fade_ms_per_call\": %.4f", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n);
  }
  printf("}\n");
  hmme_destroy(ctx);
  return 0;
}
