// me_stream -- open-loop motion estimation of a YUV sequence from C++ (hm-opencl_amd/host/SequenceME.h over the C ABI): the
// C++ counterpart of tools/me_sequence.py --stream.  Reads the file through a reader thread, streams the pictures
// through a plane ring, searches the GOP's picture pairs (optionally several per launch, optionally refined) and leaves the tables
// in page-locked host memory; prints one JSON line and, with --out, writes the tables for the parity test.
// --gpus N (devices 0..N-1) or --devices a,b,... (an index may repeat: several contexts on one GPU, a rehearsal): picture pair p goes
// to device p mod N, one host thread per device, tables gathered into rank 0's memory (hm-opencl_amd/host/MultiDeviceME.h;
// --gather rccl = ncclSend / ncclRecv to device 0, peer = hipMemcpyPeerAsync, host = every device downloads into its places).
// Build: make -C hm-opencl_amd/host me_stream
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../hm-opencl_amd/host/MultiDeviceME.h"

// (current POC, reference POC) pairs of the reference's GOP presets in coding order -- hmme/shard.py gop_pairs
// (cfg/encoder_randomaccess_main.cfg:28-31, cfg/encoder_lowdelay_P_main.cfg:24-27)
static std::vector<std::pair<int, int> > gop_pairs(int n_frames, const std::string& gop) {
  static const int ra_pos[4] = {4, 2, 1, 3};
  static const int ra_refs[5][3] = {{0, 0, 0}, {-1, 1, 3}, {-2, 2, 0}, {-1, 1, 0}, {-4, 0, 0}};
  static const int ra_n[5] = {0, 3, 2, 2, 1};
  static const int ld_refs[5][4] = {{0, 0, 0, 0}, {-1, -5, -9, -13}, {-1, -2, -6, -10}, {-1, -3, -7, -11}, {-1, -4, -8, -12}};
  std::vector<std::pair<int, int> > pairs;
  for (int base = 0; base < n_frames; base += 4)
    for (int i = 0; i < 4; ++i) {
      const int pos = gop == "randomaccess" ? ra_pos[i] : i + 1, cur = base + pos;
      if (cur >= n_frames) continue;
      const int n = gop == "randomaccess" ? ra_n[pos] : 4;
      for (int j = 0; j < n; ++j) {
        const int ref = cur + (gop == "randomaccess" ? ra_refs[pos][j] : ld_refs[pos][j]);
        if (ref >= 0 && ref < n_frames) pairs.push_back(std::make_pair(cur, ref));
      }
    }
  return pairs;
}

int main(int argc, char** argv) {
  std::string yuv, gop = "randomaccess", out, gather = "rccl";
  std::vector<int> devices;
  int w = 0, h = 0, frames = 16, sr = 64, bd = 8, k = 1, slots = 0, repeat = 1, chroma = 1;
  bool refine = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    const char* v = i + 1 < argc ? argv[i + 1] : "";
    if (a == "--yuv") { yuv = v; ++i; }
    else if (a == "--size") { if (sscanf(v, "%dx%d", &w, &h) != 2) { fprintf(stderr, "--size WxH\n"); return 2; } ++i; }
    else if (a == "--frames") { frames = atoi(v); ++i; }
    else if (a == "--gop") { gop = v; ++i; }
    else if (a == "--search-range") { sr = atoi(v); ++i; }
    else if (a == "--bit-depth") { bd = atoi(v); ++i; }
    else if (a == "--pairs-per-launch") { k = atoi(v); ++i; }
    else if (a == "--slots") { slots = atoi(v); ++i; }
    else if (a == "--chroma") { chroma = atoi(v) == 400 ? 0 : 1; ++i; }
    else if (a == "--repeat") { repeat = atoi(v); ++i; }
    else if (a == "--gpus") { devices.clear(); for (int d = 0; d < atoi(v); ++d) devices.push_back(d); ++i; }
    else if (a == "--devices") { devices.clear(); for (const char* q = v; *q;) { devices.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q) ++q; } ++i; }
    else if (a == "--gather") { gather = v; ++i; }
    else if (a == "--refine") refine = true;
    else if (a == "--out") { out = v; ++i; }
    else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  if (yuv.empty() || w <= 0 || h <= 0 || (gop != "randomaccess" && gop != "lowdelay_P")) {
    fprintf(stderr, "usage: me_stream --yuv FILE --size WxH [--frames N --gop randomaccess|lowdelay_P --search-range SR --bit-depth BD "
                    "--pairs-per-launch K --slots S --chroma 420|400 --refine --repeat R --out FILE --gpus N | --devices a,b,.. --gather rccl|peer|host]\n");
    return 2;
  }
  std::string err;
  hmme_host::LumaReader reader = hmme_host::yuv_file_reader(yuv, w, h, bd, chroma, &err);
  if (!reader) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
  const std::vector<std::pair<int, int> > pairs = gop_pairs(frames, gop);
  if (!devices.empty()) {   // ---- N devices from this one process
    const hmme_host::GatherVia via = gather == "peer" ? hmme_host::kGatherPeer : (gather == "host" ? hmme_host::kGatherHost : hmme_host::kGatherRccl);
    if (gather != "peer" && gather != "host" && gather != "rccl") { fprintf(stderr, "--gather rccl|peer|host\n"); return 2; }
    hmme_host::SequenceConfig cfg = {w, h, bd, sr, k, slots, 0, refine, false};
    hmme_host::MultiDeviceSearch md(devices, cfg, via, 57.9);
    hmme_host::MultiDeviceStats ms;
    for (int r = 0; r < (repeat < 1 ? 1 : repeat); ++r)
      if (md.run(pairs, [&](int) { return reader; }, &ms) != HMME_OK) { fprintf(stderr, "%s\n", md.error().c_str()); return 1; }   // pread: one reader serves every rank
    const size_t per_pair = (size_t)md.num_ctus() * HMME_NUM_CTU_PARTS, n = pairs.size();
    std::string ppd, dsec;
    for (size_t r = 0; r < ms.pairs_per_device.size(); ++r) {
      char b[64];
      snprintf(b, sizeof b, "%s%d", r ? ", " : "", ms.pairs_per_device[r]); ppd += b;
      snprintf(b, sizeof b, "%s%.4f", r ? ", " : "", ms.device_seconds[r]); dsec += b;
    }
    printf("{\"pairs\": %zu, \"n_ctu\": %d, \"gpus\": %d, \"seconds\": %.4f, \"pairs_per_s\": %.2f, \"search_seconds\": %.4f, \"gather\": \"%s\", "
           "\"gather_seconds\": %.4f, \"gather_bytes\": %zu, \"pairs_per_device\": [%s], \"device_seconds\": [%s], \"pairs_per_launch\": %d, "
           "\"refine\": %s, \"bit_depth\": %d, \"host\": \"C++ (hm-opencl_amd/host/MultiDeviceME)\"}\n",
           n, md.num_ctus(), md.world(), ms.seconds, n / ms.seconds, ms.search_seconds, gather.c_str(), ms.gather_seconds, ms.gather_bytes, ppd.c_str(),
           dsec.c_str(), k, refine ? "true" : "false", bd);
    if (!out.empty()) {
      FILE* f = fopen(out.c_str(), "wb");
      if (!f) { fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
      const int32_t hdr[4] = {(int32_t)n, md.num_ctus(), refine ? 1 : 0, 0};
      fwrite(hdr, sizeof hdr, 1, f);
      for (size_t i = 0; i < n; ++i) { const int32_t pr[2] = {pairs[i].first, pairs[i].second}; fwrite(pr, sizeof pr, 1, f); }
      fwrite(md.mv(), 4, per_pair * n, f);
      fwrite(md.sad(), 4, per_pair * n, f);
      if (refine) { fwrite(md.qmv(), 4, per_pair * n, f); fwrite(md.cost(), 4, per_pair * n, f); }
      fclose(f);
    }
    return 0;
  }
  hmme_ctx* ctx = 0;
  if (hmme_create(0, sr > 64 ? sr : 64, 0, &ctx) != HMME_OK) { fprintf(stderr, "hmme_create: %s\n", hmme_last_error(0)); return 1; }
  hmme_set_lambda(ctx, 57.9);
  hmme_host::SequenceConfig cfg = {w, h, bd, sr, k, slots, 0, refine, false};
  hmme_host::SequenceStats st = {0, 0, 0, 0, 0};
  {
    hmme_host::SequenceSearch seq(ctx, cfg);
    for (int r = 0; r < (repeat < 1 ? 1 : repeat); ++r)
      if (seq.run(pairs, reader, &st) != HMME_OK) { fprintf(stderr, "%s\n", seq.error().c_str()); return 1; }
    const size_t per_pair = (size_t)seq.num_ctus() * HMME_NUM_CTU_PARTS, n = pairs.size();
    printf("{\"pairs\": %zu, \"n_ctu\": %d, \"seconds\": %.4f, \"pairs_per_s\": %.2f, \"read_seconds\": %.4f, \"launches\": %d, \"uploads\": %d, "
           "\"plane_slots\": %d, \"pairs_per_launch\": %d, \"refine\": %s, \"bit_depth\": %d, \"host\": \"C++ (hm-opencl_amd/host/SequenceME)\"}\n",
           n, seq.num_ctus(), st.seconds, n / st.seconds, st.read_seconds, st.launches, st.uploads, st.plane_slots, k, refine ? "true" : "false", bd);
    if (!out.empty()) {
      FILE* f = fopen(out.c_str(), "wb");
      if (!f) { fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
      const int32_t hdr[4] = {(int32_t)n, seq.num_ctus(), refine ? 1 : 0, 0};
      fwrite(hdr, sizeof hdr, 1, f);
      for (size_t i = 0; i < n; ++i) { const int32_t pr[2] = {pairs[i].first, pairs[i].second}; fwrite(pr, sizeof pr, 1, f); }
      fwrite(seq.mv(), 4, per_pair * n, f);
      fwrite(seq.sad(), 4, per_pair * n, f);
      if (refine) { fwrite(seq.qmv(), 4, per_pair * n, f); fwrite(seq.cost(), 4, per_pair * n, f); }
      fclose(f);
    }
  }
  hmme_destroy(ctx);
  return 0;
}
