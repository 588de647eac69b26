// CPU test of the C++ sequence driver's planning (hm-opencl_amd/host/SequenceME.h): launches cover the pairs in order, every launch
// finds its pictures in the slots the plan says, no launch's own picture is evicted for it.  Exit code 0 = all invariants hold.
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include <thread>

#include "../../hm-opencl_amd/host/MultiDeviceME.h"

using namespace hmme_host;

static int check(const std::vector<std::pair<int, int> >& pairs, int k, int slots, bool expect_ok) {
  const std::vector<std::pair<int, int> > batches = plan_batches((int)pairs.size(), k);
  int next = 0;
  for (size_t b = 0; b < batches.size(); ++b) {
    if (batches[b].first != next || batches[b].second <= batches[b].first || batches[b].second - batches[b].first > k) return 1;
    next = batches[b].second;
  }
  if (next != (int)pairs.size()) return 2;
  std::vector<std::vector<PlaneLoad> > loads;
  std::vector<std::vector<std::pair<int, int> > > where;
  std::string err;
  const bool ok = plan_plane_loads(pairs, batches, slots, &loads, &where, &err);
  if (ok != expect_ok) return 3;
  if (!ok) return err.empty() ? 4 : 0;
  std::map<int, int> held;   // slot -> picture
  std::set<int> distinct;
  int n_loads = 0;
  for (size_t b = 0; b < batches.size(); ++b) {
    std::set<int> need;
    for (int i = batches[b].first; i < batches[b].second; ++i) { need.insert(pairs[i].first); need.insert(pairs[i].second); }
    for (size_t j = 0; j < loads[b].size(); ++j) {
      const PlaneLoad& l = loads[b][j];
      if (l.slot < 0 || l.slot >= slots) return 5;
      if (held.count(l.slot) && need.count(held[l.slot])) return 6;   // evicted a picture this very launch reads
      held[l.slot] = l.poc;
      ++n_loads;
    }
    if ((int)where[b].size() != batches[b].second - batches[b].first) return 7;
    for (int i = batches[b].first; i < batches[b].second; ++i) {
      const std::pair<int, int>& w = where[b][i - batches[b].first];
      if (held[w.first] != pairs[i].first || held[w.second] != pairs[i].second) return 8;
      distinct.insert(pairs[i].first); distinct.insert(pairs[i].second);
    }
  }
  if (slots >= 8 && k == 1 && n_loads != (int)distinct.size()) return 10;   // the GOP's working set fits 8 slots: one upload per picture
  return n_loads >= (int)distinct.size() ? 0 : 9;
}

int main() {
  // the random-access GOP of cfg/encoder_randomaccess_main.cfg:28-31 over 64 pictures
  std::vector<std::pair<int, int> > ra;
  const int pos[4] = {4, 2, 1, 3}, nref[5] = {0, 3, 2, 2, 1}, refs[5][3] = {{0, 0, 0}, {-1, 1, 3}, {-2, 2, 0}, {-1, 1, 0}, {-4, 0, 0}};
  for (int base = 0; base < 64; base += 4)
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < nref[pos[i]]; ++j) {
        const int cur = base + pos[i], ref = cur + refs[pos[i]][j];
        if (cur < 64 && ref >= 0 && ref < 64) ra.push_back(std::make_pair(cur, ref));
      }
  if (ra.size() != 124) { printf("FAIL: %zu pairs\n", ra.size()); return 1; }
  int rc = 0;
  const int cases[][3] = {{1, 8, 1}, {4, 10, 1}, {1, 2, 1}, {16, 34, 1}, {2, 4, 1}, {4, 3, 0}, {3, 5, 1}};
  for (size_t c = 0; c < sizeof cases / sizeof cases[0]; ++c) {
    const int r = check(ra, cases[c][0], cases[c][1], cases[c][2] != 0);
    if (r) { printf("FAIL: k=%d slots=%d -> %d\n", cases[c][0], cases[c][1], r); rc = 1; }
  }
  // run() without a context fails with a message, no crash; the planning error (too few slots) surfaces through run() as well
  {
    SequenceConfig cfg = {128, 64, 8, 16, 4, 3, 0, false, false};
    SequenceSearch s(0, cfg);
    if (s.run(ra, LumaReader(), 0) != HMME_ERR_ARG || s.error().empty()) { printf("FAIL: run() without a context\n"); rc = 1; }
  }
  // the YUV file reader: 8-bit 4:2:0 and 16-bit 4:0:0, pictures out of order, beyond the end of the file
  for (int bd = 8; bd <= 10; bd += 2) {
    const int w = 48, h = 32, n = 3, bps = bd == 8 ? 1 : 2, chroma = bd == 8 ? 1 : 0;
    const size_t luma = (size_t)w * h * bps, frame = luma + (chroma ? luma / 2 : 0);
    std::vector<unsigned char> file(frame * n);
    for (size_t i = 0; i < file.size(); ++i) file[i] = (unsigned char)(i * 7 + i / frame);
    char path[] = "/tmp/hmme_seq_test_XXXXXX";
    const int fd = mkstemp(path);
    if (fd < 0 || write(fd, &file[0], file.size()) != (ssize_t)file.size()) { printf("FAIL: temp file\n"); return 1; }
    close(fd);
    std::string err;
    LumaReader rd = yuv_file_reader(path, w, h, bd, chroma, &err);
    std::vector<unsigned char> buf(luma);
    const int order[3] = {2, 0, 1};
    for (int i = 0; i < 3; ++i)
      if (!rd || !rd(order[i], &buf[0]) || memcmp(&buf[0], &file[frame * order[i]], luma) != 0) { printf("FAIL: reader bd %d picture %d\n", bd, order[i]); rc = 1; }
    if (rd && rd(n, &buf[0])) { printf("FAIL: read beyond the end of the file\n"); rc = 1; }
    unlink(path);
    if (yuv_file_reader("/nonexistent/file.yuv", w, h, bd, chroma, &err) || err.empty()) { printf("FAIL: missing file\n"); rc = 1; }
  }
  // ---- N devices: the shard rule, the planner called from N host threads at once (one per device, as MultiDeviceSearch does), and the
  // failure paths of MultiDeviceSearch::run that need no GPU
  {
    for (int world = 1; world <= 8; ++world) {
      std::set<int> seen;
      size_t most = 0, least = 1 << 30;
      for (int r = 0; r < world; ++r) {
        const std::vector<int> mine = pairs_for_device((int)ra.size(), r, world);
        for (size_t i = 0; i < mine.size(); ++i) { if (mine[i] % world != r || !seen.insert(mine[i]).second) rc = 1; }
        most = mine.size() > most ? mine.size() : most; least = mine.size() < least ? mine.size() : least;
      }
      if (seen.size() != ra.size() || most - least > 1) { printf("FAIL: shard of %d\n", world); rc = 1; }
    }
    const int world = 8;
    std::vector<int> results(world, -1);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
      th.push_back(std::thread([&, r]() {
        std::vector<std::pair<int, int> > mine;
        const std::vector<int> idx = pairs_for_device((int)ra.size(), r, world);
        for (size_t i = 0; i < idx.size(); ++i) mine.push_back(ra[idx[i]]);
        int bad = 0;
        for (int k = 1; k <= 4; ++k) bad += check(mine, k, 2 * k + 6, true) != 0;
        results[r] = bad;
      }));
    for (int r = 0; r < world; ++r) th[r].join();
    for (int r = 0; r < world; ++r) if (results[r] != 0) { printf("FAIL: planner on thread %d\n", r); rc = 1; }
    SequenceConfig cfg = {128, 64, 8, 16, 2, 0, 0, false, false};
    std::vector<int> twice(2, 0), none;
    MultiDeviceSearch dup(twice, cfg, kGatherRccl, 57.9);
    if (dup.run(ra, [](int) { return LumaReader(); }, 0) != HMME_ERR_ARG || dup.error().find("distinct") == std::string::npos) { printf("FAIL: RCCL gather with one device twice\n"); rc = 1; }
    MultiDeviceSearch empty(none, cfg, kGatherPeer, 57.9);
    if (empty.run(ra, [](int) { return LumaReader(); }, 0) != HMME_ERR_ARG) { printf("FAIL: no devices\n"); rc = 1; }
    std::vector<std::pair<int, int> > nothing;
    MultiDeviceStats ms;
    if (empty.run(nothing, [](int) { return LumaReader(); }, &ms) != HMME_ERR_ARG) { printf("FAIL: no devices, no pairs\n"); rc = 1; }
    MultiDeviceSearch idle(twice, cfg, kGatherPeer, 57.9);
    if (idle.run(nothing, [](int) { return LumaReader(); }, &ms) != HMME_OK || ms.pairs_per_device.size() != 2) { printf("FAIL: empty pair list\n"); rc = 1; }
    std::vector<int> far(1, 1000);   // a device that does not exist: hmme_create fails, run() reports it, nothing leaks
    MultiDeviceSearch nodev(far, cfg, kGatherHost, 57.9);
    if (nodev.run(ra, [](int) { return LumaReader(); }, 0) != HMME_ERR_DEVICE || nodev.error().empty()) { printf("FAIL: missing device\n"); rc = 1; }
  }
  printf("%s\n", rc ? "FAIL" : "PASS");
  return rc;
}
