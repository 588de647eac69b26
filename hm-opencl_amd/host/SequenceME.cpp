// SequenceME.cpp -- see SequenceME.h.  Only include/hmme.h and the HIP runtime API; no torch, no Python.
#include "SequenceME.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <thread>

#include <hip/hip_runtime_api.h>

namespace hmme_host {

std::vector<std::pair<int, int> > plan_batches(int n_pairs, int pairs_per_launch) {
  const int k = std::max(1, std::min(16, pairs_per_launch));
  std::vector<std::pair<int, int> > b;
  for (int i = 0; i < n_pairs; i += k) b.push_back(std::make_pair(i, std::min(n_pairs, i + k)));
  return b;
}

bool plan_plane_loads(const std::vector<std::pair<int, int> >& pairs, const std::vector<std::pair<int, int> >& batches, int n_slots,
                      std::vector<std::vector<PlaneLoad> >* loads, std::vector<std::vector<std::pair<int, int> > >* where, std::string* err) {
  const int nb = (int)batches.size();
  std::vector<std::vector<int> > need(nb);      // sorted distinct pictures of each launch
  std::map<int, std::vector<int> > uses;        // picture -> launches that read it, ascending
  for (int b = 0; b < nb; ++b) {
    std::set<int> s;
    for (int i = batches[b].first; i < batches[b].second; ++i) { s.insert(pairs[i].first); s.insert(pairs[i].second); }
    need[b].assign(s.begin(), s.end());
    if ((int)need[b].size() > n_slots) {
      if (err) *err = "a launch needs " + std::to_string(need[b].size()) + " pictures resident, only " + std::to_string(n_slots) + " plane slots";
      return false;
    }
    for (size_t j = 0; j < need[b].size(); ++j) uses[need[b][j]].push_back(b);
  }
  std::map<int, int> resident;                  // picture -> slot
  int next_free = 0;
  loads->assign(nb, std::vector<PlaneLoad>());
  where->assign(nb, std::vector<std::pair<int, int> >());
  for (int b = 0; b < nb; ++b) {
    const std::vector<int>& n = need[b];
    for (size_t j = 0; j < n.size(); ++j) {
      const int p = n[j];
      if (resident.count(p)) continue;
      int slot;
      if (next_free < n_slots) {
        slot = next_free++;
      } else {
        // victim: not read by this launch; prefer one the previous launch does not read either (its refill then overlaps that
        // launch's search), then the one whose next use lies farthest ahead
        int victim = -1, v_notprev = -1, v_next = -1;
        for (std::map<int, int>::const_iterator it = resident.begin(); it != resident.end(); ++it) {
          const int q = it->first;
          if (std::binary_search(n.begin(), n.end(), q)) continue;
          const int notprev = (b == 0 || !std::binary_search(need[b - 1].begin(), need[b - 1].end(), q)) ? 1 : 0;
          const std::vector<int>& u = uses[q];
          const std::vector<int>::const_iterator nu = std::lower_bound(u.begin(), u.end(), b);
          const int next = nu == u.end() ? (1 << 30) : *nu;
          if (victim < 0 || notprev > v_notprev || (notprev == v_notprev && next > v_next)) { victim = q; v_notprev = notprev; v_next = next; }
        }
        slot = resident[victim];
        resident.erase(victim);
      }
      resident[p] = slot;
      PlaneLoad l = {p, slot};
      (*loads)[b].push_back(l);
    }
    for (int i = batches[b].first; i < batches[b].second; ++i)
      (*where)[b].push_back(std::make_pair(resident[pairs[i].first], resident[pairs[i].second]));
  }
  return true;
}

namespace {
// single-producer / single-consumer hand-over between the reader thread and the launching thread
template <typename T>
class Channel {
 public:
  void put(const T& v) { { std::lock_guard<std::mutex> g(m_); q_.push_back(v); } cv_.notify_one(); }
  T get() {
    std::unique_lock<std::mutex> g(m_);
    cv_.wait(g, [this] { return !q_.empty(); });
    T v = q_.front();
    q_.pop_front();
    return v;
  }
 private:
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<T> q_;
};
struct Ready { int poc, buf; };           // poc < 0: the reader failed / stopped
struct Free { int buf; hipEvent_t ev; };  // buf < 0: stop
}  // namespace

SequenceSearch::SequenceSearch(hmme_ctx* ctx, const SequenceConfig& cfg)
    : ctx_(ctx), cfg_(cfg), n_ctu_(hmme_num_ctus(cfg.width, cfg.height)), s_copy_(0), s_compute_(0), s_download_(0), d_mv_(0), d_sad_(0),
      d_qmv_(0), d_cost_(0), h_mv_(0), h_qmv_(0), h_sad_(0), h_cost_(0), cap_pairs_(0), x_mv_(0), x_qmv_(0), x_sad_(0), x_cost_(0) {
  if (cfg_.pairs_per_launch < 1) cfg_.pairs_per_launch = 1;
  if (cfg_.pairs_per_launch > 16) cfg_.pairs_per_launch = 16;
  if (cfg_.plane_slots <= 0) cfg_.plane_slots = std::max(8, 2 * cfg_.pairs_per_launch + 2);
  if (cfg_.host_buffers <= 0) cfg_.host_buffers = 4;
}

SequenceSearch::~SequenceSearch() { release(); }

void SequenceSearch::set_host_output(int16_t* mv, uint32_t* sad, int16_t* qmv, uint32_t* cost, const std::vector<int>& dest_index) {
  x_mv_ = mv; x_sad_ = sad; x_qmv_ = qmv; x_cost_ = cost;
  x_index_ = dest_index;
}

void SequenceSearch::release() {
  for (size_t i = 0; i < planes_.size(); ++i) hmme_plane_destroy(planes_[i]);
  planes_.clear();
  for (size_t i = 0; i < host_bufs_.size(); ++i) hipHostFree(host_bufs_[i]);
  host_bufs_.clear();
  if (s_copy_) hipStreamDestroy((hipStream_t)s_copy_);
  if (s_compute_) hipStreamDestroy((hipStream_t)s_compute_);
  if (s_download_) hipStreamDestroy((hipStream_t)s_download_);
  s_copy_ = s_compute_ = s_download_ = 0;
  hipFree(d_mv_); hipFree(d_sad_); hipFree(d_qmv_); hipFree(d_cost_);
  d_mv_ = d_sad_ = d_qmv_ = d_cost_ = 0;
  if (h_mv_) hipHostFree(h_mv_);
  if (h_sad_) hipHostFree(h_sad_);
  if (h_qmv_) hipHostFree(h_qmv_);
  if (h_cost_) hipHostFree(h_cost_);
  h_mv_ = h_qmv_ = 0; h_sad_ = h_cost_ = 0;
  cap_pairs_ = 0;
}

int SequenceSearch::fail(int code, const std::string& what) {
  err_ = what;
  fprintf(stderr, "hmme SequenceSearch: ERROR: %s\n", what.c_str());
  return code;
}

#define SEQ_HIP(call)                                                                                         \
  do {                                                                                                        \
    const hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) return fail(HMME_ERR_DEVICE, std::string(#call " -> ") + hipGetErrorString(e_));    \
  } while (0)
#define SEQ_ME(call)                                                                                          \
  do {                                                                                                        \
    const int rc_ = (call);                                                                                   \
    if (rc_ != HMME_OK) return fail(rc_, std::string(#call ": ") + hmme_last_error(ctx_));                    \
  } while (0)

int SequenceSearch::run(const std::vector<std::pair<int, int> >& pairs, const LumaReader& read_luma, SequenceStats* stats) {
  if (!ctx_) return fail(HMME_ERR_ARG, "no context");
  if (!read_luma) return fail(HMME_ERR_ARG, "no picture source");
  const int device = hmme_device_index(ctx_);
  SEQ_HIP(hipSetDevice(device));              // this object makes HIP calls of its own beside the library's
  const int n_pairs = (int)pairs.size();
  const int bps = cfg_.bit_depth == 8 ? 1 : 2;
  const std::vector<std::pair<int, int> > batches = plan_batches(n_pairs, cfg_.pairs_per_launch);
  std::vector<std::vector<PlaneLoad> > loads;
  std::vector<std::vector<std::pair<int, int> > > where;
  std::string perr;
  if (!plan_plane_loads(pairs, batches, cfg_.plane_slots, &loads, &where, &perr)) return fail(HMME_ERR_ARG, perr);
  int slots_used = 0, n_uploads = 0;
  for (size_t b = 0; b < loads.size(); ++b)
    for (size_t j = 0; j < loads[b].size(); ++j) { slots_used = std::max(slots_used, loads[b][j].slot + 1); ++n_uploads; }

  // ---- resources (kept between runs)
  if (!s_copy_) {
    hipStream_t s;
    SEQ_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); s_copy_ = s;
    SEQ_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); s_compute_ = s;
    SEQ_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); s_download_ = s;
  }
  while ((int)planes_.size() < slots_used) {
    hmme_plane* pl = 0;
    SEQ_ME(hmme_plane_create_ex(ctx_, cfg_.width, cfg_.height, cfg_.bit_depth, &pl));
    planes_.push_back(pl);
  }
  const size_t pic_bytes = (size_t)cfg_.width * cfg_.height * bps;
  while ((int)host_bufs_.size() < cfg_.host_buffers) {
    void* p = 0;
    SEQ_HIP(hipHostMalloc(&p, pic_bytes, hipHostMallocDefault));
    host_bufs_.push_back(p);
  }
  const size_t per_pair = (size_t)n_ctu_ * HMME_NUM_CTU_PARTS;   // slots of one pair's tables
  if (x_mv_ && ((int)x_index_.size() < n_pairs || !x_sad_ || (cfg_.refine && (!x_qmv_ || !x_cost_))))
    return fail(HMME_ERR_ARG, "set_host_output: arrays / destination indices do not cover the pairs of this run");
  if ((size_t)n_pairs > cap_pairs_) {
    hipFree(d_mv_); hipFree(d_sad_); hipFree(d_qmv_); hipFree(d_cost_);
    if (h_mv_) hipHostFree(h_mv_);
    if (h_sad_) hipHostFree(h_sad_);
    if (h_qmv_) hipHostFree(h_qmv_);
    if (h_cost_) hipHostFree(h_cost_);
    d_mv_ = d_sad_ = d_qmv_ = d_cost_ = 0; h_mv_ = h_qmv_ = 0; h_sad_ = h_cost_ = 0; cap_pairs_ = 0;
    const bool own_host = !cfg_.no_download && !x_mv_;   // otherwise the tables stay on the device or go to the caller's arrays
    SEQ_HIP(hipMalloc(&d_mv_, 4 * per_pair * n_pairs));
    SEQ_HIP(hipMalloc(&d_sad_, 4 * per_pair * n_pairs));
    if (own_host) SEQ_HIP(hipHostMalloc((void**)&h_mv_, 4 * per_pair * n_pairs, hipHostMallocDefault));
    if (own_host) SEQ_HIP(hipHostMalloc((void**)&h_sad_, 4 * per_pair * n_pairs, hipHostMallocDefault));
    if (cfg_.refine) {
      SEQ_HIP(hipMalloc(&d_qmv_, 4 * per_pair * n_pairs));
      SEQ_HIP(hipMalloc(&d_cost_, 4 * per_pair * n_pairs));
      if (own_host) SEQ_HIP(hipHostMalloc((void**)&h_qmv_, 4 * per_pair * n_pairs, hipHostMallocDefault));
      if (own_host) SEQ_HIP(hipHostMalloc((void**)&h_cost_, 4 * per_pair * n_pairs, hipHostMallocDefault));
    }
    cap_pairs_ = n_pairs;
  }
  // events of this run: destroyed on every way out of run(), the early error returns included
  struct Events {
    std::vector<hipEvent_t> v;
    ~Events() { for (size_t i = 0; i < v.size(); ++i) if (v[i]) hipEventDestroy(v[i]); }
    hipError_t create(size_t n) {
      v.assign(n, (hipEvent_t)0);
      for (size_t i = 0; i < n; ++i) { const hipError_t e = hipEventCreateWithFlags(&v[i], hipEventDisableTiming); if (e != hipSuccess) return e; }
      return hipSuccess;
    }
    hipEvent_t operator[](size_t i) const { return v[i]; }
  } buf_events, batch_events;
  SEQ_HIP(buf_events.create(cfg_.host_buffers));
  SEQ_HIP(batch_events.create(batches.size()));

  const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  // ---- reader thread: pictures in upload order into the page-locked buffers
  Channel<Ready> ready;
  Channel<Free> free_bufs;
  for (int i = 0; i < cfg_.host_buffers; ++i) { Free f = {i, (hipEvent_t)0}; free_bufs.put(f); }
  std::atomic<bool> stop(false);
  double read_s = 0.0;
  std::vector<int> order;
  for (size_t b = 0; b < loads.size(); ++b)
    for (size_t j = 0; j < loads[b].size(); ++j) order.push_back(loads[b][j].poc);
  std::thread reader([&]() {
    hipSetDevice(device);
    for (size_t n = 0; n < order.size(); ++n) {
      const Free f = free_bufs.get();        // handed back only after the buffer's upload has been ISSUED
      if (f.buf < 0 || stop.load()) break;
      bool ok = !f.ev || hipEventSynchronize(f.ev) == hipSuccess;   // ... and that upload has run (a failed copy must not be overwritten silently)
      const std::chrono::steady_clock::time_point r0 = std::chrono::steady_clock::now();
      if (ok) {
        try { ok = read_luma(order[n], host_bufs_[f.buf]); } catch (...) { ok = false; }   // a throwing source is a failed picture, not std::terminate
      }
      read_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
      Ready r = {ok ? order[n] : -1, f.buf};
      ready.put(r);
      if (!ok) break;
    }
  });
  int rc = HMME_OK;
  std::string what;
  hmme_frame_params fp = {cfg_.search_range, 1, cfg_.bit_depth, 0, n_ctu_};
  const hipStream_t s_copy = (hipStream_t)s_copy_, s_compute = (hipStream_t)s_compute_, s_dl = (hipStream_t)s_download_;
  auto upload = [&](int b) -> int {
    for (size_t j = 0; j < loads[b].size(); ++j) {
      const Ready r = ready.get();
      if (r.poc != loads[b][j].poc) { what = "the picture source failed at picture " + std::to_string(loads[b][j].poc); return HMME_ERR_ARG; }
      const int urc = hmme_plane_upload_async(planes_[loads[b][j].slot], host_bufs_[r.buf], cfg_.width, bps, s_copy);
      if (urc != HMME_OK) { what = std::string("hmme_plane_upload_async: ") + hmme_last_error(ctx_); return urc; }
      if (hipEventRecord(buf_events[r.buf], s_copy) != hipSuccess) { what = "hipEventRecord failed"; return HMME_ERR_DEVICE; }
      Free f = {r.buf, buf_events[r.buf]};
      free_bufs.put(f);                      // the reader waits for this upload before it overwrites the buffer
    }
    return HMME_OK;
  };
  auto launch = [&](int b) -> int {
    const int i0 = batches[b].first, k = batches[b].second - i0;
    const hmme_plane* curs[16];
    const hmme_plane* refs[16];
    for (int i = 0; i < k; ++i) { curs[i] = planes_[where[b][i].first]; refs[i] = planes_[where[b][i].second]; }
    char* mv = (char*)d_mv_ + 4 * per_pair * i0;
    char* sad = (char*)d_sad_ + 4 * per_pair * i0;
    int lrc = hmme_search_pairs_device(ctx_, curs, refs, k, &fp, 0, mv, sad, s_compute);
    if (lrc == HMME_OK && cfg_.refine)
      lrc = hmme_refine_pairs_device(ctx_, curs, refs, k, &fp, 0, mv, 1, (char*)d_qmv_ + 4 * per_pair * i0, (char*)d_cost_ + 4 * per_pair * i0, s_compute);
    if (lrc != HMME_OK) { what = std::string("launch: ") + hmme_last_error(ctx_); return lrc; }
    hipError_t e = hipEventRecord(batch_events[b], s_compute);
    if (cfg_.no_download) return e == hipSuccess ? HMME_OK : HMME_ERR_DEVICE;
    if (e == hipSuccess) e = hipStreamWaitEvent(s_dl, batch_events[b], 0);
    const size_t pair_bytes = 4 * per_pair;
    if (x_mv_) {   // the caller's arrays: every pair has its own place
      for (int i = 0; i < k && e == hipSuccess; ++i) {
        const size_t src = pair_bytes * (i0 + i), dst = pair_bytes * (size_t)x_index_[i0 + i];
        e = hipMemcpyAsync((char*)x_mv_ + dst, (char*)d_mv_ + src, pair_bytes, hipMemcpyDeviceToHost, s_dl);
        if (e == hipSuccess) e = hipMemcpyAsync((char*)x_sad_ + dst, (char*)d_sad_ + src, pair_bytes, hipMemcpyDeviceToHost, s_dl);
        if (e == hipSuccess && cfg_.refine) e = hipMemcpyAsync((char*)x_qmv_ + dst, (char*)d_qmv_ + src, pair_bytes, hipMemcpyDeviceToHost, s_dl);
        if (e == hipSuccess && cfg_.refine) e = hipMemcpyAsync((char*)x_cost_ + dst, (char*)d_cost_ + src, pair_bytes, hipMemcpyDeviceToHost, s_dl);
      }
    } else {
      const size_t bytes = pair_bytes * k, off = pair_bytes * i0;
      if (e == hipSuccess) e = hipMemcpyAsync((char*)h_mv_ + off, mv, bytes, hipMemcpyDeviceToHost, s_dl);
      if (e == hipSuccess) e = hipMemcpyAsync((char*)h_sad_ + off, sad, bytes, hipMemcpyDeviceToHost, s_dl);
      if (e == hipSuccess && cfg_.refine) e = hipMemcpyAsync((char*)h_qmv_ + off, (char*)d_qmv_ + off, bytes, hipMemcpyDeviceToHost, s_dl);
      if (e == hipSuccess && cfg_.refine) e = hipMemcpyAsync((char*)h_cost_ + off, (char*)d_cost_ + off, bytes, hipMemcpyDeviceToHost, s_dl);
    }
    if (e != hipSuccess) { what = std::string("download: ") + hipGetErrorString(e); return HMME_ERR_DEVICE; }
    return HMME_OK;
  };
  const int nb = (int)batches.size();
  if (nb) rc = upload(0);
  for (int b = 0; b < nb && rc == HMME_OK; ++b) {
    rc = launch(b);                           // issued first: the uploads below wait for it only where they evict its planes
    if (rc == HMME_OK && b + 1 < nb) rc = upload(b + 1);
  }
  stop.store(true);
  { Free f = {-1, (hipEvent_t)0}; free_bufs.put(f); }   // wakes a reader that waits for a buffer
  reader.join();
  if (rc == HMME_OK) {
    rc = hmme_upload_status(ctx_, s_copy);
    if (rc != HMME_OK) what = std::string("upload: ") + hmme_last_error(ctx_);
  }
  hipStreamSynchronize(s_copy);
  hipStreamSynchronize(s_compute);
  hipStreamSynchronize(s_dl);
  if (stats) {
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stats->read_seconds = read_s;
    stats->launches = nb; stats->uploads = n_uploads; stats->plane_slots = slots_used;
  }
  return rc == HMME_OK ? HMME_OK : fail(rc, what);
}

LumaReader yuv_file_reader(const std::string& path, int width, int height, int bit_depth, int chroma_format_idc, std::string* err) {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) {
    if (err) *err = "cannot open " + path;
    return LumaReader();
  }
  struct FdOwner { int fd; explicit FdOwner(int f) : fd(f) {} ~FdOwner() { close(fd); } };
  std::shared_ptr<FdOwner> owner(new FdOwner(fd));
  const size_t bps = bit_depth == 8 ? 1 : 2, luma = (size_t)width * height * bps;
  const size_t frame = luma + (chroma_format_idc == 1 ? luma / 2 : 0);
  return [owner, luma, frame](int poc, void* dst) -> bool {
    size_t got = 0;
    while (got < luma) {
      const ssize_t n = pread(owner->fd, (char*)dst + got, luma - got, (off_t)((size_t)poc * frame + got));
      if (n <= 0) return false;               // beyond the end of the file / I/O error
      got += (size_t)n;
    }
    return true;
  };
}

}  // namespace hmme_host

// ---- C entry points of the planner: hm-opencl_amd/hmme/sequence.py binds these with ctypes, so the launch / plane-slot plan exists
// once (round 3 kept a Python twin of plan_plane_loads in step with this one by tests only)
extern "C" {
// pairs: n_pairs x (current POC, reference POC).  Outputs: batch_first[n_batches + 1] (launch b = pairs [batch_first[b],
// batch_first[b + 1])), loads[3 * n_loads] = (launch, POC, slot) in issue order, where[2 * n_pairs] = (slot of the current picture, slot of
// the reference picture) of every pair when its launch runs.  Capacities: batch_first n_pairs + 1, loads 3 * max_loads, where
// 2 * n_pairs.  Returns the number of loads, or -1 with a message in err (too few slots, capacity).
int hmme_host_plan(const int32_t* pairs, int n_pairs, int pairs_per_launch, int n_slots, int32_t* batch_first, int* n_batches,
                   int32_t* loads, int max_loads, int32_t* where, char* err, int err_len) {
  std::vector<std::pair<int, int> > pr;
  for (int i = 0; i < n_pairs; ++i) pr.push_back(std::make_pair((int)pairs[2 * i], (int)pairs[2 * i + 1]));
  const std::vector<std::pair<int, int> > batches = hmme_host::plan_batches(n_pairs, pairs_per_launch);
  std::vector<std::vector<hmme_host::PlaneLoad> > ld;
  std::vector<std::vector<std::pair<int, int> > > wh;
  std::string e;
  bool ok = hmme_host::plan_plane_loads(pr, batches, n_slots, &ld, &wh, &e);
  int n = 0;
  if (ok) {
    for (size_t b = 0; b < ld.size(); ++b) n += (int)ld[b].size();
    if (n > max_loads) { ok = false; e = "load list capacity"; }
  }
  if (!ok) {
    if (err && err_len > 0) snprintf(err, (size_t)err_len, "%s", e.c_str());
    return -1;
  }
  *n_batches = (int)batches.size();
  for (size_t b = 0; b < batches.size(); ++b) batch_first[b] = batches[b].first;
  batch_first[batches.size()] = n_pairs;
  int k = 0;
  for (size_t b = 0; b < ld.size(); ++b)
    for (size_t j = 0; j < ld[b].size(); ++j) { loads[3 * k] = (int32_t)b; loads[3 * k + 1] = ld[b][j].poc; loads[3 * k + 2] = ld[b][j].slot; ++k; }
  for (size_t b = 0; b < wh.size(); ++b)
    for (size_t i = 0; i < wh[b].size(); ++i) { where[2 * (batches[b].first + i)] = wh[b][i].first; where[2 * (batches[b].first + i) + 1] = wh[b][i].second; }
  return n;
}
}  // extern "C"

