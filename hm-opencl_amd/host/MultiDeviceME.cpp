// MultiDeviceME.cpp -- see MultiDeviceME.h.  include/hmme.h, the HIP runtime API and RCCL; no torch, no Python.
#include "MultiDeviceME.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <set>
#include <thread>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

namespace hmme_host {

std::vector<int> pairs_for_device(int n_pairs, int rank, int world) {
  std::vector<int> v;
  for (int p = rank; p < n_pairs; p += world) v.push_back(p);
  return v;
}

MultiDeviceSearch::MultiDeviceSearch(const std::vector<int>& devices, const SequenceConfig& cfg, GatherVia via, double lambda)
    : devices_(devices), cfg_(cfg), via_(via), lambda_(lambda), n_ctu_(hmme_num_ctus(cfg.width, cfg.height)), h_mv_(0), h_qmv_(0), h_sad_(0),
      h_cost_(0), cap_pairs_(0) {
  for (int t = 0; t < 4; ++t) d_gather_[t] = 0;
  cfg_.no_download = via_ != kGatherHost;   // device-side gathers take the tables from device memory
}

MultiDeviceSearch::~MultiDeviceSearch() { release(); }

void MultiDeviceSearch::release() {
  for (size_t r = 0; r < seq_.size(); ++r) delete seq_[r];
  seq_.clear();
  for (size_t r = 0; r < comms_.size(); ++r)
    if (comms_[r]) ncclCommDestroy((ncclComm_t)comms_[r]);
  comms_.clear();
  for (size_t r = 0; r < streams_.size(); ++r)
    if (streams_[r]) { hipSetDevice(devices_[r]); hipStreamDestroy((hipStream_t)streams_[r]); }
  streams_.clear();
  if (!devices_.empty()) hipSetDevice(devices_[0]);
  for (int t = 0; t < 4; ++t) { hipFree(d_gather_[t]); d_gather_[t] = 0; }
  if (h_mv_) hipHostFree(h_mv_);
  if (h_sad_) hipHostFree(h_sad_);
  if (h_qmv_) hipHostFree(h_qmv_);
  if (h_cost_) hipHostFree(h_cost_);
  h_mv_ = h_qmv_ = 0; h_sad_ = h_cost_ = 0;
  cap_pairs_ = 0;
  for (size_t r = 0; r < ctx_.size(); ++r)
    if (ctx_[r]) hmme_destroy(ctx_[r]);
  ctx_.clear();
}

int MultiDeviceSearch::fail(int code, const std::string& what) {
  err_ = what;
  fprintf(stderr, "hmme MultiDeviceSearch: ERROR: %s\n", what.c_str());
  return code;
}

#define MD_HIP(call)                                                                                          \
  do {                                                                                                        \
    const hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) return fail(HMME_ERR_DEVICE, std::string(#call " -> ") + hipGetErrorString(e_));    \
  } while (0)
#define MD_NCCL(call)                                                                                         \
  do {                                                                                                        \
    const ncclResult_t r_ = (call);                                                                           \
    if (r_ != ncclSuccess) return fail(HMME_ERR_DEVICE, std::string(#call " -> ") + ncclGetErrorString(r_));  \
  } while (0)

int MultiDeviceSearch::init() {
  const int world = (int)devices_.size();
  ctx_.assign(world, (hmme_ctx*)0);
  seq_.assign(world, (SequenceSearch*)0);
  streams_.assign(world, (void*)0);
  for (int r = 0; r < world; ++r) {
    if (hmme_create(devices_[r], std::max(64, cfg_.search_range), 0, &ctx_[r]) != HMME_OK)
      return fail(HMME_ERR_DEVICE, std::string("hmme_create on device ") + std::to_string(devices_[r]) + ": " + hmme_last_error(0));
    hmme_set_lambda(ctx_[r], lambda_);
    seq_[r] = new SequenceSearch(ctx_[r], cfg_);
    MD_HIP(hipSetDevice(devices_[r]));
    hipStream_t s;
    MD_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    streams_[r] = s;
  }
  if (via_ == kGatherRccl) {
    std::vector<ncclComm_t> c(world);
    MD_NCCL(ncclCommInitAll(c.data(), world, devices_.data()));
    comms_.assign(c.begin(), c.end());
  }
  if (via_ == kGatherPeer)
    for (int r = 1; r < world; ++r)
      if (devices_[r] != devices_[0]) {   // direct xGMI copies into device 0's buffer; "already enabled" is fine
        MD_HIP(hipSetDevice(devices_[r]));
        const hipError_t e = hipDeviceEnablePeerAccess(devices_[0], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(HMME_ERR_DEVICE, std::string("peer access -> ") + hipGetErrorString(e));
        (void)hipGetLastError();
      }
  return HMME_OK;
}

int MultiDeviceSearch::run(const std::vector<std::pair<int, int> >& pairs, const std::function<LumaReader(int)>& make_reader,
                           MultiDeviceStats* stats) {
  const int world = (int)devices_.size(), n_pairs = (int)pairs.size();
  if (world < 1) return fail(HMME_ERR_ARG, "no devices");
  if (!make_reader) return fail(HMME_ERR_ARG, "no picture source");
  if (via_ == kGatherRccl && std::set<int>(devices_.begin(), devices_.end()).size() != devices_.size())
    return fail(HMME_ERR_ARG, "kGatherRccl needs distinct devices (RCCL refuses one GPU twice in a communicator); use kGatherPeer for a rehearsal");
  const size_t per_pair = (size_t)n_ctu_ * HMME_NUM_CTU_PARTS, pair_bytes = 4 * per_pair;
  const int n_tables = cfg_.refine ? 4 : 2;
  if (n_pairs == 0) {   // nothing to search: an empty result, not an allocation of zero bytes
    if (stats) { *stats = MultiDeviceStats(); stats->pairs_per_device.assign(world, 0); stats->device_seconds.assign(world, 0.0); }
    return HMME_OK;
  }

  // ---- one context, one sequence driver, one exchange stream per rank (kept between runs; a failed set-up leaves nothing behind)
  if (ctx_.empty()) {
    const int rc = init();
    if (rc != HMME_OK) { const std::string keep = err_; release(); err_ = keep; return rc; }
  }
  MD_HIP(hipSetDevice(devices_[0]));
  if ((size_t)n_pairs > cap_pairs_) {
    for (int t = 0; t < 4; ++t) { hipFree(d_gather_[t]); d_gather_[t] = 0; }
    if (h_mv_) hipHostFree(h_mv_);
    if (h_sad_) hipHostFree(h_sad_);
    if (h_qmv_) hipHostFree(h_qmv_);
    if (h_cost_) hipHostFree(h_cost_);
    h_mv_ = h_qmv_ = 0; h_sad_ = h_cost_ = 0; cap_pairs_ = 0;
    // hipHostMallocPortable: every device's download stream may write into the one result (kGatherHost)
    MD_HIP(hipHostMalloc((void**)&h_mv_, pair_bytes * n_pairs, hipHostMallocPortable));
    MD_HIP(hipHostMalloc((void**)&h_sad_, pair_bytes * n_pairs, hipHostMallocPortable));
    if (cfg_.refine) {
      MD_HIP(hipHostMalloc((void**)&h_qmv_, pair_bytes * n_pairs, hipHostMallocPortable));
      MD_HIP(hipHostMalloc((void**)&h_cost_, pair_bytes * n_pairs, hipHostMallocPortable));
    }
    if (via_ != kGatherHost)
      for (int t = 0; t < n_tables; ++t) MD_HIP(hipMalloc(&d_gather_[t], pair_bytes * n_pairs));
    cap_pairs_ = n_pairs;
  }

  // ---- the search: every rank its own pairs, on its own host thread, nothing exchanged
  std::vector<std::vector<int> > mine(world);
  std::vector<std::vector<std::pair<int, int> > > my_pairs(world);
  for (int r = 0; r < world; ++r) {
    mine[r] = pairs_for_device(n_pairs, r, world);
    for (size_t i = 0; i < mine[r].size(); ++i) my_pairs[r].push_back(pairs[mine[r][i]]);
    if (via_ == kGatherHost) seq_[r]->set_host_output(h_mv_, h_sad_, h_qmv_, h_cost_, mine[r]);   // straight into pair order
  }
  std::vector<int> rcs(world, HMME_OK);
  std::vector<SequenceStats> st(world);
  const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  {
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
      th.push_back(std::thread([&, r]() {
        if (my_pairs[r].empty()) { st[r].seconds = 0; return; }
        rcs[r] = seq_[r]->run(my_pairs[r], make_reader(r), &st[r]);
      }));
    for (int r = 0; r < world; ++r) th[r].join();
  }
  for (int r = 0; r < world; ++r)
    if (rcs[r] != HMME_OK) return fail(rcs[r], "rank " + std::to_string(r) + ": " + seq_[r]->error());
  const std::chrono::steady_clock::time_point t1 = std::chrono::steady_clock::now();

  // ---- the one exchange step: tables of ranks 1.. into device 0's gather buffer (rank-major), then into the host result in pair order
  size_t moved = 0;
  if (via_ != kGatherHost) {
    std::vector<size_t> first(world + 1, 0);   // first table of each rank in the rank-major gather buffer
    for (int r = 0; r < world; ++r) first[r + 1] = first[r] + mine[r].size();
    const void* src[4];
    if (via_ == kGatherRccl) {
      // one group: rank r > 0 sends its four tables, rank 0 posts the matching receives (send i of a peer pairs with receive i)
      MD_NCCL(ncclGroupStart());
      for (int r = 1; r < world; ++r) {
        if (mine[r].empty()) continue;
        src[0] = seq_[r]->d_mv(); src[1] = seq_[r]->d_sad(); src[2] = seq_[r]->d_qmv(); src[3] = seq_[r]->d_cost();
        const size_t count = per_pair * mine[r].size();   // 32-bit words (a TComMv is one word)
        for (int t = 0; t < n_tables; ++t) {
          MD_NCCL(ncclSend(src[t], count, ncclInt32, 0, (ncclComm_t)comms_[r], (hipStream_t)streams_[r]));
          MD_NCCL(ncclRecv((char*)d_gather_[t] + pair_bytes * first[r], count, ncclInt32, r, (ncclComm_t)comms_[0], (hipStream_t)streams_[0]));
          moved += 4 * count;
        }
      }
      MD_NCCL(ncclGroupEnd());
      for (int r = 1; r < world; ++r) { MD_HIP(hipSetDevice(devices_[r])); MD_HIP(hipStreamSynchronize((hipStream_t)streams_[r])); }
    } else {
      for (int r = 1; r < world; ++r) {
        if (mine[r].empty()) continue;
        src[0] = seq_[r]->d_mv(); src[1] = seq_[r]->d_sad(); src[2] = seq_[r]->d_qmv(); src[3] = seq_[r]->d_cost();
        const size_t bytes = pair_bytes * mine[r].size();
        MD_HIP(hipSetDevice(devices_[r]));
        for (int t = 0; t < n_tables; ++t) {
          MD_HIP(hipMemcpyPeerAsync((char*)d_gather_[t] + pair_bytes * first[r], devices_[0], src[t], devices_[r], bytes, (hipStream_t)streams_[r]));
          moved += bytes;
        }
      }
      for (int r = 1; r < world; ++r) { MD_HIP(hipSetDevice(devices_[r])); MD_HIP(hipStreamSynchronize((hipStream_t)streams_[r])); }
    }
    // device 0 -> page-locked host result, pair by pair into pair order (rank 0's own tables straight from its search buffers)
    MD_HIP(hipSetDevice(devices_[0]));
    const hipStream_t s0 = (hipStream_t)streams_[0];
    void* dst[4] = {h_mv_, h_sad_, h_qmv_, h_cost_};
    for (int r = 0; r < world; ++r) {
      src[0] = seq_[0]->d_mv(); src[1] = seq_[0]->d_sad(); src[2] = seq_[0]->d_qmv(); src[3] = seq_[0]->d_cost();
      for (size_t i = 0; i < mine[r].size(); ++i)
        for (int t = 0; t < n_tables; ++t) {
          const char* from = r == 0 ? (const char*)src[t] + pair_bytes * i : (const char*)d_gather_[t] + pair_bytes * (first[r] + i);
          MD_HIP(hipMemcpyAsync((char*)dst[t] + pair_bytes * mine[r][i], from, pair_bytes, hipMemcpyDeviceToHost, s0));
        }
    }
    MD_HIP(hipStreamSynchronize(s0));
  } else {
    for (int r = 1; r < world; ++r) moved += pair_bytes * n_tables * mine[r].size();   // what ranks 1.. wrote into the result (their download streams)
  }
  const std::chrono::steady_clock::time_point t2 = std::chrono::steady_clock::now();
  if (stats) {
    stats->seconds = std::chrono::duration<double>(t2 - t0).count();
    stats->search_seconds = std::chrono::duration<double>(t1 - t0).count();
    stats->gather_seconds = std::chrono::duration<double>(t2 - t1).count();
    stats->gather_bytes = moved;
    stats->pairs_per_device.clear(); stats->device_seconds.clear(); stats->device_info.clear();
    for (int r = 0; r < world; ++r) {
      stats->pairs_per_device.push_back((int)mine[r].size());
      stats->device_seconds.push_back(st[r].seconds);
      stats->device_info.push_back(hmme_device_info(ctx_[r]));
    }
  }
  return HMME_OK;
}

}  // namespace hmme_host
