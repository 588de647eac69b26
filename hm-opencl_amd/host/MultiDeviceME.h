// MultiDeviceME.h -- open-loop motion estimation of one sequence on N GPUs of a node from ONE C++ process (the reference's
// host creates its OpenCL context over all GPUs it finds, TEncOpenCL.cpp:128-129, and drives them through one queue, :185).
//
// Picture pair p of the sequence is searched by device p mod N (the same shard rule as hmme/shard.py): one hmme context, one
// SequenceSearch (SequenceME.h: reader thread, plane ring, copy / compute / download streams) and one host thread per device; no
// exchange during the search.  The one exchange step is the gather of the result tables into rank 0's memory:
//   kGatherRccl   every device sends its tables to device 0 over RCCL -- ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on
//                 communicators made by ncclCommInitAll (xGMI between the GPUs) -- and device 0 writes them, in pair order, into
//                 page-locked host memory.  Needs N distinct devices (RCCL refuses a communicator with one GPU twice).
//   kGatherPeer   hipMemcpyPeerAsync from each device's tables into device 0's gather buffer (the same bytes over the same links,
//                 without the communicator); also works when several contexts share one device, which is how the N > 1 path is
//                 exercised on a one-GPU box.
//   kGatherHost   every device's download stream writes its tables straight into its place in the page-locked result (no device-to-
//                 device hop at all: the tables are consumed on the host).
// Link: -lhmme_multi -lhmme_host -lhmme -lrccl -lamdhip64.
#ifndef HMME_MULTI_DEVICE_ME_H
#define HMME_MULTI_DEVICE_ME_H

#include <string>
#include <utility>
#include <vector>

#include "SequenceME.h"

namespace hmme_host {

enum GatherVia { kGatherRccl = 0, kGatherPeer = 1, kGatherHost = 2 };

struct MultiDeviceStats {
  double seconds;                        // wall clock of run(): first read to last table in the gathered result
  double search_seconds;                 // ... up to the slowest device's last table (before the gather)
  double gather_seconds;                 // the exchange step alone
  size_t gather_bytes;                   // bytes that travelled to device 0 / rank 0's memory from the OTHER devices
  std::vector<int> pairs_per_device;     // load balance: 124 pairs on 8 devices are 16, 16, 16, 16, 15, 15, 15, 15
  std::vector<double> device_seconds;    // each device's own search time
  std::vector<std::string> device_info;  // hmme_device_info of each context
};

// pairs of `n_pairs` that device `rank` of `world` searches: rank, rank + world, ...
std::vector<int> pairs_for_device(int n_pairs, int rank, int world);

class MultiDeviceSearch {
 public:
  // devices: HIP device index per rank (rank 0 = devices[0] owns the gathered result); an index may repeat (several contexts on
  // one GPU: rehearsal only, kGatherRccl is then refused)
  MultiDeviceSearch(const std::vector<int>& devices, const SequenceConfig& cfg, GatherVia via, double lambda);
  ~MultiDeviceSearch();
  // HMME_OK or a negative HMME_ERR_*; error() has the text.  make_reader(rank) returns the picture source of that rank's reader thread
  // (each rank reads the pictures of its own pairs).
  int run(const std::vector<std::pair<int, int> >& pairs, const std::function<LumaReader(int)>& make_reader, MultiDeviceStats* stats);
  // gathered tables in PAIR order, page-locked, valid until the next run(): mv int16 [n_pairs][n_ctu][593][2], sad uint32 [n_pairs][n_ctu][593]
  // (+ qmv / cost with cfg.refine)
  const int16_t* mv() const { return h_mv_; }
  const uint32_t* sad() const { return h_sad_; }
  const int16_t* qmv() const { return h_qmv_; }
  const uint32_t* cost() const { return h_cost_; }
  int num_ctus() const { return n_ctu_; }
  int world() const { return (int)devices_.size(); }
  const std::string& error() const { return err_; }

 private:
  MultiDeviceSearch(const MultiDeviceSearch&);
  MultiDeviceSearch& operator=(const MultiDeviceSearch&);
  int fail(int code, const std::string& what);
  int init();
  void release();
  std::vector<int> devices_;
  SequenceConfig cfg_;
  GatherVia via_;
  double lambda_;
  int n_ctu_;
  std::string err_;
  std::vector<hmme_ctx*> ctx_;
  std::vector<SequenceSearch*> seq_;
  std::vector<void*> comms_;     // ncclComm_t per rank (kGatherRccl)
  std::vector<void*> streams_;   // hipStream_t per rank for the exchange step
  void* d_gather_[4];            // device 0: tables of all pairs, rank-major (mv, sad, qmv, cost)
  int16_t *h_mv_, *h_qmv_;
  uint32_t *h_sad_, *h_cost_;
  size_t cap_pairs_;
};

}  // namespace hmme_host

#endif
