// SequenceME.h -- open-loop motion estimation of a whole sequence from C++, over the C ABI (include/hmme.h) and the HIP
// runtime only: the C++ twin of hm-opencl_amd/hmme/sequence.py (the Python driver of BASELINE config 4), for a host
// application that is C++ like the reference encoder -- a look-ahead / pre-analysis stage that searches the (current, reference)
// picture pairs of a GOP (cfg/encoder_randomaccess_main.cfg:28-31, cfg/encoder_lowdelay_P_main.cfg:24-27) ahead of the
// CTU loop, where the reference calls the GPU once per CTU from inside it (TEncSearch.cpp:3743-3771).
//
// Pictures stream from a source through a ring of plane slots: a reader thread fills page-locked host buffers, a copy stream
// uploads the pictures of the next launch (hmme_plane_upload_async) while the compute stream searches -- and optionally refines --
// the current one (hmme_search_pairs_device / hmme_refine_pairs_device, up to 16 pairs per launch), and a third stream brings the
// tables back into page-locked host memory.  The library orders plane refills against searches that still read the old contents.
#ifndef HMME_SEQUENCE_ME_H
#define HMME_SEQUENCE_ME_H

#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "../../include/hmme.h"

namespace hmme_host {

struct PlaneLoad { int poc, slot; };

// launches of at most `pairs_per_launch` (<= 16) consecutive pairs: [first, last) index ranges
std::vector<std::pair<int, int> > plan_batches(int n_pairs, int pairs_per_launch);
// Which picture is uploaded into which of n_slots planes before which launch (same rule as hmme/sequence.py plan_plane_loads:
// a slot whose picture the launch does not read is reused; prefer one the previous launch does not read either, then the one whose
// next use lies farthest ahead).  loads[b]: uploads before launch b; where[b][i]: (cur slot, ref slot) of pair first + i.
// Returns false (err set) if a launch needs more pictures than there are slots.
bool plan_plane_loads(const std::vector<std::pair<int, int> >& pairs, const std::vector<std::pair<int, int> >& batches, int n_slots,
                      std::vector<std::vector<PlaneLoad> >* loads, std::vector<std::vector<std::pair<int, int> > >* where, std::string* err);

struct SequenceConfig {
  int width, height;
  int bit_depth;         // 8: 1-byte samples, u8 planes; 9..12: 2-byte little-endian samples, u16 planes
  int search_range;
  int pairs_per_launch;  // 1..16
  int plane_slots;       // >= 2 * pairs_per_launch is always enough; 0 = max(8, 2 * pairs_per_launch + 2)
  int host_buffers;      // page-locked picture buffers between reader and copy stream; 0 = 4
  bool refine;           // also run xPatternSearchFracDIF (Hadamard) for every slot
  bool no_download;      // leave the tables in device memory (d_mv() ...): a multi-device driver gathers them on the device side
};

struct SequenceStats {
  double seconds;        // wall clock of run(), first read to last table on the host
  double read_seconds;   // time the reader thread spent inside read_luma
  int launches, uploads, plane_slots;
};

// read_luma(poc, dst): fill width * height samples (1 or 2 bytes each, see bit_depth) of picture `poc`; false = failure.
// Called from a reader thread, in upload order.
typedef std::function<bool(int, void*)> LumaReader;

// Searches `pairs` ((current POC, reference POC), searched in this order).  Tables land in page-locked memory owned by the
// object, valid until the next run() / destruction: mv() int16 [n_pairs][n_ctu][593][2], sad() uint32 [n_pairs][n_ctu][593], and with
// refine qmv() / cost() likewise.  Not thread-safe; one object per context.
class SequenceSearch {
 public:
  SequenceSearch(hmme_ctx* ctx, const SequenceConfig& cfg);
  ~SequenceSearch();
  // HMME_OK or a negative HMME_ERR_*; error() has the text
  int run(const std::vector<std::pair<int, int> >& pairs, const LumaReader& read_luma, SequenceStats* stats);
  const int16_t* mv() const { return h_mv_; }
  const uint32_t* sad() const { return h_sad_; }
  const int16_t* qmv() const { return h_qmv_; }
  const uint32_t* cost() const { return h_cost_; }
  // the same tables in device memory, pairs in the order given to run() (valid until the next run() / destruction)
  const void* d_mv() const { return d_mv_; }
  const void* d_sad() const { return d_sad_; }
  const void* d_qmv() const { return d_qmv_; }
  const void* d_cost() const { return d_cost_; }
  // Download into the caller's page-locked arrays instead of this object's own: pair i of run() lands at table index dest_index[i]
  // (a multi-device driver gives every device its places in ONE result in global pair order).  The arrays must stay valid and
  // page-locked (hipHostMalloc / hmme_host_register) during run(); qmv / cost may be null without refine.
  void set_host_output(int16_t* mv, uint32_t* sad, int16_t* qmv, uint32_t* cost, const std::vector<int>& dest_index);
  int num_ctus() const { return n_ctu_; }
  const std::string& error() const { return err_; }

 private:
  SequenceSearch(const SequenceSearch&);
  SequenceSearch& operator=(const SequenceSearch&);
  int fail(int code, const std::string& what);
  void release();
  hmme_ctx* ctx_;
  SequenceConfig cfg_;
  int n_ctu_;
  std::string err_;
  std::vector<hmme_plane*> planes_;
  std::vector<void*> host_bufs_;
  void *s_copy_, *s_compute_, *s_download_;   // hipStream_t
  void *d_mv_, *d_sad_, *d_qmv_, *d_cost_;
  int16_t *h_mv_, *h_qmv_;
  uint32_t *h_sad_, *h_cost_;
  size_t cap_pairs_;
  int16_t *x_mv_, *x_qmv_;         // external output (set_host_output), or null
  uint32_t *x_sad_, *x_cost_;
  std::vector<int> x_index_;
};

// planar YUV file reader for SequenceSearch: luma of picture `poc` (4:2:0 / 4:0:0; 8-bit or 16-bit little-endian samples,
// TVideoIOYuv.cpp:247, :680); thread-safe (pread)
LumaReader yuv_file_reader(const std::string& path, int width, int height, int bit_depth, int chroma_format_idc /* 0 = 400, 1 = 420 */,
                           std::string* err);

}  // namespace hmme_host

#endif
