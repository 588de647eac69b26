/* hmme_test.h -- entry points of libhmme.so that exist for tests and measurements only.  NOT part of the drop-in boundary
 * (include/hmme.h): nothing a host application needs, no stability promise, not counted in HMME_ABI_VERSION. */
#ifndef HMME_TEST_H
#define HMME_TEST_H

#include "hmme.h"

#ifdef __cplusplus
extern "C" {
#endif

/* device address of a plane's sample (0,0) (plane != NULL) or of the context's per-CTU current-block staging area: lets a test
 * prove that a launch ran on addresses whose low dword has bit 31 set (tests/test_gpu_parity.py, high-address case) */
uint64_t hmme_test_device_address(const hmme_ctx* ctx, const hmme_plane* plane);
/* average device time in ms of the search kernel(s) alone over `reps` back-to-back launches on `stream` (job tables prepared once,
 * outside the timed region), measured with hipEvents recorded on that stream */
int hmme_test_time_search_kernel(hmme_ctx* ctx, const hmme_plane* cur, const hmme_plane* ref, const hmme_frame_params* fp,
                                 const void* d_pred_q, void* d_out_mv, void* d_out_sad, void* stream, int reps,
                                 float* avg_ms);

/* how an 8-bit whole-picture search of n_pairs pictures of width x height at `search_range` (<= 64) is dealt to a chip of `slots` workgroup slots
 * (hmme.hip prep_jobs; host arithmetic, needs no device): out[0] = jobs, out[1] = jobs searched whole (the head; == jobs: no tail plan),
 * out[2] = workgroups (segments) of the tail, out[3] = 1 if head and tail are one launch */
int hmme_test_tail_plan(int width, int height, int search_range, int n_pairs, int slots, int* out);

/* which job the k-th workgroup of a refinement launch over `n_pairs` whole pictures of width x height takes (me_frac_deal, me_kernels.hpp:
 * edge CTUs of every pair first, then the interiors); host code, needs no device.  -1 for k outside the launch */
int hmme_test_frac_deal(int k, int n_pairs, int width, int height);

#ifdef __cplusplus
}
#endif
#endif
