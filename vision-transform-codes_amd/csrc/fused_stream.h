// Interface of the fused persistent kernel with streamed state
// (fused_stream.hip): 16x16 patches against dictionaries too large for the
// on-chip state of fc_fused.hip -- the subspace plugin's padded (G*m, n)
// dictionary (configs[3]: 4096 slots) and fully-connected dictionaries beyond
// 1024 atoms.
#pragma once
#include "common.h"

namespace vtc {

// n == 256, slots a multiple of 256, group size m in {1, 2, 4, 8},
// precision VTC_BF16X3 or VTC_F16X3
bool stream_shape_supported(int64_t b, int64_t n, int64_t slots, int64_t m,
                            int precision);
size_t stream_workspace_bytes(int64_t b, int64_t n, int64_t slots,
                              int precision);
// m == 1: element-wise `threshold` (vtc_threshold); m > 1: group soft
// threshold over m adjacent slots (subspace_ista_fista.py:149-156).
// dictionary (slots, 256); initial (b, slots) or null; codes (b, slots) out.
// eta_dev != nullptr: step size read from device memory.
int run_stream(const float* images, const float* dictionary,
               const float* initial, float* codes, int64_t b, int64_t n,
               int64_t slots, int64_t m, float eta, const float* eta_dev,
               float sparsity_weight, int num_iters, int variant,
               int threshold, int precision, void* workspace,
               size_t workspace_bytes, int* iters_run, hipStream_t st);

// iterations per call the momentum table covers
int fused_max_iters_for_stream();

}  // namespace vtc
