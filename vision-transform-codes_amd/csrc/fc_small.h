// Interface of the on-chip ISTA/FISTA kernel for 8x8 patches (fc_small.hip).
#pragma once
#include "common.h"

namespace vtc {

// n == 64 pixels, s in {64, 128, 192} atoms: dictionary and state stay
// on the CU
bool small_shape_supported(int64_t n, int64_t s);
// exact-f32 arithmetic (f32 MFMA); eta_dev != nullptr: step size read from
// device memory.  No workspace.
int run_small(const float* images, const float* dictionary,
              const float* initial_codes, float* codes, int64_t b, int64_t n,
              int64_t s, float eta, const float* eta_dev,
              float sparsity_weight, int num_iters, int variant,
              int threshold, int* iters_run, hipStream_t st);

}  // namespace vtc
