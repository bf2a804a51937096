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

// 12x12 patches (n = 144) against 288 / 576 atoms, 8x8 patches against 256 /
// 512: state in registers, the dictionary streamed from L2 (fc_chip16.hip);
// workspace = one packed copy
bool chip16_shape_supported(int64_t n, int64_t s);
size_t chip16_workspace_bytes(int64_t n, int64_t s);
int run_chip16(const float* images, const float* dictionary,
               const float* initial_codes, float* codes, int64_t b, int64_t n,
               int64_t s, float eta, const float* eta_dev,
               float sparsity_weight, int num_iters, int variant,
               int threshold, void* workspace, size_t workspace_bytes,
               int* iters_run, hipStream_t st);

}  // namespace vtc
