// Interface of the fused persistent FISTA kernel (fc_fused.hip).
#pragma once
#include "common.h"
#include <vector>

namespace vtc {

// Shapes the fused bf16 / bf16x3 kernel is built for.
bool fused_shape_supported(int64_t b, int64_t n, int64_t s, int precision);
size_t fused_workspace_bytes(int64_t b, int64_t n, int64_t s, int precision);
// eta_dev != nullptr: the step size is read from device memory (eta ignored).
int run_fused(const float* images, const float* dictionary,
              const float* initial_codes, float* codes, int64_t b, int64_t n,
              int64_t s, float eta, const float* eta_dev,
              float sparsity_weight, int num_iters, int variant,
              int threshold, int precision, void* workspace,
              size_t workspace_bytes, int* iters_run, hipStream_t st);
// iterations per call the fused kernels' momentum table covers, and the table
// itself (one copy per device, created on first use there)
int fused_max_iters();
const float* fista_beta_table_on_this_device();

// host-side FISTA momentum schedule (fc_inference.hip)
void fista_betas(int num_iters, std::vector<float>* out);

}  // namespace vtc
