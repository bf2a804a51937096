// Device reductions behind the validation metrics of the trainer
// (training/sparse_coding.py:177-229 `compute_metrics`): LASSO loss terms,
// normalised L0, pSNR inputs, dictionary change.  The residual itself comes
// from vtc_fc_residual / vtc_conv_residual; everything here is a row
// reduction with a fixed summation order (a block per row, strided partials,
// LDS tree), so results do not depend on scheduling.
#include "common.h"

namespace vtc {

constexpr int kStatThreads = 256;

__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float total = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) total += scratch[w];
  return total;
}

// per row: sum x^2, sum |x|, count x != 0   (any output may be null)
__global__ __launch_bounds__(kStatThreads) void row_stats_kernel(
    const float* __restrict__ x, int64_t cols, float* __restrict__ sumsq,
    float* __restrict__ l1, float* __restrict__ l0) {
  __shared__ float scratch[kStatThreads / 64];
  const float* row = x + (int64_t)blockIdx.x * cols;
  float sq = 0.f, ab = 0.f, nz = 0.f;
  for (int64_t j = threadIdx.x; j < cols; j += kStatThreads) {
    const float v = row[j];
    sq = fmaf(v, v, sq);
    ab += fabsf(v);
    nz += (v != 0.f) ? 1.f : 0.f;
  }
  const float tsq = block_sum(sq, scratch);
  const float tab = block_sum(ab, scratch);
  const float tnz = block_sum(nz, scratch);
  if (threadIdx.x == 0) {
    if (sumsq) sumsq[blockIdx.x] = tsq;
    if (l1) l1[blockIdx.x] = tab;
    if (l0) l0[blockIdx.x] = tnz;
  }
}

// out[r] = sum_g || codes[r, group g] ||_2   (padded group tables)
__global__ __launch_bounds__(kStatThreads) void group_norm_sum_kernel(
    const float* __restrict__ codes, const int32_t* __restrict__ index,
    const uint8_t* __restrict__ valid, int64_t s, int64_t groups, int m,
    float* __restrict__ out) {
  __shared__ float scratch[kStatThreads / 64];
  const float* row = codes + (int64_t)blockIdx.x * s;
  float acc = 0.f;
  for (int64_t g = threadIdx.x; g < groups; g += kStatThreads) {
    float sq = 0.f;
    for (int j = 0; j < m; ++j) {
      if (!valid[g * m + j]) continue;
      const float v = row[index[g * m + j]];
      sq = fmaf(v, v, sq);
    }
    acc += sqrtf(sq);
  }
  const float total = block_sum(acc, scratch);
  if (threadIdx.x == 0) out[blockIdx.x] = total;
}

// min / max over a strided window: outer x rows x cols elements,
// element (o, r, c) at x[o * outer_pitch + r * row_pitch + c]; out = {min, max}
__global__ __launch_bounds__(kStatThreads) void window_minmax_partial_kernel(
    const float* __restrict__ x, int64_t outer, int64_t rows, int64_t cols,
    int64_t outer_pitch, int64_t row_pitch, float* __restrict__ partial) {
  __shared__ float smin[kStatThreads / 64], smax[kStatThreads / 64];
  const int64_t total = outer * rows * cols;
  const int64_t stride = (int64_t)gridDim.x * kStatThreads;
  float lo = INFINITY, hi = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * kStatThreads + threadIdx.x; i < total;
       i += stride) {
    const int64_t o = i / (rows * cols), rem = i % (rows * cols);
    const float v = x[o * outer_pitch + (rem / cols) * row_pitch + rem % cols];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off, 64));
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { smin[wave] = lo; smax[wave] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kStatThreads / 64; ++w) {
      lo = fminf(lo, smin[w]);
      hi = fmaxf(hi, smax[w]);
    }
    partial[2 * blockIdx.x] = lo;
    partial[2 * blockIdx.x + 1] = hi;
  }
}

__global__ void minmax_final_kernel(const float* __restrict__ partial,
                                    int count, float* __restrict__ out) {
  float lo = INFINITY, hi = -INFINITY;
  for (int i = threadIdx.x; i < count; i += 64) {
    lo = fminf(lo, partial[2 * i]);
    hi = fmaxf(hi, partial[2 * i + 1]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off, 64));
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
  }
  if (threadIdx.x == 0) { out[0] = lo; out[1] = hi; }
}

// out[r] = mean_c | a[r, c] - b[r, c] |
__global__ __launch_bounds__(kStatThreads) void rows_mean_abs_diff_kernel(
    const float* __restrict__ a, const float* __restrict__ b, int64_t cols,
    float* __restrict__ out) {
  __shared__ float scratch[kStatThreads / 64];
  const int64_t base = (int64_t)blockIdx.x * cols;
  float acc = 0.f;
  for (int64_t j = threadIdx.x; j < cols; j += kStatThreads)
    acc += fabsf(sub_rn(a[base + j], b[base + j]));
  const float total = block_sum(acc, scratch);
  if (threadIdx.x == 0) out[blockIdx.x] = total / (float)cols;
}

constexpr int kMinMaxBlocks = 512;

}  // namespace vtc

using namespace vtc;

extern "C" int vtc_row_stats(const float* x, int64_t rows, int64_t cols,
                             float* sumsq, float* l1, float* l0,
                             void* stream) {
  VTC_REQUIRE(x || rows == 0, "vtc_row_stats: null pointer");
  VTC_REQUIRE(rows >= 0 && cols > 0 && rows <= 0x7fffffffLL,
              "vtc_row_stats: bad sizes");
  if (rows == 0) return VTC_OK;
  hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)rows),
                     dim3(kStatThreads), 0, as_stream(stream), x, cols, sumsq,
                     l1, l0);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

extern "C" int vtc_group_norm_sum(const float* codes, const int32_t* index,
                                  const uint8_t* valid, float* out, int64_t b,
                                  int64_t s, int64_t groups, int64_t m,
                                  void* stream) {
  VTC_REQUIRE((codes && index && valid && out) || b == 0,
              "vtc_group_norm_sum: null pointer");
  VTC_REQUIRE(b >= 0 && s > 0 && groups > 0 && m > 0 && b <= 0x7fffffffLL,
              "vtc_group_norm_sum: bad sizes");
  if (b == 0) return VTC_OK;
  hipLaunchKernelGGL(group_norm_sum_kernel, dim3((unsigned)b),
                     dim3(kStatThreads), 0, as_stream(stream), codes, index,
                     valid, s, groups, (int)m, out);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

extern "C" size_t vtc_window_minmax_workspace_bytes(void) {
  return (size_t)kMinMaxBlocks * 2 * sizeof(float);
}

extern "C" int vtc_window_minmax(const float* x, int64_t outer, int64_t rows,
                                 int64_t cols, int64_t outer_pitch,
                                 int64_t row_pitch, float* out_min_max,
                                 void* workspace, size_t workspace_bytes,
                                 void* stream) {
  VTC_REQUIRE(x && out_min_max, "vtc_window_minmax: null pointer");
  VTC_REQUIRE(outer > 0 && rows > 0 && cols > 0,
              "vtc_window_minmax: empty window");
  if (!workspace || workspace_bytes < vtc_window_minmax_workspace_bytes()) {
    set_error("vtc_window_minmax: workspace too small");
    return VTC_ERR_WORKSPACE;
  }
  const int64_t total = outer * rows * cols;
  int64_t blocks = ceil_div(total, kStatThreads);
  if (blocks > kMinMaxBlocks) blocks = kMinMaxBlocks;
  float* partial = static_cast<float*>(workspace);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(window_minmax_partial_kernel, dim3((unsigned)blocks),
                     dim3(kStatThreads), 0, st, x, outer, rows, cols,
                     outer_pitch, row_pitch, partial);
  VTC_LAUNCH_CHECK();
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(64), 0, st, partial,
                     (int)blocks, out_min_max);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

extern "C" int vtc_rows_mean_abs_diff(const float* a, const float* b,
                                      int64_t rows, int64_t cols, float* out,
                                      void* stream) {
  VTC_REQUIRE(a && b && out, "vtc_rows_mean_abs_diff: null pointer");
  VTC_REQUIRE(rows > 0 && cols > 0 && rows <= 0x7fffffffLL,
              "vtc_rows_mean_abs_diff: bad sizes");
  hipLaunchKernelGGL(rows_mean_abs_diff_kernel, dim3((unsigned)rows),
                     dim3(kStatThreads), 0, as_stream(stream), a, b, cols,
                     out);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}
