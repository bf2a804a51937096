// Subspace (group-LASSO) ISTA/FISTA and the padded-group index kernels.
//
// Restates analysis_transforms/fully_connected/subspace_ista_fista.py:94-190.
// The reference rearranges codes into a zero-padded (b, G, m) tensor and the
// dictionary into a (G*m, n) matrix with duplicated rows for atoms that sit in
// several groups; the same layout is used here, so the two contractions are
// the fully-connected ones on G*m "slots".  Per iteration:
//     R = Y Dg - X            exact-f32 MFMA, or bf16x3 tiles (split over the
//                             slot axis when n gives too few output tiles)
//     P = Y - eta * (R Dg^T)
//     C = P * max(1 - lambda*eta / ||P_group||, 0);  Y = C + beta (C - C_prev)
// bf16x3: the second product carries the last two lines in its epilogue
// (epi_prox.h) for group sizes that are powers of two <= 32; otherwise, and on
// the exact-f32 path, P is written by the product and one of the proximal
// kernels below follows.
#include "common.h"
#include "fused_stream.h"
#include "gemm_f32.h"
#include "gemm_x3.h"
#include "epi_prox.h"
#include "fc_fused.h"

#include <vector>

namespace vtc {

struct EpiGradStep {  // Yo <- Y - eta * g  (out of place, see epi_prox.h)
  const float* Y;
  float* Yo;
  int64_t ld;
  float eta;
  __device__ __forceinline__ void operator()(int64_t row, int64_t col, float g,
                                             int) const {
    const int64_t i = row * ld + col;
    Yo[i] = sub_rn(Y[i], mul_rn(eta, g));
  }
  // exact-f32 kernels (gemm_f32.h): Y is read before the K loop
  static constexpr bool kElemFetch = true;
  struct Fetched {
    float y;
  };
  __device__ __forceinline__ Fetched fetch(int64_t row, int64_t col) const {
    return Fetched{Y[row * ld + col]};
  }
  __device__ __forceinline__ void apply(int64_t row, int64_t col, float g, int,
                                        const Fetched& f) const {
    Yo[row * ld + col] = sub_rn(f.y, mul_rn(eta, g));
  }
  __device__ __forceinline__ void block_end() const {}
};

// G = R Dg^T with the gradient step and the group proximal step in the
// epilogue, on the bf16x3 tiles or the exact-f32 ones
template <int M>
static int launch_grad_prox(bool x3, const float* R, const float* Dg, float* Y,
                            float* C, float* Yo, float* Co, int64_t b,
                            int64_t slots, int64_t n, float eta, float cutoff,
                            float beta, int fista, double* delta_sum,
                            hipStream_t st, X3Scale sc = X3Scale(),
                            unsigned* y_max_out = nullptr) {
  if (y_max_out) {                 // f16 split: the new iterate's maximum too
    EpiGroupProx<M, false, true> t{Y, C, slots, eta, cutoff, beta, fista,
                                   delta_sum, 0.0, 0};
    t.Yo = Yo;
    t.Co = Co;
    t.track.max_out = y_max_out;
    return launch_gemm_x3(R, n, Dg, n, b, slots, n, t, st, 1, sc);
  }
  EpiGroupProx<M> e{Y, C, slots, eta, cutoff, beta, fista, delta_sum, 0.0, 0};
  e.Yo = Yo;
  e.Co = Co;
  return x3 ? launch_gemm_x3(R, n, Dg, n, b, slots, n, e, st, 1, sc)
            : launch_gemm_f32<true, true>(R, n, Dg, n, b, slots, n, 1, e, st);
}

__global__ void gather_rows_kernel(const float* __restrict__ D,
                                   const int32_t* __restrict__ index,
                                   const uint8_t* __restrict__ valid,
                                   float* __restrict__ out, int64_t slots,
                                   int64_t n) {
  const int64_t t = blockIdx.x;
  if (t >= slots) return;
  const bool ok = valid[t] != 0;
  const float* src = D + (int64_t)index[t] * n;
  for (int64_t c = threadIdx.x; c < n; c += blockDim.x)
    out[t * n + c] = ok ? src[c] : 0.f;
}

__global__ void gather_cols_kernel(const float* __restrict__ codes,
                                   const int32_t* __restrict__ index,
                                   const uint8_t* __restrict__ valid,
                                   float* __restrict__ out, int64_t b,
                                   int64_t s, int64_t slots) {
  const int64_t total = b * slots;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += stride) {
    const int64_t r = i / slots, t = i % slots;
    out[i] = valid[t] ? codes[r * s + index[t]] : 0.f;
  }
}

__global__ void scatter_add_kernel(const float* __restrict__ grouped,
                                   const int32_t* __restrict__ atom_ptr,
                                   const int32_t* __restrict__ atom_slots,
                                   float* __restrict__ codes, int64_t b,
                                   int64_t s, int64_t slots) {
  const int64_t total = b * s;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += stride) {
    const int64_t r = i / s, a = i % s;
    float acc = 0.f;
    for (int t = atom_ptr[a]; t < atom_ptr[a + 1]; ++t)
      acc = add_rn(acc, grouped[r * slots + atom_slots[t]]);
    codes[i] = acc;
  }
}

// One thread per (patch, group): norm over the m slots, shrink, extrapolate.
__global__ __launch_bounds__(256) void group_prox_kernel(
    float* __restrict__ Y, float* __restrict__ C, int64_t b, int64_t groups,
    int m, float cutoff, float beta, int fista, float eta,
    double* __restrict__ delta_sum) {
  const int64_t total = b * groups;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double local = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += stride) {
    float* y = Y + i * m;
    float* c = C + i * m;
    float sumsq = 0.f;
    for (int j = 0; j < m; ++j) sumsq = fmaf(y[j], y[j], sumsq);
    float norm = sqrtf(sumsq);
    if (norm == 0.f) norm = 1.f;  // subspace_ista_fista.py:150
    const float scale = clamp_min0(sub_rn(1.f, cutoff / norm));
    for (int j = 0; j < m; ++j) {
      const float cn = mul_rn(y[j], scale);
      const float d = sub_rn(cn, c[j]);
      y[j] = fista ? add_rn(cn, mul_rn(beta, d)) : cn;
      c[j] = cn;
      if (delta_sum) local += (double)(fabsf(d) / eta);
    }
  }
  if (delta_sum) {
    const double w = wave_sum(local);
    if ((threadIdx.x & 63) == 0) atomicAdd(delta_sum, w);
  }
}

// Same proximal step for group sizes that are powers of two (the padded layout
// keeps groups contiguous and aligned): a thread owns 4 consecutive slots
// (one 16-byte load per array), the group norm is completed with lane
// shuffles.  Fully coalesced: the thread-per-group kernel above touches 4 B of
// every 32 B per load instruction at m = 8 and ran at 1.8 TB/s.
template <int M>
__global__ __launch_bounds__(256) void group_prox_pow2_kernel(
    float* __restrict__ Y, float* __restrict__ C, int64_t quads, float cutoff,
    float beta, int fista, float eta, double* __restrict__ delta_sum) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double local = 0.0;
  // every lane of a wave runs the same number of iterations (shuffles below)
  const int64_t rounds = (quads + stride - 1) / stride;
  for (int64_t it = 0; it < rounds; ++it) {
    const int64_t i = it * stride + (int64_t)blockIdx.x * blockDim.x +
                      threadIdx.x;
    const bool on = i < quads;
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f), c = y;
    if (on) {
      y = reinterpret_cast<const float4*>(Y)[i];
      c = reinterpret_cast<const float4*>(C)[i];
    }
    float yv[4] = {y.x, y.y, y.z, y.w};
    const float cv[4] = {c.x, c.y, c.z, c.w};
    float scale[4];
    if (M >= 4) {
      float sumsq = fmaf(yv[0], yv[0], fmaf(yv[1], yv[1],
                    fmaf(yv[2], yv[2], yv[3] * yv[3])));
#pragma unroll
      for (int off = 1; off < M / 4; off <<= 1)
        sumsq += __shfl_xor(sumsq, off, 64);
      float norm = sqrtf(sumsq);
      if (norm == 0.f) norm = 1.f;
      const float sc = clamp_min0(sub_rn(1.f, cutoff / norm));
      scale[0] = scale[1] = scale[2] = scale[3] = sc;
    } else if (M == 2) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float norm = sqrtf(fmaf(yv[2 * g], yv[2 * g],
                                yv[2 * g + 1] * yv[2 * g + 1]));
        if (norm == 0.f) norm = 1.f;
        scale[2 * g] = scale[2 * g + 1] =
            clamp_min0(sub_rn(1.f, cutoff / norm));
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float norm = fabsf(yv[k]);
        if (norm == 0.f) norm = 1.f;
        scale[k] = clamp_min0(sub_rn(1.f, cutoff / norm));
      }
    }
    float yn[4], cn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      cn[k] = mul_rn(yv[k], scale[k]);
      const float d = sub_rn(cn[k], cv[k]);
      yn[k] = fista ? add_rn(cn[k], mul_rn(beta, d)) : cn[k];
      if (delta_sum && on) local += (double)(fabsf(d) / eta);
    }
    if (on) {
      reinterpret_cast<float4*>(Y)[i] = make_float4(yn[0], yn[1], yn[2], yn[3]);
      reinterpret_cast<float4*>(C)[i] = make_float4(cn[0], cn[1], cn[2], cn[3]);
    }
  }
  if (delta_sum) {
    const double w = wave_sum(local);
    if ((threadIdx.x & 63) == 0) atomicAdd(delta_sum, w);
  }
}

static bool launch_group_prox_pow2(float* Y, float* C, int64_t b,
                                   int64_t groups, int m, float cutoff,
                                   float beta, int fista, float eta,
                                   double* delta_sum, hipStream_t st) {
  const int64_t total = b * groups * m;
  if (total % 4 != 0 || (reinterpret_cast<uintptr_t>(Y) & 15) ||
      (reinterpret_cast<uintptr_t>(C) & 15))
    return false;
  const int64_t quads = total / 4;
  int64_t blocks = ceil_div(quads, 256);
  if (blocks > 8192) blocks = 8192;
#define VTC_PROX_CASE(MM)                                                    \
  case MM:                                                                   \
    hipLaunchKernelGGL(group_prox_pow2_kernel<MM>, dim3((unsigned)blocks),   \
                       dim3(256), 0, st, Y, C, quads, cutoff, beta, fista,   \
                       eta, delta_sum);                                      \
    return true;
  switch (m) {
    VTC_PROX_CASE(1)
    VTC_PROX_CASE(2)
    VTC_PROX_CASE(4)
    VTC_PROX_CASE(8)
    VTC_PROX_CASE(16)
    VTC_PROX_CASE(32)
    VTC_PROX_CASE(64)
  }
#undef VTC_PROX_CASE
  return false;
}

// f16x3 scale state of the tiled path (x3_scale.h)
constexpr int kSubStateWords = 64 + 4 * kCxMaxSlotWords;

static size_t subspace_ws_bytes(int64_t b, int64_t n, int64_t slots) {
  return 3 * align_up((size_t)b * slots * sizeof(float), 256) +  // Y, Y', C'
         align_up((size_t)b * n * sizeof(float), 256) +       // R
         align_up((size_t)slots * n * sizeof(float), 256) +   // Dg^T (bf16x3)
         align_up((size_t)gemm_x3_want_slices(b, n, slots) * b * n *
                      sizeof(float), 256) +                       // split-K slabs
         align_up(kSubStateWords * sizeof(unsigned), 256) +       // f16 scales
         256;
}

// out[c][r] = in[r][c]
__global__ void transpose_kernel(const float* __restrict__ in,
                                 float* __restrict__ out, int64_t rows,
                                 int64_t cols) {
  __shared__ float tile[32][33];
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int64_t r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < rows && c < cols) ? in[r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int64_t c = c0 + i, r = r0 + threadIdx.x;
    if (c < cols && r < rows) out[c * rows + r] = tile[threadIdx.x][i];
  }
}

int launch_transpose(const float* in, float* out, int64_t rows, int64_t cols,
                     hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel,
                     dim3((unsigned)ceil_div(cols, 32),
                          (unsigned)ceil_div(rows, 32)),
                     dim3(32, 8), 0, st, in, out, rows, cols);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

static unsigned flat_grid(int64_t total) {
  int64_t blocks = ceil_div(total, 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

}  // namespace vtc

using namespace vtc;

extern "C" int vtc_group_gather_rows(const float* dictionary,
                                     const int32_t* index,
                                     const uint8_t* valid,
                                     float* grouped_dictionary, int64_t slots,
                                     int64_t n, void* stream) {
  VTC_REQUIRE(dictionary && index && valid && grouped_dictionary,
              "vtc_group_gather_rows: null pointer");
  VTC_REQUIRE(slots > 0 && n > 0, "vtc_group_gather_rows: bad sizes");
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)slots), dim3(256), 0,
                     as_stream(stream), dictionary, index, valid,
                     grouped_dictionary, slots, n);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

extern "C" int vtc_group_gather_cols(const float* codes, const int32_t* index,
                                     const uint8_t* valid,
                                     float* grouped_codes, int64_t b,
                                     int64_t s, int64_t slots, void* stream) {
  VTC_REQUIRE(b >= 0 && s > 0 && slots > 0, "vtc_group_gather_cols: sizes");
  if (b == 0) return VTC_OK;   // empty tensors have null data pointers
  VTC_REQUIRE(codes && index && valid && grouped_codes,
              "vtc_group_gather_cols: null pointer");
  hipLaunchKernelGGL(gather_cols_kernel, dim3(flat_grid(b * slots)), dim3(256),
                     0, as_stream(stream), codes, index, valid, grouped_codes,
                     b, s, slots);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

extern "C" int vtc_group_scatter_add(const float* grouped_codes,
                                     const int32_t* atom_ptr,
                                     const int32_t* atom_slots, float* codes,
                                     int64_t b, int64_t s, int64_t slots,
                                     void* stream) {
  VTC_REQUIRE(b >= 0 && s > 0 && slots > 0, "vtc_group_scatter_add: sizes");
  if (b == 0) return VTC_OK;   // empty tensors have null data pointers
  VTC_REQUIRE(grouped_codes && atom_ptr && atom_slots && codes,
              "vtc_group_scatter_add: null pointer");
  hipLaunchKernelGGL(scatter_add_kernel, dim3(flat_grid(b * s)), dim3(256), 0,
                     as_stream(stream), grouped_codes, atom_ptr, atom_slots,
                     codes, b, s, slots);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

extern "C" size_t vtc_subspace_ista_fista_workspace_bytes(int64_t b, int64_t n,
                                                          int64_t groups,
                                                          int64_t m) {
  if (b <= 0 || n <= 0 || groups <= 0 || m <= 0) return 256;
  const size_t tiled = subspace_ws_bytes(b, n, groups * m);
  const size_t streamed =
      stream_shape_supported(b, n, groups * m, m, VTC_F16X3)
          ? stream_workspace_bytes(b, n, groups * m, VTC_F16X3)
          : 0;
  return tiled > streamed ? tiled : streamed;
}

extern "C" int vtc_subspace_ista_fista(
    const float* images, const float* grouped_dictionary,
    const float* initial_grouped, float* grouped_codes, int64_t b, int64_t n,
    int64_t groups, int64_t m, float stepsize, float sparsity_weight,
    int num_iters, int variant, float early_stopping_epsilon, int precision,
    void* workspace, size_t workspace_bytes, int* iters_run, void* stream) {
  VTC_REQUIRE(b == 0 || (images && grouped_dictionary && grouped_codes),
              "vtc_subspace_ista_fista: null pointer");
  VTC_REQUIRE(precision == VTC_F32 || precision == VTC_BF16X3 ||
                  precision == VTC_F16X3,
              "vtc_subspace_ista_fista: precision must be VTC_F32, "
              "VTC_BF16X3 or VTC_F16X3");
  VTC_REQUIRE(b >= 0 && n > 0 && groups > 0 && m > 0,
              "vtc_subspace_ista_fista: bad sizes");
  VTC_REQUIRE(variant == VTC_ISTA || variant == VTC_FISTA,
              "vtc_subspace_ista_fista: variant must be ista or fista");
  VTC_REQUIRE(num_iters >= 1, "vtc_subspace_ista_fista: num_iters >= 1");
  if (iters_run) *iters_run = 0;
  if (b == 0) return VTC_OK;
  const int64_t slots = groups * m;
  // 16x16 patches, groups of 1/2/4/8 slots, no early stopping: the fused
  // persistent kernel with streamed state (fused_stream.hip)
  if (precision != VTC_F32 && early_stopping_epsilon < 0.f &&
      num_iters <= fused_max_iters_for_stream() &&
      stream_shape_supported(b, n, slots, m, precision))
    return run_stream(images, grouped_dictionary, initial_grouped,
                      grouped_codes, b, n, slots, m, stepsize, nullptr,
                      sparsity_weight, num_iters, variant, VTC_SOFT, precision,
                      workspace, workspace_bytes, iters_run,
                      as_stream(stream));
  // tiled path: the f16 split (scaled units, x3_scale.h) runs where the
  // proximal step is fused into the gradient product's epilogue; the other
  // group shapes take the bf16 split
  bool f16 = (precision == VTC_F16X3);
  if (f16) precision = VTC_BF16X3;
  if (!workspace || workspace_bytes < subspace_ws_bytes(b, n, slots)) {
    set_error("vtc_subspace_ista_fista: workspace too small");
    return VTC_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  Carver ws(workspace);
  float* Y = ws.take<float>((size_t)b * slots);
  float* Yout = ws.take<float>((size_t)b * slots);  // out-of-place targets of
  float* Cout = ws.take<float>((size_t)b * slots);  // the proximal epilogue
  float* Cin = grouped_codes;
  float* R = ws.take<float>((size_t)b * n);
  float* DgT = ws.take<float>((size_t)slots * n);
  const int k1_slices = (precision == VTC_BF16X3)
                            ? gemm_x3_want_slices(b, n, slots)
                            : 1;
  float* slabs = ws.take<float>((size_t)gemm_x3_want_slices(b, n, slots) * b * n);
  double* delta_sum = ws.take<double>(1);
  unsigned* state = ws.take<unsigned>(kSubStateWords);
  const float eta = stepsize;
  const float cutoff = sparsity_weight * stepsize;
  const float eps = early_stopping_epsilon;
  // bf16x3 contractions need both operands k-contiguous: Y (b,slots) with
  // Dg^T (n,slots) for the residual, R (b,n) with Dg (slots,n) for the gradient
  const bool x3 = (precision == VTC_BF16X3);
  if (x3) {
    if (!gemm_x3_usable(Y, slots, DgT, slots) ||
        !gemm_x3_usable(R, n, grouped_dictionary, n)) {
      set_error("vtc_subspace_ista_fista: bf16x3 needs n and G*m to be "
                "multiples of 4 and 16-byte aligned operands");
      return VTC_ERR_UNSUPPORTED;
    }
    int rc = launch_transpose(grouped_dictionary, DgT, slots, n, st);
    if (rc != VTC_OK) return rc;
  }
  {
    const bool wide = slots % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(Y) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(grouped_codes) & 15) == 0;
    const bool pow2 = m == 1 || m == 2 || m == 4 || m == 8 || m == 16 ||
                      m == 32;
    f16 = f16 && x3 && wide && pow2;
  }
  float* dscale = f16 ? reinterpret_cast<float*>(state) : nullptr;
  unsigned* y_slot[2] = {state + 64, state + 64 + kCxMaxSlotWords};
  unsigned* r_slot[2] = {state + 64 + 2 * kCxMaxSlotWords,
                         state + 64 + 3 * kCxMaxSlotWords};
  if (f16) {
    VTC_HIP_CHECK(hipMemsetAsync(state, 0, kSubStateWords * sizeof(unsigned),
                                 st));
    hipLaunchKernelGGL(cx_array_scale_kernel, dim3(1), dim3(1024), 0, st,
                       grouped_dictionary, slots * n, dscale);
    VTC_LAUNCH_CHECK();
    if (initial_grouped) {
      hipLaunchKernelGGL(cx_array_max_kernel, dim3(1024), dim3(256), 0, st,
                         initial_grouped, b * slots, y_slot[0]);
      VTC_LAUNCH_CHECK();
    }
  }
  const bool fista = (variant == VTC_FISTA);
  const size_t bytes = (size_t)b * slots * sizeof(float);
  if (initial_grouped) {
    VTC_HIP_CHECK(hipMemcpyAsync(Y, initial_grouped, bytes,
                                 hipMemcpyDeviceToDevice, st));
    VTC_HIP_CHECK(hipMemcpyAsync(grouped_codes, initial_grouped, bytes,
                                 hipMemcpyDeviceToDevice, st));
  } else {
    VTC_HIP_CHECK(hipMemsetAsync(Y, 0, bytes, st));
    VTC_HIP_CHECK(hipMemsetAsync(grouped_codes, 0, bytes, st));
  }
  std::vector<float> betas;
  fista_betas(num_iters, &betas);
  int done = 0;
  for (int k = 0; k < num_iters; ++k) {
    EpiMinus e1{R, images, n, n};
    // f16 split: see run_generic (fc_inference.hip)
    X3Scale sc1, sc2;
    if (f16) {
      sc1.b_scale = sc2.b_scale = dscale;
      sc1.a_max = y_slot[k & 1];
      sc1.clear = y_slot[(k + 1) & 1];
      sc2.a_max = r_slot[k & 1];
      sc2.clear = r_slot[(k + 1) & 1];
    }
    int rc;
    if (x3 && k1_slices > 1) {
      // few output tiles (n is small): split the long slot axis over blocks
      EpiSlab es{slabs, b * n, n};
      rc = launch_gemm_x3(Y, slots, DgT, slots, b, n, slots, es, st,
                          k1_slices, sc1);
      if (rc == VTC_OK)
        rc = launch_slab_reduce_minus(slabs, k1_slices, b * n, images, R, st,
                                      f16 ? r_slot[k & 1] : nullptr);
    } else if (x3 && f16) {
      EpiMinusMax e1m{R, images, n, n, r_slot[k & 1]};
      rc = launch_gemm_x3(Y, slots, DgT, slots, b, n, slots, e1m, st, 1, sc1);
    } else if (x3) {
      rc = launch_gemm_x3(Y, slots, DgT, slots, b, n, slots, e1, st, 1, sc1);
    } else {
      rc = launch_gemm_f32<true, false>(Y, slots, grouped_dictionary, n, b, n,
                                        slots, 1, e1, st);
    }
    if (rc != VTC_OK) return rc;
    if (eps >= 0.f)
      VTC_HIP_CHECK(hipMemsetAsync(delta_sum, 0, sizeof(double), st));
    const float beta_k = fista ? betas[k] : 0.f;
    double* dsum = eps >= 0.f ? delta_sum : nullptr;
    bool fused_prox = false;
    // 16-byte epilogue accesses: slots a multiple of 4, 16-byte aligned state
    const bool wide_ok = slots % 4 == 0 &&
                         (reinterpret_cast<uintptr_t>(Y) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(grouped_codes) & 15) == 0;
    // (exact f32, few tiles: the 32x32-tile kernel with the element-wise
    // gradient step and the separate proximal kernel beats a handful of
    // blocks of the whole-tile epilogue)
    if (wide_ok && (x3 || !gemm_prefers_small(b, slots))) {
#define VTC_FUSED_PROX(MM)                                                  \
  case MM:                                                                  \
    rc = launch_grad_prox<MM>(x3, R, grouped_dictionary, Y, Cin, Yout,     \
                                 Cout, b, slots, n, eta, cutoff, beta_k,    \
                                 fista ? 1 : 0, dsum, st, sc2,              \
                                 f16 ? y_slot[(k + 1) & 1] : nullptr);      \
    fused_prox = true;                                                      \
    break;
      switch (m) {
        VTC_FUSED_PROX(1)
        VTC_FUSED_PROX(2)
        VTC_FUSED_PROX(4)
        VTC_FUSED_PROX(8)
        VTC_FUSED_PROX(16)
        VTC_FUSED_PROX(32)
        default: break;
      }
#undef VTC_FUSED_PROX
      if (fused_prox) {
        if (rc != VTC_OK) return rc;
        float* t = Cin; Cin = Cout; Cout = t;
        t = Y; Y = Yout; Yout = t;
      }
    }
    if (!fused_prox) {
    // gradient step out of place, then the proximal kernels in place on the
    // new buffer (their blocks own whole, 128-byte aligned runs of rows)
    EpiGradStep e2{Y, Yout, slots, eta};
    rc = x3 ? launch_gemm_x3(R, n, grouped_dictionary, n, b, slots, n, e2, st)
            : launch_gemm_f32<true, true>(R, n, grouped_dictionary, n, b,
                                          slots, n, 1, e2, st);
    if (rc != VTC_OK) return rc;
    { float* t = Y; Y = Yout; Yout = t; }
    if (!launch_group_prox_pow2(Y, Cin, b, groups, (int)m, cutoff,
                                fista ? betas[k] : 0.f, fista ? 1 : 0, eta,
                                eps >= 0.f ? delta_sum : nullptr, st))
      hipLaunchKernelGGL(group_prox_kernel, dim3(flat_grid(b * groups)),
                         dim3(256), 0, st, Y, Cin, b, groups, (int)m,
                         cutoff, fista ? betas[k] : 0.f, fista ? 1 : 0, eta,
                         eps >= 0.f ? delta_sum : nullptr);
    VTC_LAUNCH_CHECK();
    }
    done = k + 1;
    if (eps >= 0.f) {
      double total = 0.0;
      VTC_HIP_CHECK(hipMemcpyAsync(&total, delta_sum, sizeof(double),
                                   hipMemcpyDeviceToHost, st));
      VTC_HIP_CHECK(hipStreamSynchronize(st));
      // the reference averages over the padded (b, G, m) tensor
      const float mean = (float)(total / ((double)b * (double)slots));
      if (mean < eps && k > 0) break;
    }
  }
  if (Cin != grouped_codes)
    VTC_HIP_CHECK(hipMemcpyAsync(grouped_codes, Cin, bytes,
                                 hipMemcpyDeviceToDevice, st));
  if (iters_run) *iters_run = done;
  return VTC_OK;
}
