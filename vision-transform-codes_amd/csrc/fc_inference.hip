// Fully-connected ISTA/FISTA, general exact-f32 path (any b, n, s; all four
// thresholds; warm start; early stopping).
//
// Follows analysis_transforms/fully_connected/ista_fista.py:100-146 of the
// reference: per iteration
//     R = Y D - X                      (kernel 1, epilogue subtracts X)
//     C = shrink(Y - eta * (R D^T))    (kernel 2, the whole proximal step and
//     Y = C + beta (C - C_prev)         the FISTA extrapolation live in its
//     C_prev = C                        epilogue: no elementwise passes)
// The reference's formulation is kept (residual form, eta*g rounded before the
// subtraction, beta rounded to f32 before the multiply) so that differences
// stay at f32 summation-order level.
#include "common.h"
#include "gemm_f32.h"
#include "gemm_x3.h"
#include "epi_prox.h"
#include "fc_fused.h"
#include "fused_stream.h"
#include "fc_small.h"

#include <math.h>
#include <vector>

namespace vtc {

struct EpiProx {
  float* Y;        // gradient evaluation points, updated in place
  float* C;        // codes of the previous iteration in, new codes out
  int64_t ld;
  float eta, cutoff, beta;
  int mode;        // vtc_threshold
  int fista;       // 0: Y and C are the same buffer
  double* delta_sum;  // sum |C - C_prev| / eta, or nullptr
  double local;
  const float* eta_dev = nullptr;   // sync-free callers, see epi_prox.h
  float lam = 0.f;
  float* Yo = nullptr;              // out-of-place targets, see epi_prox.h
  float* Co = nullptr;
  __device__ __forceinline__ void resolve() {
    if (eta_dev) {
      eta = *eta_dev;
      cutoff = mul_rn(lam, eta);
    }
  }

  // two-phase protocol of gemm_f32.h (the targets Yo / Co are other buffers):
  // Y and the previous codes are read before the K loop
  static constexpr bool kElemFetch = true;
  struct Fetched {
    float y, c;
  };
  __device__ __forceinline__ Fetched fetch(int64_t row, int64_t col) const {
    const int64_t i = row * ld + col;
    Fetched f;
    f.y = Y[i];
    f.c = fista ? C[i] : 0.f;
    return f;
  }
  __device__ __forceinline__ void apply(int64_t row, int64_t col, float g, int,
                                        const Fetched& f) {
    const int64_t i = row * ld + col;
    const float c = shrink(sub_rn(f.y, mul_rn(eta, g)), cutoff, mode);
    float d;
    if (fista) {
      d = sub_rn(c, f.c);
      Yo[i] = add_rn(c, mul_rn(beta, d));
    } else {
      d = sub_rn(c, f.y);
    }
    Co[i] = c;
    if (delta_sum) local += (double)(fabsf(d) / eta);
  }
  __device__ __forceinline__ void block_end() {
    if (!delta_sum) return;
    const double w = wave_sum(local);
    if ((threadIdx.x & 63) == 0) atomicAdd(delta_sum, w);
  }
};

void fista_betas(int num_iters, std::vector<float>* out) {
  // ista_fista.py:123-125, Python float64 arithmetic; the product
  // beta * (codes - old) rounds beta to f32 first.
  out->resize(num_iters);
  double t = 1.0;
  for (int k = 0; k < num_iters; ++k) {
    const double t_next = (1.0 + sqrt(1.0 + 4.0 * t * t)) / 2.0;
    (*out)[k] = (float)((t - 1.0) / t_next);
    t = t_next;
  }
}

int launch_transpose(const float* in, float* out, int64_t rows, int64_t cols,
                     hipStream_t st);  // subspace.hip

// f16x3 scale state of the tiled path (x3_scale.h): {sigma_D, 1 / sigma_D} and
// two slots each for max |Y| and max |R|
constexpr int kX3StateWords = 64 + 4 * kCxMaxSlotWords;

static size_t generic_workspace_bytes(int64_t b, int64_t n, int64_t s) {
  size_t bytes = 0;
  bytes += 3 * align_up((size_t)b * s * sizeof(float), 256);  // Y, Y', C'
  bytes += align_up((size_t)b * n * sizeof(float), 256);  // R
  bytes += align_up((size_t)s * n * sizeof(float), 256);  // D^T (bf16x3)
  bytes += align_up((size_t)gemm_x3_want_slices(b, n, s) * b * n *
                        sizeof(float), 256);  // split-K slabs
  bytes += 256;                                           // stop accumulator
  bytes += align_up(kX3StateWords * sizeof(unsigned), 256);  // f16x3 scales
  return bytes;
}

// x3 = false: exact-f32 MFMA; x3 = true: hi/lo split tiles (gemm_x3.h), the
// f16 split in power-of-two scaled units when f16 is set, else bf16
static int run_generic(const float* images, const float* dictionary,
                       const float* initial_codes, float* codes, int64_t b,
                       int64_t n, int64_t s, float eta, const float* eta_dev,
                       float lam, int num_iters, int variant, int threshold,
                       float eps, bool x3, bool f16, void* workspace,
                       size_t workspace_bytes, int* iters_run, hipStream_t st) {
  // lambda * eta: the Python float rounded to f32, then one f32 multiply (the
  // epilogues redo it on the device when eta lives there)
  const float cutoff = lam * eta;
  if (workspace_bytes < generic_workspace_bytes(b, n, s) || !workspace) {
    set_error("vtc_fc_ista_fista: workspace too small (%zu < %zu)",
              workspace_bytes, generic_workspace_bytes(b, n, s));
    return VTC_ERR_WORKSPACE;
  }
  Carver ws(workspace);
  float* Ybuf = ws.take<float>((size_t)b * s);
  float* Yalt = ws.take<float>((size_t)b * s);   // out-of-place targets of the
  float* Calt = ws.take<float>((size_t)b * s);   // proximal epilogue
  float* R = ws.take<float>((size_t)b * n);
  float* Dt = ws.take<float>((size_t)s * n);
  const int k1_slices = x3 ? gemm_x3_want_slices(b, n, s) : 1;
  float* slabs = ws.take<float>((size_t)gemm_x3_want_slices(b, n, s) * b * n);
  double* delta_sum = ws.take<double>(1);
  unsigned* state = ws.take<unsigned>(kX3StateWords);
  f16 = f16 && x3;
  float* dscale = f16 ? reinterpret_cast<float*>(state) : nullptr;
  unsigned* y_slot[2] = {state + 64, state + 64 + kCxMaxSlotWords};
  unsigned* r_slot[2] = {state + 64 + 2 * kCxMaxSlotWords,
                         state + 64 + 3 * kCxMaxSlotWords};
  if (f16) {
    VTC_HIP_CHECK(hipMemsetAsync(state, 0, kX3StateWords * sizeof(unsigned),
                                 st));
    hipLaunchKernelGGL(cx_array_scale_kernel, dim3(1), dim3(1024), 0, st,
                       dictionary, s * n, dscale);
    VTC_LAUNCH_CHECK();
    if (initial_codes) {
      hipLaunchKernelGGL(cx_array_max_kernel, dim3(1024), dim3(256), 0, st,
                         initial_codes, b * s, y_slot[0]);
      VTC_LAUNCH_CHECK();
    }
  }

  const bool fista = (variant == VTC_FISTA);
  float* Y = fista ? Ybuf : codes;  // ISTA evaluates the gradient at the codes
  if (x3) {
    if (!gemm_x3_usable(Y, s, Dt, s) || !gemm_x3_usable(R, n, dictionary, n)) {
      set_error("vtc_fc_ista_fista: bf16x3 outside the fused kernel needs n "
                "and s to be multiples of 4 and 16-byte aligned operands");
      return VTC_ERR_UNSUPPORTED;
    }
    int rc = launch_transpose(dictionary, Dt, s, n, st);
    if (rc != VTC_OK) return rc;
  }
  const size_t code_bytes = (size_t)b * s * sizeof(float);
  if (initial_codes) {
    VTC_HIP_CHECK(hipMemcpyAsync(codes, initial_codes, code_bytes,
                                 hipMemcpyDeviceToDevice, st));
    if (fista)
      VTC_HIP_CHECK(hipMemcpyAsync(Y, initial_codes, code_bytes,
                                   hipMemcpyDeviceToDevice, st));
  } else {
    VTC_HIP_CHECK(hipMemsetAsync(codes, 0, code_bytes, st));
    if (fista) VTC_HIP_CHECK(hipMemsetAsync(Y, 0, code_bytes, st));
  }

  std::vector<float> betas;
  fista_betas(num_iters, &betas);
  int done = 0;
  float* Cin = codes;      // (Y, Cin) are read, (Yout, Cout) written, then the
  float* Cout = Calt;      // roles swap
  float* Yout = Yalt;
  for (int k = 0; k < num_iters; ++k) {
    if (!fista) Y = Cin;   // ISTA evaluates the gradient at the codes
    // R = Y D - X : A = Y (b,s) k-contiguous, B = D (s,n) = [K][N]
    EpiMinus e1{R, images, n, n};
    // f16 split: the residual product reads max |Y_k|, leaves max |R_k| and
    // clears the slot of max |Y_(k+1)|; the gradient product reads max |R_k|,
    // leaves max |Y_(k+1)| and clears the slot of max |R_(k+1)|
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
      EpiSlab es{slabs, b * n, n};
      rc = launch_gemm_x3(Y, s, Dt, s, b, n, s, es, st, k1_slices, sc1);
      if (rc == VTC_OK)
        rc = launch_slab_reduce_minus(slabs, k1_slices, b * n, images, R, st,
                                      f16 ? r_slot[k & 1] : nullptr);
    } else if (x3 && f16) {
      EpiMinusMax e1m{R, images, n, n, r_slot[k & 1]};
      rc = launch_gemm_x3(Y, s, Dt, s, b, n, s, e1m, st, 1, sc1);
    } else if (x3) {
      rc = launch_gemm_x3(Y, s, Dt, s, b, n, s, e1, st, 1, sc1);
    } else {
      rc = launch_gemm_f32<true, false>(Y, s, dictionary, n, b, n, s, 1, e1,
                                        st);
    }
    if (rc != VTC_OK) return rc;
    if (eps >= 0.f)
      VTC_HIP_CHECK(hipMemsetAsync(delta_sum, 0, sizeof(double), st));
    // G = R D^T : A = R (b,n) k-contiguous, B = D (s,n) = [N][K]
    const bool wide_ok = s % 4 == 0 &&
                         (reinterpret_cast<uintptr_t>(Y) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(codes) & 15) == 0;
    // (few-tile products take the 32x32-tile kernel with the element-wise
    // epilogue: 40 blocks of the whole-tile epilogue leave most CUs idle)
    if (x3 || (wide_ok && !gemm_prefers_small(b, s))) {
      // 16-byte, pipelined epilogue (epi_prox.h); in ISTA Y and the codes are
      // one buffer and both stores carry the same value
      EpiGroupProx<1, true> e2{Y, Cin, s, eta, cutoff,
                               fista ? betas[k] : 0.f, fista ? 1 : 0,
                               eps >= 0.f ? delta_sum : nullptr, 0.0,
                               threshold, eta_dev, lam};
      e2.Yo = fista ? Yout : Cout;
      e2.Co = Cout;
      if (f16) {
        EpiGroupProx<1, true, true> e2t{Y, Cin, s, eta, cutoff,
                                        fista ? betas[k] : 0.f, fista ? 1 : 0,
                                        eps >= 0.f ? delta_sum : nullptr, 0.0,
                                        threshold, eta_dev, lam};
        e2t.Yo = e2.Yo;
        e2t.Co = e2.Co;
        e2t.track.max_out = y_slot[(k + 1) & 1];
        rc = launch_gemm_x3(R, n, dictionary, n, b, s, n, e2t, st, 1, sc2);
      } else {
        rc = x3 ? launch_gemm_x3(R, n, dictionary, n, b, s, n, e2, st, 1, sc2)
                : launch_gemm_f32<true, true>(R, n, dictionary, n, b, s, n, 1,
                                              e2, st);
      }
    } else {
      EpiProx e2{Y, Cin, s, eta, cutoff, fista ? betas[k] : 0.f, threshold,
                 fista ? 1 : 0, eps >= 0.f ? delta_sum : nullptr, 0.0,
                 eta_dev, lam};
      e2.Yo = Yout;
      e2.Co = Cout;
      rc = launch_gemm_f32<true, true>(R, n, dictionary, n, b, s, n, 1, e2,
                                       st);
    }
    if (rc != VTC_OK) return rc;
    {
      float* t = Cin; Cin = Cout; Cout = t;
      if (fista) { t = Y; Y = Yout; Yout = t; }
    }
    done = k + 1;
    if (eps >= 0.f) {
      double total = 0.0;
      VTC_HIP_CHECK(hipMemcpyAsync(&total, delta_sum, sizeof(double),
                                   hipMemcpyDeviceToHost, st));
      VTC_HIP_CHECK(hipStreamSynchronize(st));
      const float mean = (float)(total / ((double)b * (double)s));
      if (mean < eps && k > 0) break;  // ista_fista.py:143-144
    }
  }
  if (Cin != codes)
    VTC_HIP_CHECK(hipMemcpyAsync(codes, Cin, code_bytes,
                                 hipMemcpyDeviceToDevice, st));
  if (iters_run) *iters_run = done;
  return VTC_OK;
}

}  // namespace vtc

using namespace vtc;

extern "C" size_t vtc_fc_ista_fista_workspace_bytes(int64_t b, int64_t n,
                                                    int64_t s,
                                                    int precision) {
  if (b <= 0 || n <= 0 || s <= 0) return 256;
  const size_t generic = generic_workspace_bytes(b, n, s);
  if (precision == VTC_F32) return generic;
  const size_t fused = fused_workspace_bytes(b, n, s, precision);
  // bf16x3 falls back to the tiled bf16x3 contraction for shapes (or options)
  // the fused kernel does not cover: size for the larger of the two
  if (precision == VTC_BF16X3 || precision == VTC_F16X3) {
    const size_t streamed = stream_shape_supported(b, n, s, 1, precision)
                                ? stream_workspace_bytes(b, n, s, precision)
                                : 0;
    const size_t most = fused > generic ? fused : generic;
    return most > streamed ? most : streamed;
  }
  return fused;
}

static int fc_ista_fista_impl(const float* images, const float* dictionary,
                              const float* initial_codes, float* codes,
                              int64_t b, int64_t n, int64_t s, float stepsize,
                              const float* stepsize_dev,
                              float sparsity_weight, int num_iters,
                              int variant, int threshold,
                              float early_stopping_epsilon, int precision,
                              void* workspace, size_t workspace_bytes,
                              int* iters_run, void* stream) {
  VTC_REQUIRE(b >= 0 && n > 0 && s > 0, "vtc_fc_ista_fista: bad sizes");
  VTC_REQUIRE(b == 0 || (images && dictionary && codes),
              "vtc_fc_ista_fista: null pointer");
  VTC_REQUIRE(variant == VTC_ISTA || variant == VTC_FISTA,
              "vtc_fc_ista_fista: variant must be ista or fista");
  VTC_REQUIRE(threshold >= VTC_SOFT && threshold <= VTC_HARD_NONNEG,
              "vtc_fc_ista_fista: unknown threshold mode %d", threshold);
  VTC_REQUIRE(num_iters >= 1,
              "vtc_fc_ista_fista: num_iters must be >= 1 (the reference "
              "leaves `codes` unbound for 0)");
  VTC_REQUIRE(precision >= VTC_F32 && precision <= VTC_F16X3,
              "vtc_fc_ista_fista: unknown precision %d", precision);
  if (iters_run) *iters_run = 0;
  if (b == 0) return VTC_OK;
  hipStream_t st = as_stream(stream);
  const bool fused_ok = early_stopping_epsilon < 0.f &&
                        num_iters <= fused_max_iters() &&
                        fused_shape_supported(b, n, s, precision);
  if (precision == VTC_BF16 && !fused_ok) {
    set_error("vtc_fc_ista_fista: VTC_BF16 exists only as the fused kernel "
              "(n == 256, s in {256, 512, 1024}, no early stopping); use "
              "VTC_F16X3, VTC_BF16X3 or VTC_F32");
    return VTC_ERR_UNSUPPORTED;
  }
  // (the register-resident kernels move patches and codes as float4)
  const bool aligned16 =
      ((reinterpret_cast<uintptr_t>(images) |
        reinterpret_cast<uintptr_t>(dictionary) |
        reinterpret_cast<uintptr_t>(codes) |
        reinterpret_cast<uintptr_t>(initial_codes)) & 15) == 0;
  // 8x8 patches against 64 / 128 / 192 atoms, exact f32: everything on the CU
  if (precision == VTC_F32 && early_stopping_epsilon < 0.f && aligned16 &&
      num_iters <= fused_max_iters() && small_shape_supported(n, s))
    return run_small(images, dictionary, initial_codes, codes, b, n, s,
                     stepsize, stepsize_dev, sparsity_weight, num_iters,
                     variant, threshold, iters_run, st);
  // 12x12 patches against 288 / 576 atoms, exact f32: state on the CU
  if (precision == VTC_F32 && early_stopping_epsilon < 0.f && aligned16 &&
      num_iters <= fused_max_iters() && chip16_shape_supported(n, s))
    return run_chip16(images, dictionary, initial_codes, codes, b, n, s,
                      stepsize, stepsize_dev, sparsity_weight, num_iters,
                      variant, threshold, workspace, workspace_bytes,
                      iters_run, st);
  if (precision != VTC_F32 && fused_ok)
    return run_fused(images, dictionary, initial_codes, codes, b, n, s,
                     stepsize, stepsize_dev, sparsity_weight, num_iters,
                     variant, threshold, precision, workspace, workspace_bytes,
                     iters_run, st);
  // 16x16 patches against more atoms than the on-chip state holds: the fused
  // kernel with streamed state
  if (precision != VTC_F32 && precision != VTC_BF16 &&
      early_stopping_epsilon < 0.f && num_iters <= fused_max_iters() &&
      stream_shape_supported(b, n, s, 1, precision))
    return run_stream(images, dictionary, initial_codes, codes, b, n, s, 1,
                      stepsize, stepsize_dev, sparsity_weight, num_iters,
                      variant, threshold, precision, workspace,
                      workspace_bytes, iters_run, st);
  // elsewhere the split-operand modes run on the tiled hi/lo contraction
  // (gemm_x3.h): f16x3 in scaled units, bf16x3 as is
  return run_generic(images, dictionary, initial_codes, codes, b, n, s,
                     stepsize, stepsize_dev, sparsity_weight, num_iters,
                     variant, threshold, early_stopping_epsilon,
                     precision != VTC_F32, precision == VTC_F16X3, workspace,
                     workspace_bytes, iters_run, st);
}

extern "C" int vtc_fc_ista_fista(const float* images, const float* dictionary,
                                 const float* initial_codes, float* codes,
                                 int64_t b, int64_t n, int64_t s,
                                 float stepsize, float sparsity_weight,
                                 int num_iters, int variant, int threshold,
                                 float early_stopping_epsilon, int precision,
                                 void* workspace, size_t workspace_bytes,
                                 int* iters_run, void* stream) {
  return fc_ista_fista_impl(images, dictionary, initial_codes, codes, b, n, s,
                            stepsize, nullptr, sparsity_weight, num_iters,
                            variant, threshold, early_stopping_epsilon,
                            precision, workspace, workspace_bytes, iters_run,
                            stream);
}

extern "C" int vtc_fc_ista_fista_dev(
    const float* images, const float* dictionary, const float* initial_codes,
    float* codes, int64_t b, int64_t n, int64_t s, const float* stepsize_dev,
    float sparsity_weight, int num_iters, int variant, int threshold,
    float early_stopping_epsilon, int precision, void* workspace,
    size_t workspace_bytes, int* iters_run, void* stream) {
  VTC_REQUIRE(stepsize_dev != nullptr,
              "vtc_fc_ista_fista_dev: stepsize_dev is null");
  return fc_ista_fista_impl(images, dictionary, initial_codes, codes, b, n, s,
                            0.f, stepsize_dev, sparsity_weight, num_iters,
                            variant, threshold, early_stopping_epsilon,
                            precision, workspace, workspace_bytes, iters_run,
                            stream);
}
