// Largest eigenvalue of a small symmetric matrix on the device.
//
// The reference takes the ISTA/FISTA step size from
//   torch.symeig(D^T D)[0][-1]        (ista_fista.py:72-80, and the subspace /
//                                      convolutional variants)
// once per plugin call, i.e. once per training step.  A full rocSOLVER
// eigen-decomposition of a 256 x 256 matrix costs ~4 ms on MI355X (measured:
// 5 % of a whole training step).  Only the top eigenvalue is needed, so this is a
// single-workgroup Lanczos iteration with full re-orthogonalisation (the
// Krylov basis lives in LDS), followed by 64-way sectioning on the tridiagonal
// matrix with Sturm counts, in double precision, by one wave.  For n <= 128
// the Krylov space is the whole space and the result is the exact top
// eigenvalue up to rounding; for n <= 256, 128 steps converge to f32
// resolution on the spectra of interest (extreme eigenvalue, Kaniel-Paige).
#include "common.h"

namespace vtc {

constexpr int kLanczosMaxN = 256;
constexpr int kLanczosMaxK = 128;

constexpr int kLanczosThreads = 1024;   // 16 waves: 4 per SIMD
constexpr int kLanczosParts = 4;        // row ranges of the mat-vec

// sum over the first 256 threads (the vector components); every thread of the
// block calls this and receives the result
__device__ __forceinline__ double block_sum_vec(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if (threadIdx.x < 256 && (threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// Largest eigenvalue of the symmetric tridiagonal matrix (alpha, beta) of
// order m from Sturm counts, double precision, by ONE WAVE (all 64 lanes call
// it with the same arguments and get the same result): every round each lane
// takes the Sturm count at one of 64 interior points of the bracket, which
// shrinks 65x per round (a one-lane bisection, 50 rounds of an m-long chain of
// divisions at every convergence check, was 1 ms of the 1.1 ms this kernel took).
__device__ double tridiagonal_lambda_max(const double* alpha,
                                         const double* beta, int m) {
  const int lane = threadIdx.x & 63;
  double lo = alpha[0], hi = alpha[0];
  for (int i = 0; i < m; ++i) {
    const double off = (i > 0 ? fabs(beta[i - 1]) : 0.0) +
                       (i + 1 < m ? fabs(beta[i]) : 0.0);
    lo = fmin(lo, alpha[i] - off);
    hi = fmax(hi, alpha[i] + off);
  }
  if (!(hi == hi) || !(lo == lo)) return hi + lo;  // NaN propagates
  // invariant: count(lo) < m <= count(hi), count(x) = #eigenvalues < x
  for (int it = 0; it < 40 && hi - lo > 1e-14 * fmax(fabs(hi), fabs(lo));
       ++it) {
    const double step = (hi - lo) / 65.0;
    const double x = lo + step * (double)(lane + 1);
    int below = 0;
    double d = 1.0;
    for (int i = 0; i < m; ++i) {
      const double off2 = (i > 0) ? beta[i - 1] * beta[i - 1] : 0.0;
      d = (alpha[i] - x) - (i > 0 ? off2 / d : 0.0);
      if (d == 0.0) d = -1e-300;
      if (d < 0.0) ++below;
    }
    // the lanes whose point is above the top eigenvalue form a suffix
    const unsigned long long above = __ballot(below == m);
    if (above == 0ull) {
      lo = lo + step * 64.0;
    } else {
      const int first = __ffsll((long long)above) - 1;
      const double new_hi = lo + step * (double)(first + 1);
      if (first > 0) lo = lo + step * (double)first;
      hi = new_hi;
    }
  }
  return 0.5 * (lo + hi);
}

// Thread layout: tid = part * 256 + t.  Component t of every vector belongs
// to the threads (.., t); part 0 owns the Lanczos recurrences, parts 1..3 only
// help with the mat-vec and the Gram-Schmidt sums.
__global__ __launch_bounds__(kLanczosThreads) void lanczos_lambda_max_kernel(
    const float* __restrict__ G, int n, int k, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* V = lds;                         // [k][n] Krylov basis
  float* wl = V + (size_t)k * n;          // [n] work vector
  float* coef = wl + n;                   // [k]
  float* partial = coef + k;              // [parts][256]
  __shared__ double red[4];
  __shared__ double alpha[kLanczosMaxK];
  __shared__ double beta[kLanczosMaxK];
  __shared__ int steps_done;
  __shared__ int stop_flag;
  __shared__ double last_ritz;
  const int tid = threadIdx.x;
  const int t = tid & 255, part = tid >> 8;
  const int lane = tid & 63, wave = tid >> 6;   // 16 waves
  const bool on = t < n;
  const bool owner = (part == 0) && on;

  // deterministic, generic start vector
  float v = 0.f;
  if (owner) {
    unsigned x = (unsigned)t * 2654435761u + 12345u;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    v = 0.5f + (float)(x & 0xffff) / 65536.f;
  }
  double nrm = sqrt(block_sum_vec((double)v * v, red));
  v = (float)(v / nrm);
  float v_prev = 0.f;
  double beta_prev = 0.0;
  if (tid == 0) {
    steps_done = k;
    stop_flag = 0;
    last_ritz = -1.0;
  }
  double tscale = 0.0;
  const int rows_per_part = (n + kLanczosParts - 1) / kLanczosParts;
  const int i0 = part * rows_per_part;
  const int i1 = (i0 + rows_per_part < n) ? i0 + rows_per_part : n;
  __syncthreads();

  for (int j = 0; j < k; ++j) {
    if (owner) V[(size_t)j * n + t] = v;
    __syncthreads();
    // w = G v  (G symmetric: "column" t is contiguous across threads); each
    // part sums its row range with 4 independent accumulators
    {
      const float* vj = V + (size_t)j * n;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      if (on) {
        int i = i0;
        for (; i + 3 < i1; i += 4) {
          a0 = fmaf(G[(size_t)(i + 0) * n + t], vj[i + 0], a0);
          a1 = fmaf(G[(size_t)(i + 1) * n + t], vj[i + 1], a1);
          a2 = fmaf(G[(size_t)(i + 2) * n + t], vj[i + 2], a2);
          a3 = fmaf(G[(size_t)(i + 3) * n + t], vj[i + 3], a3);
        }
        for (; i < i1; ++i) a0 = fmaf(G[(size_t)i * n + t], vj[i], a0);
      }
      partial[part * 256 + t] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    float w = 0.f;
    if (owner)
      w = (partial[t] + partial[256 + t]) + (partial[512 + t] + partial[768 + t]);
    const double a = block_sum_vec(owner ? (double)w * v : 0.0, red);
    if (owner) {
      w = w - (float)a * v - (float)beta_prev * v_prev;
      wl[t] = w;
    }
    __syncthreads();
    // full re-orthogonalisation against V[0..j]: coefficients by 16 waves,
    // correction split over the 4 parts
    for (int i = wave; i <= j; i += kLanczosThreads / 64) {
      float p = 0.f;
      for (int c = lane; c < n; c += 64) p = fmaf(V[(size_t)i * n + c], wl[c], p);
      p = wave_sum(p);
      if (lane == 0) coef[i] = p;
    }
    __syncthreads();
    {
      float corr = 0.f;
      if (on)
        for (int i = part; i <= j; i += kLanczosParts)
          corr = fmaf(coef[i], V[(size_t)i * n + t], corr);
      partial[part * 256 + t] = corr;
    }
    __syncthreads();
    if (owner)
      w -= (partial[t] + partial[256 + t]) + (partial[512 + t] + partial[768 + t]);
    const double b = sqrt(block_sum_vec(owner ? (double)w * w : 0.0, red));
    if (tid == 0) {
      alpha[j] = a;
      beta[j] = b;
    }
    tscale = fmax(tscale, fmax(fabs(a), b));
    // Krylov space exhausted (rank-deficient Gram, invariant subspace) or a
    // NaN: what is left of w is rounding noise -- normalising it would feed
    // amplified garbage into the recurrence.  The tridiagonal matrix so far
    // already holds the spectrum of the reachable space.
    if (!(b > 1e-5 * tscale)) {
      if (tid == 0) steps_done = j + 1;
      break;
    }
    // every 8 steps: has the top Ritz value stopped moving?  (it grows
    // monotonically with the Krylov dimension)
    if ((j & 7) == 7) {
      __syncthreads();                 // alpha[j], beta[j] visible to wave 0
      if (wave == 0) {
        const double ritz = tridiagonal_lambda_max(alpha, beta, j + 1);
        if (tid == 0) {
          stop_flag = (last_ritz > 0.0 &&
                       fabs(ritz - last_ritz) <= 2e-8 * fabs(ritz)) ? 1 : 0;
          last_ritz = ritz;
        }
      }
      __syncthreads();
      if (stop_flag) {
        if (tid == 0) steps_done = j + 1;
        break;
      }
    }
    v_prev = v;
    v = (float)(w / b);
    beta_prev = b;
  }
  __syncthreads();

  if (wave == 0) {
    const double lambda = tridiagonal_lambda_max(alpha, beta, steps_done);
    const float lf = (float)lambda;
    if (tid == 0) {
      out[0] = lf;
      out[1] = 1.f / lf;  // the reference's `1. / lipschitz_constant` in f32
    }
  }
}

// ---------------------------------------------------------------- n > 256
// Same iteration for 256 < n <= 1024 (e.g. 20x20 or 32x32 patches): one
// thread per vector component, the Krylov basis (k x n floats, up to 512 KiB)
// in a global workspace instead of LDS.  ~2 ms at n = 1024, once per call.
constexpr int kLanczosLargeN = 1024;

__device__ __forceinline__ double block_sum_all(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double total = 0.0;
  for (int w = 0; w < kLanczosThreads / 64; ++w) total += red[w];
  return total;
}

__global__ __launch_bounds__(kLanczosThreads) void lanczos_large_kernel(
    const float* __restrict__ G, int n, int k, float* __restrict__ V,
    float* __restrict__ out) {
  __shared__ float vcur[kLanczosLargeN];
  __shared__ float wl[kLanczosLargeN];
  __shared__ float coef[kLanczosMaxK];
  __shared__ double red[kLanczosThreads / 64];
  __shared__ double alpha[kLanczosMaxK];
  __shared__ double beta[kLanczosMaxK];
  __shared__ int steps_done;
  __shared__ int stop_flag;
  __shared__ double last_ritz;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const bool on = t < n;

  float v = 0.f;
  if (on) {
    unsigned x = (unsigned)t * 2654435761u + 12345u;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    v = 0.5f + (float)(x & 0xffff) / 65536.f;
  }
  const double nrm = sqrt(block_sum_all((double)v * v, red));
  v = (float)(v / nrm);
  float v_prev = 0.f;
  double beta_prev = 0.0;
  if (t == 0) {
    steps_done = k;
    stop_flag = 0;
    last_ritz = -1.0;
  }
  double tscale = 0.0;
  __syncthreads();

  for (int j = 0; j < k; ++j) {
    if (on) {
      V[(size_t)j * n + t] = v;
      vcur[t] = v;
    }
    __syncthreads();
    // w = G v: G symmetric, so "column" t is read along rows, coalesced
    float w = 0.f;
    if (on) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int i = 0;
      for (; i + 3 < n; i += 4) {
        a0 = fmaf(G[(size_t)(i + 0) * n + t], vcur[i + 0], a0);
        a1 = fmaf(G[(size_t)(i + 1) * n + t], vcur[i + 1], a1);
        a2 = fmaf(G[(size_t)(i + 2) * n + t], vcur[i + 2], a2);
        a3 = fmaf(G[(size_t)(i + 3) * n + t], vcur[i + 3], a3);
      }
      for (; i < n; ++i) a0 = fmaf(G[(size_t)i * n + t], vcur[i], a0);
      w = (a0 + a1) + (a2 + a3);
    }
    const double a = block_sum_all(on ? (double)w * v : 0.0, red);
    if (on) {
      w = w - (float)a * v - (float)beta_prev * v_prev;
      wl[t] = w;
    }
    __threadfence_block();
    __syncthreads();
    // full re-orthogonalisation against V[0..j] (this block wrote them)
    for (int i = wave; i <= j; i += kLanczosThreads / 64) {
      float p = 0.f;
      for (int c = lane; c < n; c += 64) p = fmaf(V[(size_t)i * n + c], wl[c], p);
      p = wave_sum(p);
      if (lane == 0) coef[i] = p;
    }
    __syncthreads();
    if (on) {
      float corr = 0.f;
      for (int i = 0; i <= j; ++i) corr = fmaf(coef[i], V[(size_t)i * n + t], corr);
      w -= corr;
    }
    const double b = sqrt(block_sum_all(on ? (double)w * w : 0.0, red));
    if (t == 0) {
      alpha[j] = a;
      beta[j] = b;
    }
    tscale = fmax(tscale, fmax(fabs(a), b));
    if (!(b > 1e-5 * tscale)) {
      if (t == 0) steps_done = j + 1;
      break;
    }
    if ((j & 7) == 7) {
      __syncthreads();
      if (wave == 0) {
        const double ritz = tridiagonal_lambda_max(alpha, beta, j + 1);
        if (t == 0) {
          stop_flag = (last_ritz > 0.0 &&
                       fabs(ritz - last_ritz) <= 2e-8 * fabs(ritz)) ? 1 : 0;
          last_ritz = ritz;
        }
      }
      __syncthreads();
      if (stop_flag) {
        if (t == 0) steps_done = j + 1;
        break;
      }
    }
    v_prev = v;
    v = (float)(w / b);
    beta_prev = b;
  }
  __syncthreads();
  if (wave == 0) {
    const double lambda = tridiagonal_lambda_max(alpha, beta, steps_done);
    const float lf = (float)lambda;
    if (t == 0) {
      out[0] = lf;
      out[1] = 1.f / lf;
    }
  }
}

}  // namespace vtc

using namespace vtc;

extern "C" size_t vtc_lambda_max_workspace_bytes(int64_t n) {
  if (n <= kLanczosMaxN || n > kLanczosLargeN) return 256;
  return align_up((size_t)kLanczosMaxK * n * sizeof(float), 256);
}

// out: 2 floats on the device: [lambda_max, 1/lambda_max]
extern "C" int vtc_lambda_max(const float* symmetric, int64_t n, float* out,
                              void* workspace, size_t workspace_bytes,
                              void* stream) {
  VTC_REQUIRE(symmetric && out, "vtc_lambda_max: null pointer");
  VTC_REQUIRE(n > 0, "vtc_lambda_max: bad size");
  if (n > kLanczosLargeN) {
    set_error("vtc_lambda_max: n = %lld exceeds the single-workgroup Lanczos "
              "limit of %d", (long long)n, kLanczosLargeN);
    return VTC_ERR_UNSUPPORTED;
  }
  const int k = (int)(n < kLanczosMaxK ? n : kLanczosMaxK);
  if (n > kLanczosMaxN) {
    if (!workspace || workspace_bytes < vtc_lambda_max_workspace_bytes(n)) {
      set_error("vtc_lambda_max: workspace too small");
      return VTC_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(lanczos_large_kernel, dim3(1), dim3(kLanczosThreads), 0,
                       as_stream(stream), symmetric, (int)n, k,
                       static_cast<float*>(workspace), out);
    VTC_LAUNCH_CHECK();
    return VTC_OK;
  }
  const size_t lds =
      ((size_t)k * n + n + k + kLanczosParts * 256) * sizeof(float);
  static unsigned long long configured = 0;
  if (first_use_on_this_device(&configured)) {
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(lanczos_lambda_max_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  }
  hipLaunchKernelGGL(lanczos_lambda_max_kernel, dim3(1),
                     dim3(kLanczosThreads), lds,
                     as_stream(stream), symmetric, (int)n, k, out);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}
