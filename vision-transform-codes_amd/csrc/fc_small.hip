// ISTA / FISTA for 8x8 patches (n = 64) against 64 .. 192 atoms with everything
// on the CU: BASELINE configs[0]'s shape.  On the tiled path this shape is
// HBM-bound -- each iteration moves the (b, s) state through memory about six
// times for 4 s n = 16-32 kflop per patch -- here a wave keeps 32 patches'
// iterate and codes in registers for all iterations and the dictionary lives in
// LDS (restates analysis_transforms/fully_connected/ista_fista.py:82-148).
//
// Exact f32 arithmetic on the f32 matrix pipe (v_mfma_f32_32x32x2_f32), both
// products transposed so that lanes are patches:
//   R^T (64 px x 32 patches)    = D^T Y^T - X^T      A = D^T (LDS), B = Y^T
//   G^T (32 atoms x 32 patches) = D   R^T            A = D   (LDS), B = R^T
// The B operand of each product IS the accumulator tile of the other: a lane
// of an accumulator tile holds, in register r, row (r & 3) + 8 (r >> 2) + 4 half
// of its column (its patch) -- exactly one k-step (2 rows, one per half-wave)
// of the next product when the A operand is packed in that row order.  No
// transposition, no LDS traffic for the state; one 4-byte LDS read per MFMA.
#include "fc_small.h"
#include "fc_fused.h"

namespace vtc {

typedef float sm_f32x16 __attribute__((ext_vector_type(16)));

bool small_shape_supported(int64_t n, int64_t s) {
  // (256 atoms: 336 live accumulator-layout registers plus the working set do
  // not allocate without spills; that shape stays on the tiled path)
  return n == 64 && (s == 64 || s == 128 || s == 192);
}

struct SmallParams {
  const float* images;      // (b, 64)
  const float* dictionary;  // (s, 64)
  const float* init;        // (b, s) or null
  float* codes;             // (b, s)
  const float* betas;       // FISTA momentum table (device)
  const float* eta_dev;     // or null
  float eta, lam;
  int64_t b;
  int num_iters, fista;
};

// row of an accumulator tile held by (register r, half-wave h)
__device__ __forceinline__ int sm_row(int r, int h) {
  return (r & 3) + 8 * (r >> 2) + 4 * h;
}

// ST = s / 32 atom tiles
template <int ST, int MODE>
__global__ __launch_bounds__(256) void fc_small_kernel(SmallParams P) {
  constexpr int S = 32 * ST;
  extern __shared__ __attribute__((aligned(16))) float sm_lds[];
  float* PA = sm_lds;             // [ST][16][2][64]  D[atom(t, r, h)][px]
  float* PB = sm_lds + S * 64;    // [2][16][2][S]    D[atom][px(u, r, h)]
  const int tid = threadIdx.x;
  for (int i = tid; i < S * 64; i += 256) {
    const int px = i & 63, h = (i >> 6) & 1, r = (i >> 7) & 15, t = i >> 11;
    PA[i] = P.dictionary[(32 * t + sm_row(r, h)) * 64 + px];
  }
  for (int i = tid; i < S * 64; i += 256) {
    const int a = i % S, rest = i / S;
    const int h = rest & 1, r = (rest >> 1) & 15, u = rest >> 5;
    PB[i] = P.dictionary[a * 64 + 32 * u + sm_row(r, h)];
  }
  __syncthreads();
  const float eta = P.eta_dev ? *P.eta_dev : P.eta;
  const float cutoff = mul_rn(P.lam, eta);
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int64_t tasks = (P.b + 31) / 32;
  for (int64_t task = (int64_t)blockIdx.x * 4 + wave; task < tasks;
       task += (int64_t)gridDim.x * 4) {
    const int64_t p = task * 32 + l31;
    const bool valid = p < P.b;
    // this lane's patch, in the row order of the accumulator tiles
    constexpr bool kKeepX = true;
    sm_f32x16 X[2], Y[ST], C[ST];
    auto load_patch = [&](sm_f32x16 (&dst)[2]) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
          if (valid)
            x = *reinterpret_cast<const float4*>(P.images + p * 64 + 32 * u +
                                                 8 * q + 4 * half);
          dst[u][4 * q] = x.x; dst[u][4 * q + 1] = x.y;
          dst[u][4 * q + 2] = x.z; dst[u][4 * q + 3] = x.w;
        }
    };
    if (kKeepX) load_patch(X);
#pragma unroll
    for (int t = 0; t < ST; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid && P.init)
          c = *reinterpret_cast<const float4*>(P.init + p * S + 32 * t + 8 * q +
                                               4 * half);
        C[t][4 * q] = c.x; C[t][4 * q + 1] = c.y;
        C[t][4 * q + 2] = c.z; C[t][4 * q + 3] = c.w;
        Y[t][4 * q] = c.x; Y[t][4 * q + 1] = c.y;
        Y[t][4 * q + 2] = c.z; Y[t][4 * q + 3] = c.w;
      }
    for (int k = 0; k < P.num_iters; ++k) {
      const float beta = P.fista ? P.betas[k] : 0.f;
      // The operand reads do not depend on k: for up to 128 atoms the compiler
      // keeps all 256 operand values of a lane in registers (no LDS traffic
      // in the loop at all); beyond that they would spill, so the lane offset
      // is made opaque and the reads stay in the loop.
      int lane_off = l31;
      if (ST > 4) asm volatile("" : "+v"(lane_off));
      // R^T = D^T Y^T - X^T
      sm_f32x16 R0, R1;
      if (!kKeepX) load_patch(X);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        R0[r] = -X[0][r];
        R1[r] = -X[1][r];
      }
#pragma unroll
      for (int t = 0; t < ST; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* a = PA + ((t * 16 + r) * 2 + half) * 64 + lane_off;
          R0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], Y[t][r], R0, 0, 0, 0);
          R1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[32], Y[t][r], R1, 0, 0, 0);
        }
      // G^T = D R^T per atom tile, gradient step, threshold, extrapolation
#pragma unroll
      for (int t = 0; t < ST; ++t) {
        sm_f32x16 G;
#pragma unroll
        for (int r = 0; r < 16; ++r) G[r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* a = PB + ((0 * 16 + r) * 2 + half) * S + 32 * t + lane_off;
          G = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], R0[r], G, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* a = PB + ((1 * 16 + r) * 2 + half) * S + 32 * t + lane_off;
          G = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], R1[r], G, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pr = sub_rn(Y[t][r], mul_rn(eta, G[r]));
          const float c = shrink(pr, cutoff, MODE);
          // beta = 0 (ISTA) takes the codes themselves, as the reference does
          Y[t][r] = beta != 0.f ? add_rn(c, mul_rn(beta, sub_rn(c, C[t][r])))
                                : c;
          C[t][r] = c;
        }
      }
    }
    if (valid) {
#pragma unroll
      for (int t = 0; t < ST; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(P.codes + p * S + 32 * t + 8 * q +
                                     4 * half) =
              make_float4(C[t][4 * q], C[t][4 * q + 1], C[t][4 * q + 2],
                          C[t][4 * q + 3]);
    }
  }
}

template <int ST, int MODE>
static int launch_small(const SmallParams& P, hipStream_t st) {
  const size_t lds = (size_t)2 * 32 * ST * 64 * sizeof(float);
  static unsigned long long configured = 0;
  if (first_use_on_this_device(&configured))
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(fc_small_kernel<ST, MODE>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess)
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount,
                                dev);
  const int64_t tasks = (P.b + 31) / 32;
  int64_t blocks = (tasks + 3) / 4;
  if (blocks > 2 * (int64_t)cus) blocks = 2 * (int64_t)cus;
  hipLaunchKernelGGL((fc_small_kernel<ST, MODE>), dim3((unsigned)blocks),
                     dim3(256), lds, st, P);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

template <int ST>
static int launch_small_mode(const SmallParams& P, int threshold,
                             hipStream_t st) {
  switch (threshold) {
    case VTC_SOFT: return launch_small<ST, VTC_SOFT>(P, st);
    case VTC_SOFT_NONNEG: return launch_small<ST, VTC_SOFT_NONNEG>(P, st);
    case VTC_HARD: return launch_small<ST, VTC_HARD>(P, st);
    default: return launch_small<ST, VTC_HARD_NONNEG>(P, st);
  }
}

int run_small(const float* images, const float* dictionary,
              const float* initial_codes, float* codes, int64_t b, int64_t n,
              int64_t s, float eta, const float* eta_dev,
              float sparsity_weight, int num_iters, int variant,
              int threshold, int* iters_run, hipStream_t st) {
  if (!small_shape_supported(n, s) || num_iters > fused_max_iters()) {
    set_error("on-chip small-patch kernel: unsupported shape");
    return VTC_ERR_UNSUPPORTED;
  }
  const float* betas = fista_beta_table_on_this_device();
  if (!betas) {
    set_error("small-patch kernel: could not place the momentum table on the "
              "device");
    return VTC_ERR_HIP;
  }
  SmallParams P;
  P.images = images;
  P.dictionary = dictionary;
  P.init = initial_codes;
  P.codes = codes;
  P.betas = betas;
  P.eta_dev = eta_dev;
  P.eta = eta;
  P.lam = sparsity_weight;
  P.b = b;
  P.num_iters = num_iters;
  P.fista = (variant == VTC_FISTA) ? 1 : 0;
  int rc;
  switch (s) {
    case 64: rc = launch_small_mode<2>(P, threshold, st); break;
    case 128: rc = launch_small_mode<4>(P, threshold, st); break;
    default: rc = launch_small_mode<6>(P, threshold, st); break;
  }
  if (rc == VTC_OK && iters_run) *iters_run = num_iters;
  return rc;
}

}  // namespace vtc
