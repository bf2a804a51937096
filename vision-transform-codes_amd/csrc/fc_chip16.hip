// ISTA / FISTA for 12x12 patches (n = 144) against 288 or 576 atoms -- and 8x8
// patches against 256 or 512, beyond what fc_small.hip holds -- with the state
// on the CU: a wave keeps 16 patches' iterate and codes in registers for
// all iterations (36 atom tiles x 4 registers x 2), the dictionary streams from
// L2 as the A operand of both products.  Same construction as fc_small.hip on
// the 16x16x4 f32 matrix instruction: both products transposed so that lanes
// are patches,
//   R^T (144 px x 16 patches)   = D^T Y^T - X^T      A = D^T (packed), B = Y^T
//   G^T (16 atoms x 16 patches) = D   R^T            A = D (as given), B = R^T
// and the accumulator tile of one product is the B operand of the other: lane
// (patch, row group g) holds rows 4 g + r in register r, which is k-step r of a
// product whose A operand is read with k = 4 g + r.  For the second product
// that A operand is 4 consecutive floats of a dictionary row (16 bytes from
// the caller's layout), for the first a packed copy (pack kernel, once per
// call).  Exact f32 arithmetic.  On the tiled path this shape moves the (b, s)
// state through HBM about six times per iteration.
#include "fc_small.h"
#include "fc_fused.h"

namespace vtc {

typedef float c16_f32x4 __attribute__((ext_vector_type(4)));

bool chip16_shape_supported(int64_t n, int64_t s) {
  return (n == 144 && (s == 288 || s == 576)) ||
         (n == 64 && (s == 256 || s == 512));
}

size_t chip16_workspace_bytes(int64_t n, int64_t s) {
  return chip16_shape_supported(n, s) ? align_up((size_t)s * n * 4, 256) + 256
                                      : 256;
}

// PA[(t * NT + mt) * 64 + lane] (float4): D[16 t + 4 g + r][16 mt + m],
// r = 0..3, lane = 16 g + m
__global__ void chip16_pack_kernel(const float* __restrict__ D,
                                   float4* __restrict__ PA, int s, int n) {
  const int nt = n / 16, total = (s / 16) * nt * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += gridDim.x * blockDim.x) {
    const int lane = i & 63, mt = (i >> 6) % nt, t = (i >> 6) / nt;
    const int m = lane & 15, g = lane >> 4;
    const float* src = D + (16 * t + 4 * g) * n + 16 * mt + m;
    PA[i] = make_float4(src[0], src[n], src[2 * n], src[3 * n]);
  }
}

struct Chip16Params {
  const float* images;      // (b, n)
  const float* dictionary;  // (s, n)
  const float4* packA;
  const float* init;        // (b, s) or null
  float* codes;             // (b, s)
  const float* betas;
  const float* eta_dev;
  float eta, lam;
  int64_t b;
  int num_iters, fista;
};

template <int NT, int ST, int MODE>
__global__ __launch_bounds__(256) void fc_chip16_kernel(Chip16Params P) {
  constexpr int N = 16 * NT, S = 16 * ST;
  const float eta = P.eta_dev ? *P.eta_dev : P.eta;
  const float cutoff = mul_rn(P.lam, eta);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int64_t tasks = (P.b + 15) / 16;
  for (int64_t task = (int64_t)blockIdx.x * 4 + wave; task < tasks;
       task += (int64_t)gridDim.x * 4) {
    const int64_t p = task * 16 + m;
    const bool valid = p < P.b;
    // (the patch itself is read again in every iteration -- 576 bytes per
    // lane from L2 -- rather than held in 36 registers: those go to the
    // double buffer of the dictionary operands)
    c16_f32x4 Y[ST], C[ST];
#pragma unroll
    for (int t = 0; t < ST; ++t) {
      float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid && P.init)
        c = *reinterpret_cast<const float4*>(P.init + p * S + 16 * t + 4 * g);
      C[t][0] = c.x; C[t][1] = c.y; C[t][2] = c.z; C[t][3] = c.w;
      Y[t] = C[t];
    }
    for (int k = 0; k < P.num_iters; ++k) {
      const float beta = P.fista ? P.betas[k] : 0.f;
      // (the operand addresses do not depend on k: keep them from being hoisted)
      int lane_op = lane;
      asm volatile("" : "+v"(lane_op));
      const float4* pa = P.packA + lane_op;
      const float* drow = P.dictionary + (lane_op & 15) * N + 4 * (lane_op >> 4);
      // Operands of step j = one atom tile of the first product (j < ST:
      // packed D^T) or of the second (j >= ST: 16 bytes of a dictionary row).
      // The loads of step j + 1 are issued before the products of step j.
      float4 a[2][NT];
      auto load_step = [&](int j, float4 (&dst)[NT]) {
#pragma unroll
        for (int q = 0; q < NT; ++q)
          dst[q] = j < ST ? pa[(j * NT + q) * 64]
                          : *reinterpret_cast<const float4*>(
                                drow + 16 * (j - ST) * N + 16 * q);
      };
      c16_f32x4 R[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid)
          x = *reinterpret_cast<const float4*>(P.images + p * N + 16 * u +
                                               4 * g);
        R[u][0] = -x.x; R[u][1] = -x.y; R[u][2] = -x.z; R[u][3] = -x.w;
      }
      load_step(0, a[0]);
      // R^T += D^T (atom tile j) Y^T (tile j)
#pragma unroll
      for (int j = 0; j < ST; ++j) {
        load_step(j + 1, a[(j + 1) & 1]);            // j + 1 == ST: first G step
        __builtin_amdgcn_sched_barrier(0);           // keep the loads up here
        const float4(&cur)[NT] = a[j & 1];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
          R[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[mt].x, Y[j][0], R[mt], 0, 0, 0);
          R[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[mt].y, Y[j][1], R[mt], 0, 0, 0);
          R[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[mt].z, Y[j][2], R[mt], 0, 0, 0);
          R[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[mt].w, Y[j][3], R[mt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // G^T (atom tile t) = D R^T, gradient step, threshold, extrapolation
#pragma unroll
      for (int t = 0; t < ST; ++t) {
        if (t + 1 < ST) load_step(ST + t + 1, a[(ST + t + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const float4(&cur)[NT] = a[(ST + t) & 1];
        // two accumulation chains: a dependent MFMA waits for its predecessor
        c16_f32x4 G0 = {0.f, 0.f, 0.f, 0.f}, G1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          G0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u].x, R[u][0], G0, 0, 0, 0);
          G1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u].y, R[u][1], G1, 0, 0, 0);
          G0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u].z, R[u][2], G0, 0, 0, 0);
          G1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u].w, R[u][3], G1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gsum = add_rn(G0[r], G1[r]);
          const float pr = sub_rn(Y[t][r], mul_rn(eta, gsum));
          const float c = shrink(pr, cutoff, MODE);
          Y[t][r] = beta != 0.f ? add_rn(c, mul_rn(beta, sub_rn(c, C[t][r])))
                                : c;
          C[t][r] = c;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (valid) {
#pragma unroll
      for (int t = 0; t < ST; ++t)
        *reinterpret_cast<float4*>(P.codes + p * S + 16 * t + 4 * g) =
            make_float4(C[t][0], C[t][1], C[t][2], C[t][3]);
    }
  }
}

template <int NT, int ST, int MODE>
static int launch_chip16(const Chip16Params& P, hipStream_t st) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess)
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount,
                                dev);
  const int64_t tasks = (P.b + 15) / 16;
  int64_t blocks = (tasks + 3) / 4;
  if (blocks > (int64_t)cus) blocks = cus;
  hipLaunchKernelGGL((fc_chip16_kernel<NT, ST, MODE>), dim3((unsigned)blocks),
                     dim3(256), 0, st, P);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

template <int NT, int ST>
static int launch_chip16_mode(const Chip16Params& P, int threshold,
                              hipStream_t st) {
  switch (threshold) {
    case VTC_SOFT: return launch_chip16<NT, ST, VTC_SOFT>(P, st);
    case VTC_SOFT_NONNEG: return launch_chip16<NT, ST, VTC_SOFT_NONNEG>(P, st);
    case VTC_HARD: return launch_chip16<NT, ST, VTC_HARD>(P, st);
    default: return launch_chip16<NT, ST, VTC_HARD_NONNEG>(P, st);
  }
}

int run_chip16(const float* images, const float* dictionary,
               const float* initial_codes, float* codes, int64_t b, int64_t n,
               int64_t s, float eta, const float* eta_dev,
               float sparsity_weight, int num_iters, int variant,
               int threshold, void* workspace, size_t workspace_bytes,
               int* iters_run, hipStream_t st) {
  if (!chip16_shape_supported(n, s) || num_iters > fused_max_iters()) {
    set_error("on-chip 12x12 kernel: unsupported shape");
    return VTC_ERR_UNSUPPORTED;
  }
  if (!workspace || workspace_bytes < chip16_workspace_bytes(n, s)) {
    set_error("on-chip 12x12 kernel: workspace too small");
    return VTC_ERR_WORKSPACE;
  }
  const float* betas = fista_beta_table_on_this_device();
  if (!betas) {
    set_error("on-chip 12x12 kernel: could not place the momentum table on "
              "the device");
    return VTC_ERR_HIP;
  }
  float4* packA = static_cast<float4*>(workspace);
  hipLaunchKernelGGL(chip16_pack_kernel, dim3(128), dim3(256), 0, st,
                     dictionary, packA, (int)s, (int)n);
  VTC_LAUNCH_CHECK();
  Chip16Params P;
  P.images = images;
  P.dictionary = dictionary;
  P.packA = packA;
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
  if (n == 144)
    rc = (s == 288) ? launch_chip16_mode<9, 18>(P, threshold, st)
                    : launch_chip16_mode<9, 36>(P, threshold, st);
  else
    rc = (s == 256) ? launch_chip16_mode<4, 16>(P, threshold, st)
                    : launch_chip16_mode<4, 32>(P, threshold, st);
  if (rc == VTC_OK && iters_run) *iters_run = num_iters;
  return rc;
}

}  // namespace vtc
