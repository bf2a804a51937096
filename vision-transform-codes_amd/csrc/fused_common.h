// Device helpers shared by the fused persistent kernels (fc_fused.hip: state on
// chip; fused_stream.hip: state streamed): fragment types, the hi/lo operand
// split, the MFMA wrapper, dictionary packing into MFMA fragment order.
#pragma once
#include "common.h"

namespace vtc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));

__host__ __device__ static inline int64_t ceil_div_dev(int64_t a, int64_t b) {
  return (a + b - 1) / b;
}

constexpr int kFP = 32;    // patches per workgroup
constexpr int kFN = 256;   // pixels per patch
constexpr int kPhaseAtoms = 128;

// ---------------------------------------------------------------- packing
// packA fragment (tile t of 32 atoms, k-step ks over pixels), lane l:
//   D[32t + (l&31)][16ks + 8(l>>5) + j],  j = 0..7
// packT fragment (phase p, pixel block nb of 32, k-step ks over the phase's
// atoms), lane l:
//   D[128p + 16ks + 8(l>>5) + j][32nb + (l&31)]
// LO = 0 stores bf16(x), LO = 1 stores bf16(x - float(bf16(x))).
template <bool F16>
__device__ __forceinline__ unsigned short split_part(float x, int lo) {
  if (F16) {
    const _Float16 hi = (_Float16)x;
    const _Float16 r = lo ? (_Float16)(x - (float)hi) : hi;
    return __builtin_bit_cast(unsigned short, r);
  }
  const __bf16 hi = (__bf16)x;
  const __bf16 r = lo ? (__bf16)(x - (float)hi) : hi;
  return __builtin_bit_cast(unsigned short, r);
}

// F16: sigma_D = 2^(8 - floor(log2 max|D|)), so that max |sigma_D D| lies in
// [256, 512): far from the f16 overflow (65504) and with the lo parts of all
// but vanishing entries in the normal range.  One block; scale[0] = sigma_D,
// scale[1] = 1 / sigma_D.
static __global__ __launch_bounds__(1024) void dictionary_scale_kernel(
    const float* __restrict__ D, int64_t count, float* __restrict__ scale) {
  __shared__ float part[16];
  float m = 0.f;
  // count = s * 256: 16-byte loads, four independent maxima in flight
  const float4* D4 = reinterpret_cast<const float4*>(D);
  float m1 = 0.f, m2 = 0.f, m3 = 0.f;
  for (int64_t i = threadIdx.x; i < count / 4; i += 1024) {
    const float4 v = D4[i];
    m = fmaxf(m, fabsf(v.x));
    m1 = fmaxf(m1, fabsf(v.y));
    m2 = fmaxf(m2, fabsf(v.z));
    m3 = fmaxf(m3, fabsf(v.w));
  }
  m = fmaxf(fmaxf(m, m1), fmaxf(m2, m3));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; ++k) m = fmaxf(m, part[k]);
    int e = 0;
    if (m > 0.f && m < __builtin_inff()) e = ilogbf(m);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    scale[0] = ldexpf(1.f, 8 - e);
    scale[1] = ldexpf(1.f, e - 8);
  }
}

// Both parts in one pass (loA / loT null: hi part only).
template <bool F16>
__global__ void pack_dictionary_kernel(const float* __restrict__ D, int s,
                                       unsigned short* __restrict__ packA,
                                       unsigned short* __restrict__ packT,
                                       unsigned short* __restrict__ loA,
                                       unsigned short* __restrict__ loT,
                                       const float* __restrict__ scale) {
  const float sg = F16 ? scale[0] : 1.f;
  const int64_t frags = (int64_t)s * kFN / 8;  // 16-byte units per packing
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < frags;
       u += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(u & 63);
    const int r = l & 31, h = l >> 5;
    {
      const int64_t f = u >> 6;  // = t*16 + ks
      const int t = (int)(f >> 4), ks = (int)(f & 15);
      const float* src = D + (int64_t)(32 * t + r) * kFN + 16 * ks + 8 * h;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = src[j] * sg;
        packA[u * 8 + j] = split_part<F16>(v, 0);
        if (loA) loA[u * 8 + j] = split_part<F16>(v, 1);
      }
    }
    {
      const int64_t f = u >> 6;  // = (p*8 + nb)*8 + ks
      const int ks = (int)(f & 7), nb = (int)((f >> 3) & 7), p = (int)(f >> 6);
      const float* src =
          D + (int64_t)(128 * p + 16 * ks + 8 * h) * kFN + 32 * nb + r;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = src[(int64_t)j * kFN] * sg;
        packT[u * 8 + j] = split_part<F16>(v, 0);
        if (loT) loT[u * 8 + j] = split_part<F16>(v, 1);
      }
    }
  }
}

template <int MODE>
__device__ __forceinline__ float shrink_fast(float c, float cutoff) {
  if (MODE == VTC_SOFT) {
    // sign(c) * max(|c| - t, 0) == c - clamp(c, -t, t) bit for bit (up to the
    // sign of a zero result): one v_med3 + one v_sub.
    return sub_rn(c, __builtin_amdgcn_fmed3f(c, -cutoff, cutoff));
  }
  return shrink(c, cutoff, MODE);
}

__device__ __forceinline__ bf16x8 as_frag(const uint4& u) {
  return __builtin_bit_cast(bf16x8, u);
}

__device__ __forceinline__ f16x8 as_frag16(const uint4& u) {
  return __builtin_bit_cast(f16x8, u);
}
template <bool F16>
__device__ __forceinline__ f32x16v mfma_frag(const uint4& a, const uint4& b,
                                             const f32x16v& c) {
  if (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag16(a), as_frag16(b),
                                                  c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a), as_frag(b), c, 0,
                                                 0, 0);
}

// four f32 values -> their 16-bit hi parts and (NP == 2) lo parts, packed
template <bool F16, int NP>
__device__ __forceinline__ void split4(const float (&v)[4], uint2* hi_out,
                                       uint2* lo_out) {
  if (F16) {
    // two values per instruction (v_cvt_pk_f16_f32, round to nearest even as
    // the scalar conversion; v_pk_add_f32): the same arithmetic in fewer VALU
    // instructions
    typedef float pair_f32 __attribute__((ext_vector_type(2)));
    typedef _Float16 pair_f16 __attribute__((ext_vector_type(2)));
    unsigned hw[2], lw[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const pair_f32 x = {v[2 * k], v[2 * k + 1]};
      const pair_f16 h = __builtin_convertvector(x, pair_f16);
      hw[k] = __builtin_bit_cast(unsigned, h);
      if (NP == 2)
        lw[k] = __builtin_bit_cast(
            unsigned, __builtin_convertvector(
                          x - __builtin_convertvector(h, pair_f32), pair_f16));
    }
    *hi_out = make_uint2(hw[0], hw[1]);
    if (NP == 2) *lo_out = make_uint2(lw[0], lw[1]);
  } else {
    bf16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = (__bf16)v[k];
      if (NP == 2) lo[k] = (__bf16)(v[k] - (float)hi[k]);
    }
    *hi_out = __builtin_bit_cast(uint2, hi);
    if (NP == 2) *lo_out = __builtin_bit_cast(uint2, lo);
  }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 buffer_load16(__amdgpu_buffer_rsrc_t rsrc,
                                               unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// In-kernel stamps (diagnostic instantiation only, STAMP = true): where a
// phase spends its cycles.  s_memtime + its wait in one statement, fenced.
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

}  // namespace vtc
