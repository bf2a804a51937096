// Power-of-two operand scales of the f16 hi/lo split ("f16x3") outside the
// fused fully-connected kernels: f16 has 5 exponent bits, so an operand is
// brought to the range [16, 32) by a power of two before it is split (exact,
// and undone exactly on the f32 accumulators).  Static operands (dictionaries)
// get one scale per call; an operand that a kernel wrote in the previous launch
// gets its scale from max |x|, which that kernel left in device memory.
// Shared by conv_x3.h and gemm_x3.h.
#pragma once

#include "common.h"

namespace vtc {

// Slots in which a kernel leaves max |x| of what it wrote, for the kernel that
// reads it next: 32 words 128 bytes apart per quantity (a block adds to word
// blockIdx % 32, so that a few thousand atomics do not queue on one address),
// bit patterns of non-negative floats (ordered like the floats, NaN on top).
constexpr int kCxMaxWords = 32;
constexpr int kCxMaxStride = 32;                  // words between two slots
constexpr int kCxMaxSlotWords = kCxMaxWords * kCxMaxStride;

// Convolutional path: one word PER IMAGE and quantity (an image's scales come
// from that image alone, so a batch gives bit for bit what its images give one
// at a time); a block works on one image and adds one atomic.
struct CxScales {
  const float* dscale;      // {sigma_D, 1 / sigma_D}; null: bf16 mode
  const unsigned* r_in;     // [image] max |R| of the residual this launch reads
  unsigned* r_out;          // ... of the residual it writes
  unsigned* r_zero;         // cleared by this launch (the next writer's words)
  const unsigned* y_in;     // the same for the momentum iterate Y
  unsigned* y_out;
  unsigned* y_zero;
  int images;
};

// A maximum word only grows within a launch: look first (device-coherent
// load) and add the atomic only when it would raise the word.  A few hundred
// read-modify-writes on ONE address serialise at ~75 ns each
// (conv_partial_reduce_kernel, 298 blocks per image: 13 -> 35 us); with the
// look-ahead all but the first few blocks of an image skip theirs.
__device__ __forceinline__ void cx_raise_word(unsigned* word, unsigned v) {
  if (v > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(word, v);
}

// block-wide maximum of m (>= 0) into ONE word; `red` = 16 words of LDS nobody
// else uses around this call (the barriers are inside)
__device__ __forceinline__ void cx_publish_max_word(float m, unsigned* word,
                                                    unsigned* red) {
  unsigned v = __float_as_uint(m);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)v, off, 64);
    v = o > v ? o : v;
  }
  const int wave = threadIdx.x >> 6, waves = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < waves; ++w) v = red[w] > v ? red[w] : v;
    cx_raise_word(word, v);
  }
}

// the first block of a launch clears `count` words
__device__ __forceinline__ void cx_clear_words(unsigned* words, int count) {
  if (words && blockIdx.x == 0 && blockIdx.y == 0)
    for (int i = threadIdx.x; i < count; i += blockDim.x) words[i] = 0u;
}

// power of two that brings a maximum with these bits to [16, 32); 1 for zero,
// subnormal or non-finite maxima
__device__ __forceinline__ void cx_scale_of_bits(unsigned bits, float* s,
                                                 float* inv) {
  const int e = (int)((bits >> 23) & 0xffu);
  int field = 258 - e;                            // 127 + 4 - (e - 127)
  field = field < 2 ? 2 : (field > 252 ? 252 : field);
  const bool usable = e > 0 && e < 255;
  *s = usable ? __uint_as_float((unsigned)field << 23) : 1.f;
  *inv = usable ? __uint_as_float((unsigned)(254 - field) << 23) : 1.f;
}

// every lane of the calling wave gets the maximum over the slot's words
__device__ __forceinline__ unsigned cx_read_max(const unsigned* slot) {
  const int lane = threadIdx.x & 63;
  unsigned v = slot ? slot[(lane & (kCxMaxWords - 1)) * kCxMaxStride] : 0u;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// block-wide maximum of m (>= 0) added to the slot; `red` = 16 words of LDS
// nobody else uses around this call (the barriers are inside)
__device__ __forceinline__ void cx_publish_max(float m, unsigned* slot,
                                               unsigned* red) {
  unsigned v = __float_as_uint(m);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)v, off, 64);
    v = o > v ? o : v;
  }
  const int wave = threadIdx.x >> 6, waves = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < waves; ++w) v = red[w] > v ? red[w] : v;
    cx_raise_word(slot + (blockIdx.x & (kCxMaxWords - 1)) * kCxMaxStride, v);
  }
}

__device__ __forceinline__ void cx_clear_slot(unsigned* slot) {
  if (slot && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < kCxMaxWords)
    slot[threadIdx.x * kCxMaxStride] = 0u;
}

// {sigma, 1 / sigma} of a small array (the kernels): one block
static __global__ __launch_bounds__(1024) void cx_array_scale_kernel(
    const float* __restrict__ x, int64_t count, float* __restrict__ scale) {
  __shared__ unsigned red[16];
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < count; i += 1024) m = fmaxf(m, fabsf(x[i]));
  unsigned v = __float_as_uint(m);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)v, off, 64);
    v = o > v ? o : v;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) v = red[w] > v ? red[w] : v;
    cx_scale_of_bits(v, scale, scale + 1);
  }
}

// max |x| of a large array into a CxScales slot (warm start: the initial
// codes are the first synthesis operand)
static __global__ __launch_bounds__(256) void cx_array_max_kernel(
    const float* __restrict__ x, int64_t count, unsigned* __restrict__ slot) {
  __shared__ unsigned red[16];
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * 256)
    m = fmaxf(m, fabsf(x[i]));
  cx_publish_max(m, slot, red);
}

// the same per image (grid.y = image): one word each (CxScales)
static __global__ __launch_bounds__(256) void cx_image_max_kernel(
    const float* __restrict__ x, int64_t per_image,
    unsigned* __restrict__ words) {
  __shared__ unsigned red[16];
  const float* xi = x + (int64_t)blockIdx.y * per_image;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_image;
       i += (int64_t)gridDim.x * 256)
    m = fmaxf(m, fabsf(xi[i]));
  cx_publish_max_word(m, words + blockIdx.y, red);
}

// wave-level form for epilogue functors (no LDS, no barrier): one atomic per
// wave
__device__ __forceinline__ void cx_publish_max_wave(float m, unsigned* slot) {
  unsigned v = __float_as_uint(m);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)v, off, 64);
    v = o > v ? o : v;
  }
  if ((threadIdx.x & 63) == 0)
    cx_raise_word(slot + ((blockIdx.x + (threadIdx.x >> 6)) &
                          (kCxMaxWords - 1)) * kCxMaxStride, v);
}

}  // namespace vtc
