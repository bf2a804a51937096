// Shared host/device helpers for libvtc_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vtc_hip.h"

namespace vtc {

// ---- error plumbing --------------------------------------------------------
void set_error(const char* fmt, ...);

#define VTC_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ::vtc::set_error(__VA_ARGS__);      \
      return VTC_ERR_INVALID_ARGUMENT;    \
    }                                     \
  } while (0)

#define VTC_HIP_CHECK(expr)                                               \
  do {                                                                    \
    hipError_t e_ = (expr);                                               \
    if (e_ != hipSuccess) {                                               \
      ::vtc::set_error("%s failed: %s (%s:%d)", #expr,                    \
                       hipGetErrorString(e_), __FILE__, __LINE__);        \
      return VTC_ERR_HIP;                                                 \
    }                                                                     \
  } while (0)

#define VTC_LAUNCH_CHECK() VTC_HIP_CHECK(hipGetLastError())

static inline hipStream_t as_stream(void* s) {
  return reinterpret_cast<hipStream_t>(s);
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) {
  return (v + a - 1) / a * a;
}

// Carves 256-byte aligned pieces out of the caller's workspace.
struct Carver {
  char* base;
  size_t used;
  explicit Carver(void* p) : base(static_cast<char*>(p)), used(0) {}
  template <class T>
  T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base + used);
    used += align_up(count * sizeof(T), 256);
    return p;
  }
};

// Function attributes (dynamic LDS above 64 KiB) are per device and a process
// may drive several: true the first time the calling site runs on the current
// device.  `seen` is a static word of the call site (bit = device ordinal; a
// concurrent first call only configures twice).
static inline bool first_use_on_this_device(unsigned long long* seen) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
  const unsigned long long bit = 1ull << dev;
  if (*seen & bit) return false;
  *seen |= bit;
  return true;
}

// ---- device helpers -----------------------------------------------------
// The reference's elementwise arithmetic is un-fused f32 (one rounding per
// torch op).  These wrappers keep hipcc from contracting a*b+c into an FMA.
__device__ __forceinline__ float mul_rn(float a, float b) {
  return __fmul_rn(a, b);
}
__device__ __forceinline__ float add_rn(float a, float b) {
  return __fadd_rn(a, b);
}
__device__ __forceinline__ float sub_rn(float a, float b) {
  return __fsub_rn(a, b);
}

// clamp_(min=0) that lets NaN through like torch does
__device__ __forceinline__ float clamp_min0(float v) {
  return (v < 0.f) ? 0.f : v;
}

// torch.sign: (0 < x) - (x < 0)
__device__ __forceinline__ float sign_of(float v) {
  return static_cast<float>((0.f < v) - (v < 0.f));
}

// The four thresholding flavours of
// analysis_transforms/fully_connected/ista_fista.py:107-120.
__device__ __forceinline__ float shrink(float c, float cutoff, int mode) {
  switch (mode) {
    case VTC_SOFT: {
      float sgn = sign_of(c);
      float mag = clamp_min0(sub_rn(fabsf(c), cutoff));
      return mul_rn(mag, sgn);
    }
    case VTC_SOFT_NONNEG:
      return clamp_min0(sub_rn(c, cutoff));
    case VTC_HARD:
      return (fabsf(c) < cutoff) ? 0.f : c;
    default:  // VTC_HARD_NONNEG
      return (c < cutoff) ? 0.f : c;
  }
}

// Epilogue functors that can take their step size from device memory expose
// resolve(); the contraction kernels call it once per thread before any use.
template <class E>
__device__ __forceinline__ auto resolve_epilogue(E& e, int)
    -> decltype(e.resolve(), void()) {
  e.resolve();
}
template <class E>
__device__ __forceinline__ void resolve_epilogue(E&, long) {}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace vtc
