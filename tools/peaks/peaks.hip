// Measured peaks of the box ("confirm on the box with a
// pure-MFMA loop and a copy kernel"): dense bf16 / f32 MFMA rate, HBM copy
// bandwidth, and the L2 -> VGPR streaming rate of a CU (the bound of the fused
// FISTA kernel).  Stand-alone:  hipcc -O3 --offload-arch=gfx950 peaks.hip -o peaks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// 4 independent accumulators per wave, `iters` x 4 MFMAs
__global__ __launch_bounds__(256) void mfma_bf16_kernel(float* out, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * j); }
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void mfma_f32_kernel(float* out, int iters) {
  float a = 0.001f * threadIdx.x, b = 0.5f;
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// round-1/2 form, one 16-byte load in flight per lane: 4.5-4.6 TB/s, NOT the
// ceiling; kept for comparison
__global__ void copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

// U independent 16-byte loads in flight per lane (a block moves U x 4 KiB per
// trip), optional non-temporal accesses: the copy rate the HBM floors of
// DESIGN.md are quoted against.  MODE 0: copy, 1: read only, 2: write only.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NT, int MODE>
__global__ __launch_bounds__(256) void stream_hbm_kernel(const f32x4* __restrict__ in,
                                                         f32x4* __restrict__ out, size_t n,
                                                         float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t base = (size_t)blockIdx.x * U * 256 + threadIdx.x; base < n;
       base += (size_t)gridDim.x * U * 256) {
    f32x4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const size_t i = base + (size_t)j * 256;
      if (MODE == 2) { v[j] = (f32x4){1.f, 2.f, 3.f, 4.f}; continue; }
      if (i < n) v[j] = NT ? __builtin_nontemporal_load(in + i) : in[i];
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const size_t i = base + (size_t)j * 256;
      if (MODE == 1) { acc += v[j]; continue; }
      if (i < n) { if (NT) __builtin_nontemporal_store(v[j], out + i); else out[i] = v[j]; }
    }
  }
  if (MODE == 1 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <int U, bool NT, int MODE>
static float time_stream_hbm(const f32x4* a, f32x4* b, size_t n, int blocks, float* sink,
                             hipEvent_t e0, hipEvent_t e1) {
  hipLaunchKernelGGL((stream_hbm_kernel<U, NT, MODE>), dim3(blocks), dim3(256), 0, 0, a, b, n, sink);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r)
    hipLaunchKernelGGL((stream_hbm_kernel<U, NT, MODE>), dim3(blocks), dim3(256), 0, 0, a, b, n, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}

// every wave streams the whole `bytes` buffer (L2 resident) with 16-byte loads,
// 1 KiB per wave instruction, 16 loads in flight per wave, `passes` times;
// one workgroup of WAVES waves per CU
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void stream_kernel(
    const uint4* __restrict__ buf, size_t frags, int passes, unsigned* out,
    unsigned long long* cycles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int p = 0; p < passes; ++p)
    for (size_t f = wave * 16; f + 16 <= frags; f += 16 * WAVES) {
      uint4 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = buf[(f + j) * 64 + lane];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int WAVES>
static void run_stream(int cus, size_t kib, hipEvent_t e0, hipEvent_t e1);

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device: %s, %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float* out; CHECK(hipMalloc(&out, (size_t)cus * 16 * 256 * sizeof(float)));

  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int blocks = cus * waves_per_simd;   // 4 waves per block = 1 per SIMD
    const int iters = 20000;
    hipLaunchKernelGGL(mfma_bf16_kernel, dim3(blocks), dim3(256), 0, 0, out, 100);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_bf16_kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    double flops = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
    float ms = time_ms(e0, e1);
    printf("bf16 MFMA 32x32x16, %d wave(s)/SIMD: %.1f TFLOP/s (%.2f ms)\n", waves_per_simd, flops / ms / 1e9, ms);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_f32_kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    flops = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 2;
    ms = time_ms(e0, e1);
    printf("f32  MFMA 32x32x2,  %d wave(s)/SIMD: %.1f TFLOP/s (%.2f ms)\n", waves_per_simd, flops / ms / 1e9, ms);
  }

  {
    const size_t bytes = (size_t)2 << 30;
    float4 *a, *b; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 1, bytes));
    hipLaunchKernelGGL(copy_kernel, dim3(cus * 16), dim3(256), 0, 0, a, b, bytes / 16);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r)
      hipLaunchKernelGGL(copy_kernel, dim3(cus * 16), dim3(256), 0, 0, a, b, bytes / 16);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    const float ms = time_ms(e0, e1) / 5;
    printf("HBM copy 2 GiB -> 2 GiB, 1 load in flight per lane (round-1/2 figure): %.2f TB/s read+write (%.2f ms)\n", 2.0 * bytes / ms / 1e9, ms);
    const f32x4* A = reinterpret_cast<const f32x4*>(a);
    f32x4* B = reinterpret_cast<f32x4*>(b);
    const size_t n = bytes / 16;
    const int full4 = (int)(n / (4 * 256)), full8 = (int)(n / (8 * 256));
    struct { const char* name; float ms; double moved; } rows[] = {
        {"copy  4 loads in flight, plain, one tile per block", time_stream_hbm<4, false, 0>(A, B, n, full4, out, e0, e1), 2.0 * bytes},
        {"copy  4 loads in flight, nt,    one tile per block", time_stream_hbm<4, true, 0>(A, B, n, full4, out, e0, e1), 2.0 * bytes},
        {"copy  8 loads in flight, nt,    one tile per block", time_stream_hbm<8, true, 0>(A, B, n, full8, out, e0, e1), 2.0 * bytes},
        {"copy  4 loads in flight, nt,    persistent 16/CU  ", time_stream_hbm<4, true, 0>(A, B, n, cus * 16, out, e0, e1), 2.0 * bytes},
        {"copy  8 loads in flight, nt,    persistent 8/CU   ", time_stream_hbm<8, true, 0>(A, B, n, cus * 8, out, e0, e1), 2.0 * bytes},
        {"read  8 loads in flight, nt,    one tile per block", time_stream_hbm<8, true, 1>(A, B, n, full8, out, e0, e1), 1.0 * bytes},
        {"read  8 loads in flight, plain, one tile per block", time_stream_hbm<8, false, 1>(A, B, n, full8, out, e0, e1), 1.0 * bytes},
        {"write 8 stores per lane, nt,    one tile per block", time_stream_hbm<8, true, 2>(A, B, n, full8, out, e0, e1), 1.0 * bytes},
        {"write 8 stores per lane, plain, one tile per block", time_stream_hbm<8, false, 2>(A, B, n, full8, out, e0, e1), 1.0 * bytes},
    };
    for (auto& r : rows)
      printf("HBM %s: %.2f TB/s (%.2f ms)\n", r.name, r.moved / r.ms / 1e9, r.ms);
    CHECK(hipFree(a)); CHECK(hipFree(b));
  }

  for (size_t kib : {512, 2048}) {
    run_stream<4>(cus, kib, e0, e1);
    run_stream<8>(cus, kib, e0, e1);
  }
  return 0;
}

static float time_ms2(hipEvent_t a, hipEvent_t b) { float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }

template <int WAVES>
static void run_stream(int cus, size_t kib, hipEvent_t e0, hipEvent_t e1) {
  const size_t bytes = kib * 1024, frags = bytes / 1024;
  uint4* buf; CHECK(hipMalloc(&buf, bytes)); CHECK(hipMemset(buf, 3, bytes));
  unsigned* o; CHECK(hipMalloc(&o, (size_t)cus * 64 * WAVES * 4));
  unsigned long long* cyc; CHECK(hipMalloc(&cyc, (size_t)cus * 8));
  const int passes = 200;
  hipLaunchKernelGGL(stream_kernel<WAVES>, dim3(cus), dim3(64 * WAVES), 0, 0, buf, frags, 2, o, cyc);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(stream_kernel<WAVES>, dim3(cus), dim3(64 * WAVES), 0, 0, buf, frags, passes, o, cyc);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  const float ms = time_ms2(e0, e1);
  std::vector<unsigned long long> h(cus);
  CHECK(hipMemcpy(h.data(), cyc, (size_t)cus * 8, hipMemcpyDeviceToHost));
  double mean = 0; for (auto c : h) mean += (double)c; mean /= cus;
  const double per_cu_bytes = (double)bytes * passes;
  printf("L2 -> VGPR stream, %zu KiB buffer, 1 block of %d waves per CU: %.1f B/clk/CU (shader cycles), "
         "%.2f TB/s aggregate, %.0f MHz implied\n", kib, WAVES, per_cu_bytes / mean,
         per_cu_bytes * cus / ms / 1e9, mean / ms / 1e3);
  CHECK(hipFree(buf)); CHECK(hipFree(o)); CHECK(hipFree(cyc));
}
