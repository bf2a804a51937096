// Micro-benchmark: cost of 16-byte LDS reads at 2-byte alignment (the window
// operand of the convolutional analysis product) against aligned ones.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_unaligned.hip -o /tmp/lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

struct __attribute__((packed, aligned(2))) U16 { uint4 v; };

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint16_t buf[32768];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 32768; i += 512) buf[i] = (uint16_t)i;
  __syncthreads();
  const int l31 = lane & 31, half = lane >> 5;
  int off;   // element offset
  if (MODE == 0) off = lane * 8;                          // aligned, contiguous
  else if (MODE == 1) off = l31 + 8 * half;               // conv pattern, 2-byte aligned
  else if (MODE == 2) off = (l31 + 8 * half) & ~7;        // same addresses rounded to 16 B
  else if (MODE == 3) off = (l31 + 8 * half) & ~3;        // 8-byte aligned
  else if (MODE == 4) off = l31 * 16 + 8 * half;          // 32-byte lane stride, halves interleaved
  else off = l31 * 72 + 8 * half;                         // 144-byte lane stride (synthesis operand rows)
  unsigned acc = 0;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = off + ((it * 16 + r) & 63) * (MODE >= 4 ? 64 : 48);
      if (MODE == 3) {
        const uint2 a = *reinterpret_cast<const uint2*>(buf + o);
        const uint2 b = *reinterpret_cast<const uint2*>(buf + o + 4);
        acc += a.x ^ a.y ^ b.x ^ b.y;
      } else {
        const U16 u = *reinterpret_cast<const U16*>(buf + o);
        acc += u.v.x ^ u.v.y ^ u.v.z ^ u.v.w;
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) atomicAdd(out, t1 - t0);
  if (acc == 0x12345678u) out[1] = acc;
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 16);
  const int iters = 2000;
  const char* names[6] = {"aligned b128, contiguous", "2-byte aligned b128 (conv window)",
                          "same, rounded to 16 B", "8-byte aligned, 2 x b64",
                          "b128, lane stride 32 B (l31), +16 B (half)",
                          "b128, lane stride 144 B (l31), +16 B (half)"};
  for (int mode = 0; mode < 6; ++mode) {
    for (int waves = 1; waves <= 8; waves *= 2) {
      hipMemset(d, 0, 16);
      const dim3 grid(256), block(64 * waves);
      if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, d, iters);
      if (mode == 3) hipLaunchKernelGGL(k<3>, grid, block, 0, 0, d, iters);
      if (mode == 4) hipLaunchKernelGGL(k<4>, grid, block, 0, 0, d, iters);
      if (mode == 5) hipLaunchKernelGGL(k<5>, grid, block, 0, 0, d, iters);
      unsigned long long h = 0;
      hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      const double per_wave = (double)h / (256.0 * waves);
      printf("%-36s %d waves/CU: %7.1f cycles per 16-byte read per wave, %6.1f per CU-read\n",
             names[mode], waves, per_wave / (iters * 16.0),
             per_wave / (iters * 16.0) / waves);
    }
  }
  return 0;
}
