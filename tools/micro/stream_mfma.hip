// How fast can a CU stream dictionary fragments from L2 into MFMA operands
// with the instruction mix of the fused FISTA kernel (csrc/fc_fused.hip), and
// does a second wave per SIMD help?
//
// One workgroup per CU.  Every wave walks its share of a 2 MiB L2-resident
// buffer ("hi" and "lo" fragment arrays, 1 KiB per wave-instruction) through a
// rolling register ring of RING fragment pairs; each pair feeds three
// 32x32x16 f16 MFMAs whose B operands come from LDS (two ds_read_b128 per
// pair, fetched one step ahead) -- exactly steps 1 / 3 of the kernel without
// its epilogue, barriers and exchanges.  Per "phase" a CU moves 256 KiB and
// issues 384 MFMAs (96 per SIMD x 32 cycles = 3072 cycles), whatever WAVES is.
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/stream_mfma.hip -o tools/micro/stream_mfma
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                        \
  do {                                                                  \
    hipError_t e_ = (x);                                                \
    if (e_ != hipSuccess) {                                             \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                    \
      return 1;                                                         \
    }                                                                   \
  } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 load16(__amdgpu_buffer_rsrc_t rs, unsigned voff,
                                        unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

constexpr int kPairsPerPhase = 128;      // 128 pairs x 2 KiB = 256 KiB per CU

template <int WAVES, int RING, bool MFMA>
__global__ __launch_bounds__(64 * WAVES) void mix_kernel(
    const uint4* __restrict__ hi, const uint4* __restrict__ lo, int phases,
    int buffer_pairs, float* out, unsigned long long* cycles) {
  __shared__ __attribute__((aligned(16))) char ldsb[2 * 16896];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 2 * 16896 / 4; i += 64 * WAVES)
    reinterpret_cast<float*>(ldsb)[i] = 0.001f * (float)(i & 255);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(
      (void*)hi, 0, buffer_pairs * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(
      (void*)lo, 0, buffer_pairs * 1024, 0x00020000);
  const unsigned voff = (unsigned)lane * 16u;
  constexpr int PER_WAVE = kPairsPerPhase / WAVES;   // pairs per wave and phase
  const int rd = (lane & 31) * 528 + 16 * (lane >> 5);
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  uint4 ring[2][RING];
  // pair j of this wave in phase p sits at fragment ((p % P) * 128 + wave *
  // PER_WAVE + j) of the buffer
  const int buffer_phases = buffer_pairs / kPairsPerPhase;
  auto frag_off = [&](int p, int j) -> unsigned {
    return (unsigned)(((p % buffer_phases) * kPairsPerPhase + wave * PER_WAVE +
                       j) * 1024);
  };
#pragma unroll
  for (int i = 0; i < RING; ++i) {
    ring[0][i] = load16(rh, voff, frag_off(0, i));
    ring[1][i] = load16(rl, voff, frag_off(0, i));
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int p = 0; p < phases; ++p) {
    uint4 b_next[2];
    b_next[0] = *reinterpret_cast<const uint4*>(ldsb + rd);
    b_next[1] = *reinterpret_cast<const uint4*>(ldsb + 16896 + rd);
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
      uint4 b[2] = {b_next[0], b_next[1]};
      if (j + 1 < PER_WAVE) {
        b_next[0] =
            *reinterpret_cast<const uint4*>(ldsb + rd + 32 * ((j + 1) & 15));
        b_next[1] = *reinterpret_cast<const uint4*>(ldsb + 16896 + rd +
                                                    32 * ((j + 1) & 15));
      }
      const uint4 ah = ring[0][j % RING], al = ring[1][j % RING];
      if (MFMA) {
        f32x16& a = acc[j & 1];
        a = __builtin_amdgcn_mfma_f32_32x32x16_f16(
            __builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, b[0]), a,
            0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_f16(
            __builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, b[1]), a,
            0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_f16(
            __builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, b[0]), a,
            0, 0, 0);
      } else {
        acc[j & 1][0] += __uint_as_float(ah.x ^ al.y ^ b[0].z ^ b[1].w);
      }
      const int jn = j + RING;
      const unsigned off = jn < PER_WAVE ? frag_off(p, jn)
                                         : frag_off(p + 1, jn - PER_WAVE);
      ring[0][j % RING] = load16(rh, voff, off);
      ring[1][j % RING] = load16(rl, voff, off);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) sum += acc[t][e];
#pragma unroll
  for (int i = 0; i < RING; ++i) sum += __uint_as_float(ring[0][i].x ^ ring[1][i].y);
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int WAVES, int RING, bool MFMA>
static int run(int cus, const uint4* hi, const uint4* lo, int buffer_pairs,
               float* out, unsigned long long* cyc) {
  const int phases = 4000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((mix_kernel<WAVES, RING, MFMA>), dim3(cus), dim3(64 * WAVES),
                     0, 0, hi, lo, 50, buffer_pairs, out, cyc);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((mix_kernel<WAVES, RING, MFMA>), dim3(cus), dim3(64 * WAVES),
                     0, 0, hi, lo, phases, buffer_pairs, out, cyc);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(cus);
  CHECK(hipMemcpy(h.data(), cyc, (size_t)cus * 8, hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto c : h) mean += (double)c;
  mean /= cus;
  const double bytes = 256.0 * 1024 * phases;
  printf("%d waves/CU, ring %2d pairs (%3d KiB in flight per CU), %s: %6.0f "
         "cycles per 256 KiB phase = %5.1f B/clk/CU, %.0f MHz, %.2f ms\n",
         WAVES, RING, WAVES * RING * 2, MFMA ? "3 MFMA per pair" : "no MFMA    ",
         mean / phases, bytes / mean, mean / ms / 1e3, ms);
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int buffer_pairs = 1024;           // 1 MiB hi + 1 MiB lo: the 1024-atom
                                           // dictionary's two packings
  uint4 *hi, *lo;
  float* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&hi, (size_t)buffer_pairs * 1024));
  CHECK(hipMalloc(&lo, (size_t)buffer_pairs * 1024));
  CHECK(hipMemset(hi, 0x11, (size_t)buffer_pairs * 1024));
  CHECK(hipMemset(lo, 0x12, (size_t)buffer_pairs * 1024));
  CHECK(hipMalloc(&out, (size_t)cus * 512 * sizeof(float)));
  CHECK(hipMalloc(&cyc, (size_t)cus * 8));
  printf("fragment stream + MFMA mix of the fused FISTA kernel, %d CUs "
         "(pure MFMA time per phase: 3072 cycles)\n", cus);
  if (run<4, 8, true>(cus, hi, lo, buffer_pairs, out, cyc)) return 1;
  if (run<4, 16, true>(cus, hi, lo, buffer_pairs, out, cyc)) return 1;
  if (run<8, 4, true>(cus, hi, lo, buffer_pairs, out, cyc)) return 1;
  if (run<8, 8, true>(cus, hi, lo, buffer_pairs, out, cyc)) return 1;
  if (run<8, 16, true>(cus, hi, lo, buffer_pairs, out, cyc)) return 1;
  if (run<4, 8, false>(cus, hi, lo, buffer_pairs, out, cyc)) return 1;
  if (run<4, 16, false>(cus, hi, lo, buffer_pairs, out, cyc)) return 1;
  if (run<8, 4, false>(cus, hi, lo, buffer_pairs, out, cyc)) return 1;
  if (run<8, 8, false>(cus, hi, lo, buffer_pairs, out, cyc)) return 1;
  if (run<8, 16, false>(cus, hi, lo, buffer_pairs, out, cyc)) return 1;
  return 0;
}
