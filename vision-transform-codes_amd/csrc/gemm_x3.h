// Tiled contraction with float32-level accuracy on the bf16 matrix pipe
// ("bf16x3"), for shapes the fused FISTA kernel does not cover.
//
//   C[M,N] = A[M,K] * B[N,K]^T         A, B float32 in HBM, k contiguous
//
// While a tile is staged into LDS every f32 value is split into
//   hi = bf16(x),  lo = bf16(x - float(hi))
// and the product is formed as hi*hi + hi*lo + lo*hi with
// v_mfma_f32_32x32x16_bf16, f32 accumulate: relative error ~2^-16 per product
// (measured end to end: same as the reference's own f32 noise after 200 FISTA
// iterations) at 3 MFMA per algorithmic product, i.e. 5.3x the peak of the
// exact-f32 MFMA used by gemm_f32.h.
//
// Block = 4 waves (2x2), block tile 128x128, wave tile 64x64 (2x2 MFMA tiles),
// K step 32.  LDS rows are 64 B (32 bf16); the 16-byte chunk index is XORed
// with (row >> 2) & 3 so that the 16 lanes of a ds_read_b128 group (16 rows,
// same chunk) hit 16 different 16-byte slots.
#pragma once

#include "common.h"
#include "gemm_f32.h"
#include "x3_scale.h"

namespace vtc {

struct GemmX3Args {
  const float* A;
  const float* B;
  int64_t M, N, K;
  int64_t lda, ldb;
  int64_t k_chunk;   // K range per blockIdx.y slice (split-K), multiple of 32
  // f16 split (kernels instantiated with F16 = true; x3_scale.h): the B
  // operand enters as b_scale[0] * B (a per-call constant, b_scale[1] its
  // inverse), the A operand times the power of two that brings max |A| --
  // left in `a_max` by the kernel that wrote A -- to [16, 32); the
  // accumulators are scaled back before the epilogue sees them.  `clear`: a
  // slot this launch zeroes for a later writer (may be null).
  const float* b_scale = nullptr;
  const unsigned* a_max = nullptr;
  unsigned* clear = nullptr;
};

typedef _Float16 x3_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 x3_f16x8 __attribute__((ext_vector_type(8)));

template <bool F16>
__device__ __forceinline__ f32x16 x3_mfma(const uint4& a, const uint4& b,
                                          const f32x16& c) {
  if (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(
        __builtin_bit_cast(x3_f16x8, a), __builtin_bit_cast(x3_f16x8, b), c, 0,
        0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(
      __builtin_bit_cast(x3_bf16x8, a), __builtin_bit_cast(x3_bf16x8, b), c, 0,
      0, 0);
}

// four values (times a power-of-two scale) -> packed hi and lo parts
template <bool F16>
__device__ __forceinline__ void x3_split4(const float (&v)[4], float scale,
                                          uint2* hi_out, uint2* lo_out) {
  if (F16) {
    // two values per instruction: v_pk_mul_f32, v_cvt_pk_f16_f32 (round to
    // nearest even, as the scalar conversion), v_pk_add_f32 -- 3 VALU
    // instructions per element instead of 5; the same arithmetic
    typedef float pair_f32 __attribute__((ext_vector_type(2)));
    typedef _Float16 pair_f16 __attribute__((ext_vector_type(2)));
    unsigned hw[2], lw[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const pair_f32 x = {v[2 * k] * scale, v[2 * k + 1] * scale};
      const pair_f16 h = __builtin_convertvector(x, pair_f16);
      const pair_f16 l = __builtin_convertvector(
          x - __builtin_convertvector(h, pair_f32), pair_f16);
      hw[k] = __builtin_bit_cast(unsigned, h);
      lw[k] = __builtin_bit_cast(unsigned, l);
    }
    *hi_out = make_uint2(hw[0], hw[1]);
    *lo_out = make_uint2(lw[0], lw[1]);
  } else {
    x3_bf16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = (__bf16)v[k];
      lo[k] = (__bf16)(v[k] - (float)hi[k]);
    }
    *hi_out = __builtin_bit_cast(uint2, hi);
    *lo_out = __builtin_bit_cast(uint2, lo);
  }
}

// operand scales of a launch: A, B, and the factor that undoes both
template <bool F16>
__device__ __forceinline__ void x3_launch_scales(const GemmX3Args& g,
                                                 float* a_scale,
                                                 float* b_scale,
                                                 float* unscale) {
  *a_scale = *b_scale = *unscale = 1.f;
  if (F16) {
    float inv_a;
    cx_scale_of_bits(cx_read_max(g.a_max), a_scale, &inv_a);
    *b_scale = g.b_scale[0];
    *unscale = inv_a * g.b_scale[1];
    cx_clear_slot(g.clear);
  }
}

__device__ __forceinline__ int x3_lds_off(int row, int chunk) {
  return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

// global (f32) -> registers: 128 rows x 32 k, thread handles 4 float4
__device__ __forceinline__ void x3_stage_load(const float* P, int64_t ld,
                                              int64_t line0, int64_t lines,
                                              int64_t k0, int64_t K, int tid,
                                              float4 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = tid + i * 256;
    const int line = f >> 3, kq = f & 7;
    const int64_t gl = line0 + line, gk = k0 + kq * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gl < lines && gk < K) {
      const float* src = P + gl * ld + gk;
      if (gk + 3 < K) {
        v = *reinterpret_cast<const float4*>(src);
      } else {
        v.x = src[0];
        if (gk + 1 < K) v.y = src[1];
        if (gk + 2 < K) v.z = src[2];
      }
    }
    regs[i] = v;
  }
}

// registers -> LDS as 16-bit hi / lo parts
template <bool F16>
__device__ __forceinline__ void x3_stage_store(char* hi_base, char* lo_base,
                                               int tid,
                                               const float4 (&regs)[4],
                                               float scale) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = tid + i * 256;
    const int line = f >> 3, kq = f & 7;
    const float v[4] = {regs[i].x, regs[i].y, regs[i].z, regs[i].w};
    uint2 hi, lo;
    x3_split4<F16>(v, scale, &hi, &lo);
    const int off = x3_lds_off(line, kq >> 1) + 8 * (kq & 1);
    *reinterpret_cast<uint2*>(hi_base + off) = hi;
    *reinterpret_cast<uint2*>(lo_base + off) = lo;
  }
}

template <class Epi, bool F16>
__global__ __launch_bounds__(256) void gemm_x3_kernel(GemmX3Args g, Epi epi) {
  // [buf][A_hi, A_lo, B_hi, B_lo][128 rows][64 B]
  __shared__ __attribute__((aligned(16))) char lds[2][4][kX3TileBytes];
  resolve_epilogue(epi, 0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  // Block -> tile.  Workgroups are dealt round-robin to the 8 XCDs (one L2
  // each).  With few column tiles the blocks of one row block -- which read
  // the same rows of A -- are made to run on one XCD back to back: block L is
  // on XCD L % 8, slots L / 8 walk the column tiles of row block
  // (slot / tiles_n) * 8 + XCD.  (Residual product of configs[3], two column
  // tiles: A was fetched twice, FETCH_SIZE 139 -> 82 MB per launch.)  With
  // many column tiles the plain order already keeps a column tile (B operand)
  // on one XCD, and the paired order costs more than it saves (measured on
  // the gradient product, 32 column tiles: FETCH_SIZE 173 -> 219 MB).
  const int64_t tiles_n = (g.N + kX3BN - 1) / kX3BN;
  const int64_t tiles_m = (g.M + kX3BM - 1) / kX3BM;
  int64_t tile_m, tile_n;
  if (tiles_n <= 8) {
    const int64_t slot = (int64_t)blockIdx.x >> 3;
    tile_m = (slot / tiles_n) * 8 + (blockIdx.x & 7);
    tile_n = slot % tiles_n;
  } else {
    tile_m = (int64_t)blockIdx.x / tiles_n;
    tile_n = (int64_t)blockIdx.x % tiles_n;
  }
  float a_scale, b_scale, unscale;
  x3_launch_scales<F16>(g, &a_scale, &b_scale, &unscale);
  if (tile_m >= tiles_m) return;                   // whole block (grid padding)
  const int64_t m0 = tile_m * kX3BM;
  const int64_t n0 = tile_n * kX3BN;
  const int z = blockIdx.y;
  const int64_t k_begin = (int64_t)z * g.k_chunk;
  const int64_t k_end = (k_begin + g.k_chunk < g.K) ? k_begin + g.k_chunk : g.K;
  const int nk = (int)((k_end - k_begin + kX3BK - 1) / kX3BK);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr bool kPipe = epi_prefetch<Epi>::value;
  float pre0[epi_prefetch_floats<Epi>::value],
      pre1[epi_prefetch_floats<Epi>::value];
  auto ctx = [&] {
    if constexpr (kPipe) return epi.begin(m0, g.M); else return 0;
  }();
  if constexpr (kPipe) epi.load(ctx, wm * 64, n0 + wn * 64, lane, g.N, pre0);

  float4 ra[4], rb[4];
  x3_stage_load(g.A, g.lda, m0, g.M, k_begin, k_end, tid, ra);
  x3_stage_load(g.B, g.ldb, n0, g.N, k_begin, k_end, tid, rb);
  x3_stage_store<F16>(lds[0][0], lds[0][1], tid, ra, a_scale);
  x3_stage_store<F16>(lds[0][2], lds[0][3], tid, rb, b_scale);
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) {
      const int64_t k0 = k_begin + (int64_t)(kt + 1) * kX3BK;
      x3_stage_load(g.A, g.lda, m0, g.M, k0, k_end, tid, ra);
      x3_stage_load(g.B, g.ldb, n0, g.N, k0, k_end, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int chunk = 2 * kk + half;
      uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int arow = wm * 64 + t * 32 + l31;
        const int brow = wn * 64 + t * 32 + l31;
        ah[t] = *reinterpret_cast<const uint4*>(lds[cur][0] +
                                                x3_lds_off(arow, chunk));
        al[t] = *reinterpret_cast<const uint4*>(lds[cur][1] +
                                                x3_lds_off(arow, chunk));
        bh[t] = *reinterpret_cast<const uint4*>(lds[cur][2] +
                                                x3_lds_off(brow, chunk));
        bl[t] = *reinterpret_cast<const uint4*>(lds[cur][3] +
                                                x3_lds_off(brow, chunk));
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[mi][ni] = x3_mfma<F16>(ah[mi], bh[ni], acc[mi][ni]);
          acc[mi][ni] = x3_mfma<F16>(ah[mi], bl[ni], acc[mi][ni]);
          acc[mi][ni] = x3_mfma<F16>(al[mi], bh[ni], acc[mi][ni]);
        }
      }
    }
    if (more) {
      x3_stage_store<F16>(lds[cur ^ 1][0], lds[cur ^ 1][1], tid, ra, a_scale);
      x3_stage_store<F16>(lds[cur ^ 1][2], lds[cur ^ 1][3], tid, rb, b_scale);
    }
    __syncthreads();
    cur ^= 1;
  }
  if (F16) {                                       // back to the caller's units
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= unscale;
  }

  if constexpr (kPipe) {
    const int r0 = wm * 64;
    const int64_t c0 = n0 + wn * 64;
    // the operand tiles are dead after the K loop (last barrier above): each
    // wave stages its accumulator tiles through its own 8 KiB of them
    float* scratch = reinterpret_cast<float*>(&lds[0][0][0]) + wave * 2048;
    epi.load(ctx, r0, c0 + 32, lane, g.N, pre1);
    epi.finish(ctx, r0, c0, lane, g.N, acc[0][0], pre0, scratch);
    epi.load(ctx, r0 + 32, c0, lane, g.N, pre0);
    epi.finish(ctx, r0, c0 + 32, lane, g.N, acc[0][1], pre1, scratch);
    epi.load(ctx, r0 + 32, c0 + 32, lane, g.N, pre1);
    epi.finish(ctx, r0 + 32, c0, lane, g.N, acc[1][0], pre0, scratch);
    epi.finish(ctx, r0 + 32, c0 + 32, lane, g.N, acc[1][1], pre1, scratch);
  } else if constexpr (epi_whole_tile<Epi>::value) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        epi.tile(m0 + wm * 64 + mi * 32, n0 + wn * 64 + ni * 32, lane,
                 acc[mi][ni], g.M, g.N);
  } else {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int64_t col = n0 + wn * 64 + ni * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row =
              m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (row < g.M && col < g.N) epi(row, col, acc[mi][ni][r], z);
        }
      }
    }
  }
  epi.block_end();
}

// Eight-wave variant for plain (per element) epilogues: same 128 x 128 x 32
// block tile and LDS layout, waves 4 x 2 with 32 x 64 wave tiles.  Half the
// accumulator and staging registers per wave let four waves share a SIMD (16
// per CU instead of 8): the residual product of the subspace / tiled FC paths
// runs latency bound on the operand loads with 8.
template <class Epi, bool F16>
__global__ __launch_bounds__(512) void gemm_x3_kernel8(GemmX3Args g, Epi epi) {
  __shared__ __attribute__((aligned(16))) char lds[2][4][kX3TileBytes];
  resolve_epilogue(epi, 0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;          // 4 x 2
  const int l31 = lane & 31, half = lane >> 5;
  const int64_t tiles_n = (g.N + kX3BN - 1) / kX3BN;
  const int64_t tiles_m = (g.M + kX3BM - 1) / kX3BM;
  int64_t tile_m, tile_n;
  if (tiles_n <= 8) {                               // see gemm_x3_kernel
    const int64_t slot = (int64_t)blockIdx.x >> 3;
    tile_m = (slot / tiles_n) * 8 + (blockIdx.x & 7);
    tile_n = slot % tiles_n;
  } else {
    tile_m = (int64_t)blockIdx.x / tiles_n;
    tile_n = (int64_t)blockIdx.x % tiles_n;
  }
  float a_scale, b_scale, unscale;
  x3_launch_scales<F16>(g, &a_scale, &b_scale, &unscale);
  if (tile_m >= tiles_m) return;
  const int64_t m0 = tile_m * kX3BM;
  const int64_t n0 = tile_n * kX3BN;
  const int z = blockIdx.y;
  const int64_t k_begin = (int64_t)z * g.k_chunk;
  const int64_t k_end = (k_begin + g.k_chunk < g.K) ? k_begin + g.k_chunk : g.K;
  const int nk = (int)((k_end - k_begin + kX3BK - 1) / kX3BK);

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // Operand tiles go global -> registers -> (split) -> LDS; 128 rows x 32 k
  // per operand = 1024 float4, 2 per thread.  The loads are buffer loads:
  // rows past the matrix are past the end of the resource and k past the
  // slice gets an out-of-range offset, both read as zero -- no branches, so
  // the compiler can count outstanding loads (s_waitcnt vmcnt(N)) and the
  // loads of K step kt+2 really stay in flight while step kt computes (with
  // guarded loads in their own basic blocks every wait was vmcnt(0) and each
  // step paid a full memory round trip).
  auto tile_rsrc = [&](const float* P, int64_t ld, int64_t line0,
                       int64_t lines) {
    const int64_t left = lines - line0 < 128 ? lines - line0 : 128;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(P + line0 * ld), 0,
                                             (int)(left * ld * 4), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t ars = tile_rsrc(g.A, g.lda, m0, g.M);
  const __amdgpu_buffer_rsrc_t brs = tile_rsrc(g.B, g.ldb, n0, g.N);
  auto stage_load = [&](const __amdgpu_buffer_rsrc_t& rs, int64_t ld,
                        int64_t k0, x3_u32x4 (&regs)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + i * 512;
      const int line = f >> 3, kq = f & 7;
      const int64_t gk = k0 + kq * 4;
      const unsigned vo = gk < k_end ? (unsigned)((line * ld + gk) * 4)
                                     : 0x80000000u;
      regs[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0);
    }
  };
  auto stage_store = [&](char* hi_base, char* lo_base,
                         const x3_u32x4 (&regs)[2], float scale) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + i * 512;
      const int line = f >> 3, kq = f & 7;
      const float v[4] = {__uint_as_float(regs[i][0]),
                          __uint_as_float(regs[i][1]),
                          __uint_as_float(regs[i][2]),
                          __uint_as_float(regs[i][3])};
      uint2 hi, lo;
      x3_split4<F16>(v, scale, &hi, &lo);
      const int off = x3_lds_off(line, kq >> 1) + 8 * (kq & 1);
      *reinterpret_cast<uint2*>(hi_base + off) = hi;
      *reinterpret_cast<uint2*>(lo_base + off) = lo;
    }
  };
  auto compute = [&](int cur) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int chunk = 2 * kk + half;
      const int arow = wm * 32 + l31;
      const uint4 ah = *reinterpret_cast<const uint4*>(lds[cur][0] +
                                                       x3_lds_off(arow, chunk));
      const uint4 al = *reinterpret_cast<const uint4*>(lds[cur][1] +
                                                       x3_lds_off(arow, chunk));
      uint4 bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int brow = wn * 64 + t * 32 + l31;
        bh[t] = *reinterpret_cast<const uint4*>(lds[cur][2] +
                                                x3_lds_off(brow, chunk));
        bl[t] = *reinterpret_cast<const uint4*>(lds[cur][3] +
                                                x3_lds_off(brow, chunk));
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        acc[ni] = x3_mfma<F16>(ah, bh[ni], acc[ni]);
        acc[ni] = x3_mfma<F16>(ah, bl[ni], acc[ni]);
        acc[ni] = x3_mfma<F16>(al, bh[ni], acc[ni]);
      }
    }
  };
  auto k_of = [&](int kt) { return k_begin + (int64_t)kt * kX3BK; };

  // two register stages: loads of step kt+2 are issued during step kt
  x3_u32x4 a0[2], b0[2], a1[2], b1[2];
  stage_load(ars, g.lda, k_of(0), a0);
  stage_load(brs, g.ldb, k_of(0), b0);
  stage_store(lds[0][0], lds[0][1], a0, a_scale);
  stage_store(lds[0][2], lds[0][3], b0, b_scale);
  __syncthreads();
  if (nk > 1) { stage_load(ars, g.lda, k_of(1), a0); stage_load(brs, g.ldb, k_of(1), b0); }
  if (nk > 2) { stage_load(ars, g.lda, k_of(2), a1); stage_load(brs, g.ldb, k_of(2), b1); }
  for (int kt = 0; kt < nk; kt += 2) {
    compute(0);                                          // step kt
    if (kt + 1 < nk) {
      stage_store(lds[1][0], lds[1][1], a0, a_scale);    // operands of kt+1
      stage_store(lds[1][2], lds[1][3], b0, b_scale);
    }
    __syncthreads();
    if (kt + 3 < nk) { stage_load(ars, g.lda, k_of(kt + 3), a0); stage_load(brs, g.ldb, k_of(kt + 3), b0); }
    if (kt + 1 < nk) {
      compute(1);                                        // step kt+1
      if (kt + 2 < nk) {
        stage_store(lds[0][0], lds[0][1], a1, a_scale);  // operands of kt+2
        stage_store(lds[0][2], lds[0][3], b1, b_scale);
      }
      __syncthreads();
      if (kt + 4 < nk) { stage_load(ars, g.lda, k_of(kt + 4), a1); stage_load(brs, g.ldb, k_of(kt + 4), b1); }
    }
  }

#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t col = n0 + wn * 64 + ni * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row < g.M && col < g.N)
        epi(row, col, F16 ? acc[ni][r] * unscale : acc[ni][r], z);
    }
  }
  epi.block_end();
}

// Usable when both operands allow aligned 16-byte loads along k.
static inline bool gemm_x3_usable(const float* A, int64_t lda, const float* B,
                                  int64_t ldb) {
  return gemm_vec_ok(A, lda) && gemm_vec_ok(B, ldb);
}

// The f16 split of a launch (GemmX3Args): all null = bf16 split.
struct X3Scale {
  const float* b_scale = nullptr;
  const unsigned* a_max = nullptr;
  unsigned* clear = nullptr;
};

// k_slices > 1 => split-K: the epilogue receives the slice index (EpiSlab).
template <class Epi>
static int launch_gemm_x3(const float* A, int64_t lda, const float* B,
                          int64_t ldb, int64_t M, int64_t N, int64_t K,
                          Epi epi, hipStream_t st, int k_slices = 1,
                          X3Scale sc = X3Scale()) {
  if (M <= 0 || N <= 0) return VTC_OK;
  if (k_slices < 1) k_slices = 1;
  int64_t chunk = ceil_div(ceil_div(K, k_slices), kX3BK) * kX3BK;
  if (chunk < kX3BK) chunk = kX3BK;
  GemmX3Args g{A, B, M, N, K, lda, ldb, chunk};
  g.b_scale = sc.b_scale;
  g.a_max = sc.a_max;
  g.clear = sc.clear;
  const bool f16 = sc.b_scale != nullptr;
  // row blocks padded to a multiple of 8 (see the block -> tile map)
  const int64_t tiles = ceil_div(ceil_div(M, kX3BM), 8) * 8 *
                        ceil_div(N, kX3BN);
  if (tiles > 0x7fffffffLL) {
    set_error("gemm_x3: too many tiles");
    return VTC_ERR_INVALID_ARGUMENT;
  }
  const dim3 grid((unsigned)tiles, (unsigned)ceil_div(K, chunk));
  if constexpr (!epi_whole_tile<Epi>::value) {
    // plain epilogues: the eight-wave kernel (same-box A/B: tiled FC path
    // 6.02 -> 5.65 ms, configs[3] 41.5 -> 41.2 ms)
    if (f16)
      hipLaunchKernelGGL((gemm_x3_kernel8<Epi, true>), grid, dim3(512), 0, st,
                         g, epi);
    else
      hipLaunchKernelGGL((gemm_x3_kernel8<Epi, false>), grid, dim3(512), 0, st,
                         g, epi);
    VTC_LAUNCH_CHECK();
    return VTC_OK;
  }
  if (f16)
    hipLaunchKernelGGL((gemm_x3_kernel<Epi, true>), grid, dim3(256), 0, st, g,
                       epi);
  else
    hipLaunchKernelGGL((gemm_x3_kernel<Epi, false>), grid, dim3(256), 0, st, g,
                       epi);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

// slices actually launched for (K, k_slices)
static inline int gemm_x3_slices(int64_t K, int k_slices) {
  if (k_slices < 1) k_slices = 1;
  int64_t chunk = ceil_div(ceil_div(K, k_slices), kX3BK) * kX3BK;
  if (chunk < kX3BK) chunk = kX3BK;
  return (int)ceil_div(K, chunk);
}

// how many K slices give a contraction with few output tiles enough blocks:
// one resident set (2 blocks per CU x 256 CUs) -- measured on configs[3]:
// 256 / 384 / 512 / 1024 / 2048 blocks -> 45.9 / 44.0 / 41.9 / 44.9 / 51.3 ms
static inline int gemm_x3_want_slices(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ceil_div(M, kX3BM) * ceil_div(N, kX3BN);
  int64_t want = ceil_div(512, tiles);
  const int64_t cap = ceil_div(K, 256);
  if (want > cap) want = cap;
  if (want > 16) want = 16;
  if (want < 1) want = 1;
  return gemm_x3_slices(K, (int)want);
}

// out = sum_z slabs[z] - X   (fixed order; the residual of a split-K product)
// max_out (may be null): x3_scale.h slot that receives max |out|
int launch_slab_reduce_minus(const float* slabs, int slices, int64_t count,
                             const float* X, float* out, hipStream_t st,
                             unsigned* max_out = nullptr);

}  // namespace vtc
