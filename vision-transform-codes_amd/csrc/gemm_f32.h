// Exact-f32 tiled contraction on the gfx950 matrix cores.
//
// v_mfma_f32_32x32x2_f32 takes f32 operands and accumulates in f32: per output
// element the result is a k-ordered fmaf chain, i.e. the same arithmetic class
// as the reference's torch.mm in float32.  This is the "parity" contraction
// every plugin can fall back to for any shape; the bf16 kernels in
// fc_fista_fused.hip are the fast path for the headline shape.
//
// Block = 256 threads = 4 waves in a 2x2 arrangement, block tile 128x128,
// wave tile 64x64 (2x2 MFMA tiles, 64 accumulator VGPRs), K step 16.
// Operands are staged k-major in LDS ([k][row], pitch 132 floats) so that the
// 32 lanes of a half-wave read 32 consecutive dwords: conflict-free
// ds_read_b32 for both A (lane -> row) and B (lane -> column).
#pragma once

#include "common.h"
#include "x3_scale.h"

#include <type_traits>

namespace vtc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kGemmBM = 128;
constexpr int kGemmBN = 128;
constexpr int kGemmBK = 16;
constexpr int kGemmPitch = 132;
constexpr int kGemmThreads = 256;

// An epilogue that declares `static constexpr bool kWholeTile` receives each
// 32x32 accumulator tile at once through
//   tile(row0, col0, lane, acc, rows, cols)
// (lane l holds column col0 + (l & 31), register r holds row
//  row0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)), every lane of the wave
// calling it, so it may combine columns with lane shuffles.
template <class E, class = void>
struct epi_whole_tile : std::false_type {};
template <class E>
struct epi_whole_tile<E, std::void_t<decltype(E::kWholeTile)>>
    : std::true_type {};
// A whole-tile epilogue that also declares `kPrefetch` (floats per lane it
// wants to read per tile) is run as a pipeline over the wave's four tiles:
//   ctx = begin(m0, M)                 once per block
//   load(ctx, row_in_block, col0, lane, N, buf)     issue the tile's reads
//   finish(ctx, row_in_block, col0, lane, N, acc, buf, scratch)   compute and
//       store; `scratch` = 8 KiB of LDS private to the wave
// load(tile 0) is issued before the K loop, load(t+1) before finish(t): a
// store orders later loads of the same array behind it, so without this each
// tile would pay a full memory round trip.
template <class E, class = void>
struct epi_prefetch : std::false_type {};
template <class E>
struct epi_prefetch<E, std::void_t<decltype(E::kPrefetch)>> : std::true_type {};
template <class E, bool = epi_prefetch<E>::value>
struct epi_prefetch_floats { static constexpr int value = 1; };
template <class E>
struct epi_prefetch_floats<E, true> {
  static constexpr int value = E::kPrefetch;
};

// An element-wise epilogue that reads per-element state (the proximal step of
// the strided convolution: Y and the previous codes) may declare
//   struct Fetched; static constexpr bool kElemFetch = true;
//   Fetched fetch(row, col) const;      apply(row, col, acc, z, fetched)
// The kernels then issue every fetch of a thread BEFORE the K loop and apply
// after it: written as load-compute-store per element, each load waits for the
// stores before it (they may alias), 64 memory round trips per thread -- 50 of
// the 71 us of the analysis contraction at the reference's example geometry.
template <class E, class = void>
struct epi_elem_fetch : std::false_type {};
template <class E>
struct epi_elem_fetch<E, std::void_t<decltype(E::kElemFetch)>>
    : std::true_type {};
template <class E, bool = epi_elem_fetch<E>::value>
struct epi_fetched { struct type {}; };
template <class E>
struct epi_fetched<E, true> { typedef typename E::Fetched type; };

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 x3_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int x3_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kX3BM = 128, kX3BN = 128, kX3BK = 32;
constexpr int kX3TileBytes = 128 * 64;   // one operand part: 128 rows x 64 B

// C[M,N] = opA * opB over k in [z*k_chunk, min(K, (z+1)*k_chunk)), z=blockIdx.y
//   A_KC: A is stored [M][K] (k contiguous, leading dim lda); else [K][M].
//   B_KC: B is stored [N][K] (k contiguous, leading dim ldb); else [K][N].
struct GemmArgs {
  const float* A;
  const float* B;
  int64_t M, N, K;
  int64_t lda, ldb;
  int64_t k_chunk;   // multiple of kGemmBK
  int a_vec, b_vec;  // 16-byte vector loads allowed for this operand
  // batch (blockIdx.z): operand offsets in elements per batch entry; the
  // epilogue sees row + batch * M, i.e. the batches stacked along the rows
  int64_t a_batch, b_batch;
};

__device__ __forceinline__ float4 load4_guarded(const float* p, int64_t avail,
                                                bool line_ok, bool vec_ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!line_ok || avail <= 0) return v;
  if (vec_ok && avail >= 4) return *reinterpret_cast<const float4*>(p);
  v.x = p[0];
  if (avail > 1) v.y = p[1];
  if (avail > 2) v.z = p[2];
  if (avail > 3) v.w = p[3];
  return v;
}

// Stage one 128 x 16 operand tile (global -> registers).
//   K_CONTIG : element (line, k) at P[line*ld + k]
//   !K_CONTIG: element (line, k) at P[k*ld + line]
template <bool K_CONTIG>
__device__ __forceinline__ void stage_load(const float* P, int64_t ld,
                                           int64_t line0, int64_t lines,
                                           int64_t k0, int64_t k_end,
                                           bool vec_ok, int tid,
                                           float4 (&regs)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = tid + i * kGemmThreads;
    if (K_CONTIG) {
      const int line = f >> 2, kq = f & 3;
      const int64_t gl = line0 + line, gk = k0 + kq * 4;
      regs[i] = load4_guarded(P + gl * ld + gk, k_end - gk, gl < lines, vec_ok);
    } else {
      const int kr = f >> 5, lq = f & 31;
      const int64_t gk = k0 + kr, gl = line0 + lq * 4;
      regs[i] = load4_guarded(P + gk * ld + gl, lines - gl, gk < k_end, vec_ok);
    }
  }
}

template <bool K_CONTIG>
__device__ __forceinline__ void stage_store(float (*S)[kGemmPitch], int tid,
                                            const float4 (&regs)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = tid + i * kGemmThreads;
    if (K_CONTIG) {
      const int line = f >> 2, kq = f & 3;
      S[kq * 4 + 0][line] = regs[i].x;
      S[kq * 4 + 1][line] = regs[i].y;
      S[kq * 4 + 2][line] = regs[i].z;
      S[kq * 4 + 3][line] = regs[i].w;
    } else {
      const int kr = f >> 5, lq = f & 31;
      *reinterpret_cast<float4*>(&S[kr][lq * 4]) = regs[i];
    }
  }
}

template <bool A_KC, bool B_KC, class Epi>
__global__ __launch_bounds__(kGemmThreads) void gemm_f32_kernel(GemmArgs g,
                                                                Epi epi) {
  __shared__ __attribute__((aligned(16))) float As[2][kGemmBK][kGemmPitch];
  __shared__ __attribute__((aligned(16))) float Bs[2][kGemmBK][kGemmPitch];

  resolve_epilogue(epi, 0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t tiles_n = (g.N + kGemmBN - 1) / kGemmBN;
  const int64_t tile = blockIdx.x;
  const int64_t m0 = (tile / tiles_n) * kGemmBM;
  const int64_t n0 = (tile % tiles_n) * kGemmBN;
  const int z = blockIdx.y;
  const int64_t k_begin = (int64_t)z * g.k_chunk;
  const int64_t k_end = (k_begin + g.k_chunk < g.K) ? k_begin + g.k_chunk : g.K;
  const int nk = (int)((k_end - k_begin + kGemmBK - 1) / kGemmBK);
  g.A += (int64_t)blockIdx.z * g.a_batch;
  g.B += (int64_t)blockIdx.z * g.b_batch;
  const int64_t row_shift = (int64_t)blockIdx.z * g.M;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // pipelined whole-tile epilogue (see the traits above): the first tile's
  // reads are issued before the K loop
  constexpr bool kPipe = epi_prefetch<Epi>::value;
  float pre0[epi_prefetch_floats<Epi>::value],
      pre1[epi_prefetch_floats<Epi>::value];
  auto ctx = [&] {
    if constexpr (kPipe) return epi.begin(m0, g.M); else return 0;
  }();
  if constexpr (kPipe)
    epi.load(ctx, wm * 64, n0 + wn * 64, lane, g.N, pre0);

  constexpr bool kFetch = epi_elem_fetch<Epi>::value;
  typename epi_fetched<Epi>::type fetched[kFetch ? 64 : 1];
  if constexpr (kFetch) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int64_t col = n0 + (wave & 1) * 64 + (t & 1) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + (wave >> 1) * 64 + (t >> 1) * 32 + (r & 3) +
                            8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.M && col < g.N)
          fetched[16 * t + r] =
              epi.fetch(row + (int64_t)blockIdx.z * g.M, col);
      }
    }
  }

  float4 ra[2], rb[2];
  if (nk > 0) {
    stage_load<A_KC>(g.A, g.lda, m0, g.M, k_begin, k_end, g.a_vec, tid, ra);
    stage_load<B_KC>(g.B, g.ldb, n0, g.N, k_begin, k_end, g.b_vec, tid, rb);
    stage_store<A_KC>(As[0], tid, ra);
    stage_store<B_KC>(Bs[0], tid, rb);
  }
  __syncthreads();

  const int half = lane >> 5, l31 = lane & 31;
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) {
      const int64_t k0 = k_begin + (int64_t)(kt + 1) * kGemmBK;
      stage_load<A_KC>(g.A, g.lda, m0, g.M, k0, k_end, g.a_vec, tid, ra);
      stage_load<B_KC>(g.B, g.ldb, n0, g.N, k0, k_end, g.b_vec, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < kGemmBK / 2; ++kk) {
      const int k = kk * 2 + half;
      const float a0 = As[cur][k][wm * 64 + l31];
      const float a1 = As[cur][k][wm * 64 + 32 + l31];
      const float b0 = Bs[cur][k][wn * 64 + l31];
      const float b1 = Bs[cur][k][wn * 64 + 32 + l31];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) {
      stage_store<A_KC>(As[cur ^ 1], tid, ra);
      stage_store<B_KC>(Bs[cur ^ 1], tid, rb);
    }
    __syncthreads();
    cur ^= 1;
  }

  if constexpr (kPipe) {
    // the operand tiles are dead after the K loop (last barrier above): two
    // waves stage their accumulator tiles through As, two through Bs
    float* scratch = (wave < 2 ? &As[0][0][0] : &Bs[0][0][0]) +
                     (wave & 1) * 2048;
    const int r0 = wm * 64;
    const int64_t c0 = n0 + wn * 64;
    epi.load(ctx, r0, c0 + 32, lane, g.N, pre1);
    epi.finish(ctx, r0, c0, lane, g.N, acc[0][0], pre0, scratch);
    epi.load(ctx, r0 + 32, c0, lane, g.N, pre0);
    epi.finish(ctx, r0, c0 + 32, lane, g.N, acc[0][1], pre1, scratch);
    epi.load(ctx, r0 + 32, c0 + 32, lane, g.N, pre1);
    epi.finish(ctx, r0 + 32, c0, lane, g.N, acc[1][0], pre0, scratch);
    epi.finish(ctx, r0 + 32, c0 + 32, lane, g.N, acc[1][1], pre1, scratch);
    epi.block_end();
    return;
  }
  // C/D layout of the 32x32 MFMA: column = lane & 31,
  // row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int64_t col = n0 + wn * 64 + ni * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row =
            m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < g.M && col < g.N) {
          if constexpr (kFetch)
            epi.apply(row + row_shift, col, acc[mi][ni][r], z,
                      fetched[16 * (2 * mi + ni) + r]);
          else
            epi(row + row_shift, col, acc[mi][ni][r], z);
        }
      }
    }
  }
  epi.block_end();
}

static int gemm_compute_units() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    cus = (hipGetDevice(&dev) == hipSuccess &&
           hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount,
                                 dev) == hipSuccess && n > 0)
              ? n
              : 256;
  }
  return cus;
}

static bool gemm_prefers_small(int64_t M, int64_t N) {
  const int64_t tiles = ceil_div(M, kGemmBM) * ceil_div(N, kGemmBN);
  const double full_tiles = (double)M * (double)N / (kGemmBM * kGemmBN);
  return tiles <= 8 || full_tiles <= (double)gemm_compute_units();
}

static inline int gemm_vec_ok(const float* p, int64_t ld) {
  return ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld % 4) == 0) ? 1 : 0;
}

// k_slices > 1 => split-K, the epilogue receives the slice index.
// Small problems (a handful of 128x128 tiles: the Gram matrix and the residual
// of the reference's example sizes, 256 atoms x 250 patches) run latency-bound
// on the kernel above -- 4 blocks, each walking all of K behind its LDS staging.
// Here a block owns ONE 32x32 output tile, its four waves split K four ways,
// operands go from L2 straight into the MFMA operand registers (one float per
// lane and product), and the four partial tiles are summed through LDS in wave
// order (bitwise reproducible).  Element-wise epilogues only.
// B operand sources of the small kernel.  BPlain: g.B as GemmArgs declares it.
// A source with `row(col)` / `at(base, k)` maps element (col, k) of a k-major
// operand onto memory itself -- the im2col view of the strided convolution
// (conv_patch.h) reads the residual image in place, no patch matrix in HBM.
struct BPlain {};
template <class B>
struct b_is_plain : std::is_same<B, BPlain> {};

template <bool A_KC, bool B_KC, class Epi, class BSrc = BPlain>
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(GemmArgs g,
                                                             Epi epi,
                                                             BSrc bsrc = BSrc()) {
  __shared__ float part[3][16][64];
  resolve_epilogue(epi, 0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tiles_n = (g.N + 31) / 32;
  const int64_t m0 = (blockIdx.x / tiles_n) * 32, n0 = (blockIdx.x % tiles_n) * 32;
  const int64_t row = m0 + (lane & 31), col = n0 + (lane & 31);
  const int kk = lane >> 5;
  // batch (blockIdx.y): the epilogue sees the batches stacked along the rows
  g.A += (int64_t)blockIdx.y * g.a_batch;
  g.B += (int64_t)blockIdx.y * g.b_batch;
  const int64_t row_shift = (int64_t)blockIdx.y * g.M;
  // this wave's K range, a multiple of 16 long
  const int64_t quarter = ((g.K + 63) / 64) * 16;
  const int64_t k_begin = wave * quarter;
  const int64_t k_end = k_begin + quarter < g.K ? k_begin + quarter : g.K;
  const bool row_ok = row < g.M, col_ok = col < g.N;
  // (addresses are clamped into the operand and the value masked afterwards:
  // sixteen unconditional loads per step go out back to back, where a guarded
  // load sits in a branch of its own)
  const int64_t row_c = row_ok ? row : g.M - 1, col_c = col_ok ? col : g.N - 1;
  const float* pa = g.A + (A_KC ? row_c * g.lda : row_c);
  const float* pb = g.B + (B_KC ? col_c * g.ldb : col_c);
  if constexpr (!b_is_plain<BSrc>::value) pb = bsrc.row(col_c);
  const int64_t sa = A_KC ? 1 : g.lda, sb = B_KC ? 1 : g.ldb;
  constexpr bool kFetch = epi_elem_fetch<Epi>::value;
  typename epi_fetched<Epi>::type fetched[kFetch ? 16 : 1];
  if constexpr (kFetch) {
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (orow < g.M && col < g.N)
          fetched[r] = epi.fetch(orow + row_shift, col);
      }
    }
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // two steps of operands in flight: the loads of step i + 1 are issued
  // before the products of step i
  const int64_t k_last = k_end > k_begin ? k_end - 1 : k_begin;
  // k of (product j, lane half kk) = k0 + 8 kk + j: a lane of a k-contiguous
  // operand reads 8 consecutive floats -- two 16-byte loads instead of eight
  // dwords that each touch 32 cache lines per wave (the kernel was bound by
  // the address unit: 23 us for the 342 blocks of the convolutional example)
  auto issue_operand = [&](const float* p, int64_t stride, bool kc_vec,
                           int64_t k0, float (&v)[8]) {
    const int64_t kb = k0 + 8 * kk;
    if (kc_vec && kb + 8 <= k_end) {
      const float4 lo = *reinterpret_cast<const float4*>(p + kb);
      const float4 hi = *reinterpret_cast<const float4*>(p + kb + 4);
      v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
      v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t k = kb + j;
      v[j] = p[(k < k_end ? k : k_last) * stride];
    }
  };
  auto issue = [&](int64_t k0, float (&av)[8], float (&bv)[8]) {
    issue_operand(pa, sa, A_KC && g.a_vec && (k_begin % 4 == 0), k0, av);
    if constexpr (b_is_plain<BSrc>::value) {
      issue_operand(pb, sb, B_KC && g.b_vec && (k_begin % 4 == 0), k0, bv);
    } else {
      // mapped operand: 8 consecutive k of a lane are contiguous in memory
      // when the source says so (two 16-byte loads), else element by element
      const int64_t kb = k0 + 8 * kk;
      if (bsrc.vec && kb + 8 <= k_end) {
        const float* p8 = bsrc.at(pb, kb);
        const float4 lo = *reinterpret_cast<const float4*>(p8);
        const float4 hi = *reinterpret_cast<const float4*>(p8 + 4);
        bv[0] = lo.x; bv[1] = lo.y; bv[2] = lo.z; bv[3] = lo.w;
        bv[4] = hi.x; bv[5] = hi.y; bv[6] = hi.z; bv[7] = hi.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int64_t k = kb + j;
          bv[j] = *bsrc.at(pb, k < k_end ? k : k_last);
        }
      }
    }
  };
  auto products = [&](int64_t k0, const float (&av)[8], const float (&bv)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool live = k0 + 8 * kk + j < k_end;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(
          (live && row_ok) ? av[j] : 0.f, (live && col_ok) ? bv[j] : 0.f, acc,
          0, 0, 0);
    }
  };
  if (k_begin < k_end) {
    float a0[8], b0[8], a1[8], b1[8];
    issue(k_begin, a0, b0);
    for (int64_t k0 = k_begin; k0 < k_end; k0 += 32) {
      if (k0 + 16 < k_end) issue(k0 + 16, a1, b1);
      products(k0, a0, b0);
      if (k0 + 16 < k_end) {
        if (k0 + 32 < k_end) issue(k0 + 32, a0, b0);
        products(k0 + 16, a1, b1);
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = (acc[r] + part[0][r][lane]) +
                      (part[1][r][lane] + part[2][r][lane]);
      const int64_t orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (orow < g.M && col < g.N) {
        if constexpr (kFetch)
          epi.apply(orow + row_shift, col, v, 0, fetched[r]);
        else
          epi(orow + row_shift, col, v, 0);
      }
    }
  }
  epi.block_end();
}

template <bool A_KC, bool B_KC, class Epi>
static int launch_gemm_f32(const float* A, int64_t lda, const float* B,
                           int64_t ldb, int64_t M, int64_t N, int64_t K,
                           int k_slices, Epi epi, hipStream_t st,
                           int64_t batches = 1, int64_t a_batch = 0,
                           int64_t b_batch = 0) {
  if (M <= 0 || N <= 0 || batches <= 0) return VTC_OK;
  if (batches > 65535) {
    set_error("gemm: too many batch entries (%lld)", (long long)batches);
    return VTC_ERR_INVALID_ARGUMENT;
  }
  GemmArgs g;
  g.A = A;
  g.B = B;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = lda;
  g.ldb = ldb;
  if (k_slices < 1) k_slices = 1;
  int64_t chunk = ceil_div(ceil_div(K, k_slices), kGemmBK) * kGemmBK;
  if (chunk < kGemmBK) chunk = kGemmBK;
  g.k_chunk = chunk;
  g.a_vec = gemm_vec_ok(A, lda) && (a_batch % 4 == 0);
  g.b_vec = gemm_vec_ok(B, ldb) && (b_batch % 4 == 0);
  g.a_batch = a_batch;
  g.b_batch = b_batch;
  const int64_t tiles = ceil_div(M, kGemmBM) * ceil_div(N, kGemmBN);
  if constexpr (!epi_whole_tile<Epi>::value) {
    // Few big tiles (the analysis contraction of the reference's
    // convolutional example: 43 blocks on 256 CUs), or tiles that are mostly
    // empty (N = 64 or 144 output columns): the 32x32-tile kernel when the
    // output amounts to no more full 128x128 tiles than there are CUs.
    // Measured on f32 FISTA runs of mid-size shapes: 1.6-2.8x below a quarter
    // of the chip, 7-18 % up to one tile per CU, 8 % SLOWER at two full tiles
    // per CU.  (Batched products stay on the big kernel: 12.6 vs 20.4 us for
    // the synthesis of that geometry, 90 tiles.)
    if (gemm_prefers_small(M, N) && k_slices == 1 && batches == 1) {
      const int64_t small_tiles = ceil_div(M, 32) * ceil_div(N, 32);
      hipLaunchKernelGGL((gemm_f32_small_kernel<A_KC, B_KC, Epi>),
                         dim3((unsigned)small_tiles, (unsigned)batches),
                         dim3(256), 0, st, g, epi);
      VTC_LAUNCH_CHECK();
      return VTC_OK;
    }
  }
  if (tiles > 0x7fffffffLL) {
    set_error("gemm: too many tiles (%lld)", (long long)tiles);
    return VTC_ERR_INVALID_ARGUMENT;
  }
  dim3 grid((unsigned)tiles, (unsigned)k_slices, (unsigned)batches);
  hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, Epi>), grid,
                     dim3(kGemmThreads), 0, st, g, epi);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

// C[M,N] = A[M,K] * Bsrc(N,K)^T on the small kernel, the B operand read
// through `bsrc` (see BPlain).  Returns VTC_ERR_UNSUPPORTED when the problem
// is not one for the small kernel (the caller then materialises B).
template <class Epi, class BSrc>
static int launch_gemm_f32_small_mapped(const float* A, int64_t lda, int64_t M,
                                        int64_t N, int64_t K, Epi epi,
                                        BSrc bsrc, hipStream_t st) {
  if (M <= 0 || N <= 0) return VTC_OK;
  const int64_t small_tiles = ceil_div(M, 32) * ceil_div(N, 32);
  if (!gemm_prefers_small(M, N) || small_tiles > 0x7fffffffLL)
    return VTC_ERR_UNSUPPORTED;
  GemmArgs g;
  g.A = A;
  g.B = nullptr;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = lda;
  g.ldb = 0;
  g.k_chunk = K;
  g.a_vec = gemm_vec_ok(A, lda);
  g.b_vec = 0;
  g.a_batch = g.b_batch = 0;
  hipLaunchKernelGGL((gemm_f32_small_kernel<true, true, Epi, BSrc>),
                     dim3((unsigned)small_tiles), dim3(256), 0, st, g, epi,
                     bsrc);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

// Number of K slices actually covered by launch_gemm_f32 for (K, k_slices).
static inline int gemm_effective_slices(int64_t K, int k_slices) {
  if (k_slices < 1) k_slices = 1;
  int64_t chunk = ceil_div(ceil_div(K, k_slices), kGemmBK) * kGemmBK;
  if (chunk < kGemmBK) chunk = kGemmBK;
  return (int)ceil_div(K > 0 ? K : 1, chunk);
}

// ---- epilogues shared by several translation units -----------------------
struct EpiStore {  // C = acc
  float* C;
  int64_t ldc;
  __device__ __forceinline__ void operator()(int64_t row, int64_t col, float v,
                                             int) const {
    C[row * ldc + col] = v;
  }
  __device__ __forceinline__ void block_end() const {}
};

struct EpiSlab {  // split-K partial: slab z
  float* slabs;
  int64_t slab_stride, ldc;
  __device__ __forceinline__ void operator()(int64_t row, int64_t col, float v,
                                             int z) const {
    slabs[(int64_t)z * slab_stride + row * ldc + col] = v;
  }
  __device__ __forceinline__ void block_end() const {}
};

struct EpiMinus {  // C = acc - X   (the reconstruction residual)
  float* C;
  const float* X;
  int64_t ldc, ldx;
  __device__ __forceinline__ void operator()(int64_t row, int64_t col, float v,
                                             int) const {
    C[row * ldc + col] = sub_rn(v, X[row * ldx + col]);
  }
  __device__ __forceinline__ void block_end() const {}
};

// the same, leaving max |C| in a slot for the f16 split of the next product
// (x3_scale.h)
struct EpiMinusMax {
  float* C;
  const float* X;
  int64_t ldc, ldx;
  unsigned* max_out;
  float mx = 0.f;
  __device__ __forceinline__ void operator()(int64_t row, int64_t col, float v,
                                             int) {
    const float r = sub_rn(v, X[row * ldx + col]);
    C[row * ldc + col] = r;
    mx = fmaxf(mx, fabsf(r));
  }
  __device__ __forceinline__ void block_end() const {
    cx_publish_max_wave(mx, max_out);
  }
};

// out[i] = sum_z slabs[z][i], z ascending: a fixed order, so the result does
// not depend on scheduling (bitwise reproducible run to run).
int launch_slab_reduce(const float* slabs, int slices, int64_t count,
                       float* out, hipStream_t st);

}  // namespace vtc
