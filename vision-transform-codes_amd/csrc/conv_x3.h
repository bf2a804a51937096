// Unit-stride, single-channel convolutional ISTA/FISTA on the bf16 matrix pipe
// with split operands ("bf16x3": hi = bf16(x), lo = bf16(x - hi), product
// hi*hi + hi*lo + lo*hi accumulated in f32 -- float32-level accuracy, see
// gemm_x3.h).  BASELINE configs[4]: 128 kernels of 11x11 on 256x256 images.
//
// Restates analysis_transforms/convolutional/ista_fista.py:152-155 as two
// contractions per iteration, neither with an im2col buffer in HBM:
//
//   synthesis  Q[tap, (u,v)] = sum_s D[s,tap] * Y[s,u,v]        M=taps N=pos K=s
//              recon[y,x]    = sum_tap Q[tap, (y-dy, x-dx)]     ("col2im")
//   analysis   G[s, (u,v)]   = sum_(dy,dx) D[s,dy,dx] * r[u+dy, v+dx]
//                                                     M=atoms N=pos K=(dy,16 dx)
//
// In both, the accumulator tile's lanes run along v (code columns), so code
// rows are read and written as 128-byte segments.
//
// Synthesis: a block owns a TH x TW tile of the residual image.  Each of its
// 8 waves takes code rows u = y0-(K-1)+wave, +8, ... (2 to 8 of them, chosen
// per launch so that the blocks fill whole rounds of CUs), forms Q for 64 code
// columns and adds it into ITS OWN copy of the tile in LDS.  The kernel taps
// are ordered over the accumulator rows so that the two half-waves of one LDS
// access never touch the same pixel; a wave's updates are then ordered by
// program order, the 8 copies are summed in wave order, and the result is
// bitwise reproducible.  The Y operand goes from HBM straight into MFMA
// operand registers (a lane holds 8 atoms of one code position).
//
// Analysis: the residual window of the tile is split once into bf16 hi/lo
// planes in LDS; a lane's operand is 8 consecutive pixels of a window row (an
// unaligned 16-byte LDS read).  The gradient step, threshold and FISTA
// extrapolation run on the accumulator tile: Y and the codes are read and
// written once per iteration.
#pragma once

#include "x3_scale.h"

namespace vtc {

typedef __bf16 cx_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int cx_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kCxStrip = 64;       // code columns per wave unit (2 MFMA tiles)
constexpr int kCxSynWaves = 8;
constexpr int kCxSynMaxRows = 8;   // code rows per wave (synthesis): 2 .. 8,
                                   // chosen per launch (cx_pick_rows)
constexpr int kCxAnaMaxRows = 8;   // code rows per block (analysis): 4 or 8,
                                   // chosen per launch (cx_fill_plan)
constexpr int kCxAnaPitch = 88;    // window row pitch, bf16 elements

template <int K>
struct CxDims {
  static constexpr int TAPS = K * K;
  static constexpr int HROWS = (K + 1) / 2;         // tap rows per half-wave
  static constexpr int HTAPS = HROWS * K;
  static constexpr int MT = (HTAPS + 15) / 16;      // 32-row operand tiles
  static constexpr int SLOTS = 32 * MT;
  static constexpr int TW = kCxStrip - (K - 1);
  static constexpr int PW = TW + 1;                 // private tile pitch
  static constexpr int NI = (MT <= 4) ? 2 : 1;      // column tiles per pass
};

// Accumulator row -> kernel tap.  In the 32x32 accumulator layout register r
// of lanes 0-31 is row rho = (r & 3) + 8 (r >> 2), of lanes 32-63 row rho + 4.
// Rows with bit 2 clear take taps idx = 0 .. HTAPS-1 (tap rows 0 .. HROWS-1,
// row-major), the row 4 further on takes the tap HROWS tap-rows below it:
//   m = 32 mt + rho + 4 h   ->   idx = 16 mt + r,   tap = idx + h * HTAPS
// so the two half-waves of one accumulator register always sit on different
// image rows (no two lanes of a ds_add share a pixel) and their pixel offsets
// differ by the constant HROWS * pitch, which goes into the lane's base.
// Returns -1 for a row without a tap (its operand row is zero).
__host__ __device__ inline int cx_slot_tap(int m, int k) {
  const int hrows = (k + 1) / 2, htaps = hrows * k;
  const int idx = ((m >> 3) << 2) | (m & 3);
  if (idx >= htaps) return -1;
  const int t = idx + ((m >> 2) & 1) * htaps;
  return t < k * k ? t : -1;
}

__device__ __forceinline__ uint16_t cx_bits(__bf16 v) {
  return __builtin_bit_cast(uint16_t, v);
}

// ------------------------------------------------ operand types and scales
// Two 16-bit operand types behind one template flag.  F16 = false: bf16 hi/lo
// (8 + 8 significand bits, 2^-17 per product: 1.6e-5 from the reference after
// 200 iterations).  F16 = true: f16 hi/lo (11 + 11 bits, 2^-22 per product,
// the float32 noise floor) at the same MFMA rate and the same bytes -- f16 has
// 5 exponent bits, so every operand is brought to the range [16, 32) by a
// power of two first (exact, and undone exactly on the f32 accumulators):
//   * the kernels: one sigma_D per call (dictionary_scale_kernel);
//   * the residual: one sigma_R per launch, from max |R| which the kernel that
//     WROTE the residual left in device memory (CxScales);
//   * the momentum iterate Y as the synthesis operand: in the fused kernel one
//     sigma per code column, taken from the values themselves (the column is
//     the N index of the product, so its scale factors out of the sum over
//     atoms); in the stand-alone synthesis kernel one sigma_Y per launch, from
//     max |Y| left by the kernel that wrote Y.
// An entry 2^12 below the maximum of its operand still has all 22 bits; below
// that the lo part goes subnormal and the absolute error stays at 2^-29 of the
// maximum.
typedef _Float16 cx_f16x8 __attribute__((ext_vector_type(8)));

template <bool F16>
__device__ __forceinline__ f32x16 cx_mfma(const uint4& a, const uint4& b,
                                          const f32x16& c) {
  if (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(
        __builtin_bit_cast(cx_f16x8, a), __builtin_bit_cast(cx_f16x8, b), c, 0,
        0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(
      __builtin_bit_cast(cx_bf16x8, a), __builtin_bit_cast(cx_bf16x8, b), c, 0,
      0, 0);
}

// one value -> its 16-bit hi and lo parts (bit patterns)
template <bool F16>
__device__ __forceinline__ void cx_split1(float v, uint16_t& hi, uint16_t& lo) {
  if (F16) {
    const _Float16 h = (_Float16)v;
    hi = __builtin_bit_cast(uint16_t, h);
    lo = __builtin_bit_cast(uint16_t, (_Float16)(v - (float)h));
  } else {
    const __bf16 h = (__bf16)v;
    hi = cx_bits(h);
    lo = cx_bits((__bf16)(v - (float)h));
  }
}

// ------------------------------------------------------------------ pack
// syn image (uint16): [channel][atom chunk][plane][slot][s16 + 8]
//                                    D[chunk * s16 + a][channel][tap(slot)], k = a
// ana image (uint16): [chunk][channel][plane][dy][2][AC][8]
//                                          D[chunk*AC + a][channel][dy][8 h + j]
// (blockIdx.y = image channel: the dictionary is (s, c, k, k) row-major)
template <bool F16>
__global__ void conv_x3_pack_kernel(const float* __restrict__ D,
                                    uint16_t* __restrict__ syn,
                                    uint16_t* __restrict__ ana, int s, int k,
                                    int s16, int syn_chunks, int slots, int AC,
                                    int chunks,
                                    const float* __restrict__ dscale) {
  const float sigma = F16 ? dscale[0] : 1.f;
  const int taps = k * k;
  const int pitch = s16 + 8;                       // s16: atoms per syn chunk
  const int64_t syn_plane = (int64_t)slots * pitch;
  const int64_t syn_all = (int64_t)syn_chunks * syn_plane;
  const int64_t ana_plane = (int64_t)k * AC * 16;
  const int64_t total = syn_all + (int64_t)chunks * ana_plane;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int channel = blockIdx.y, channels = gridDim.y;
  D += (int64_t)channel * taps;
  syn += (int64_t)channel * 2 * syn_all;
  const int64_t kernel_elems = (int64_t)channels * taps;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += stride) {
    float v = 0.f;
    uint16_t *hi, *lo;
    if (e < syn_all) {
      const int q = (int)(e / syn_plane);
      const int within = (int)(e % syn_plane);
      const int slot = within / pitch, a = within % pitch;
      const int t = cx_slot_tap(slot, k);
      if (a < s16 && q * s16 + a < s && t >= 0)
        v = D[(int64_t)(q * s16 + a) * kernel_elems + t];
      hi = syn + (int64_t)q * 2 * syn_plane + within;
      lo = hi + syn_plane;
    } else {
      const int64_t f = e - syn_all;
      const int chunk = (int)(f / ana_plane);
      const int rem = (int)(f % ana_plane);
      // [dy][half][atom][8]: the two half-waves of an operand read (taps
      // 0..7 / 8..15 of a tap row) each take 16 contiguous bytes per lane
      // (atoms 32 bytes apart cost 1.75x the LDS cycles: r02_lds_unaligned.txt)
      const int dy = rem / (AC * 16), within = rem % (AC * 16);
      const int a = chunk * AC + (within / 8) % AC,
                dx = 8 * (within / (AC * 8)) + within % 8;
      if (a < s && dx < k) v = D[(int64_t)a * kernel_elems + dy * k + dx];
      hi = ana + ((int64_t)chunk * channels + channel) * 2 * ana_plane + rem;
      lo = hi + ana_plane;
    }
    cx_split1<F16>(v * sigma, *hi, *lo);
  }
}

// registers (times a power-of-two scale) -> one MFMA operand pair
template <bool F16>
__device__ __forceinline__ void cx_split8(const float (&v)[8], float scale,
                                          uint4& hi, uint4& lo) {
  if (F16) {
    // two values per conversion (v_cvt_pk_f16_f32, round to nearest even as the
    // scalar one): the same arithmetic in fewer VALU instructions
    typedef float pair_f32 __attribute__((ext_vector_type(2)));
    typedef _Float16 pair_f16 __attribute__((ext_vector_type(2)));
    unsigned hw[4], lw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const pair_f32 x = {v[2 * j] * scale, v[2 * j + 1] * scale};
      const pair_f16 h = __builtin_convertvector(x, pair_f16);
      const pair_f16 l = __builtin_convertvector(
          x - __builtin_convertvector(h, pair_f32), pair_f16);
      hw[j] = __builtin_bit_cast(unsigned, h);
      lw[j] = __builtin_bit_cast(unsigned, l);
    }
    hi = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    lo = make_uint4(lw[0], lw[1], lw[2], lw[3]);
  } else {
    cx_bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      h[j] = (__bf16)v[j];
      l[j] = (__bf16)(v[j] - (float)h[j]);
    }
    hi = __builtin_bit_cast(uint4, h);
    lo = __builtin_bit_cast(uint4, l);
  }
}

// ------------------------------------------------------------- synthesis
template <int K, bool F16>
__global__ __launch_bounds__(512) void conv_synth_x3_kernel(
    const float* __restrict__ Y, const uint16_t* __restrict__ syn_image,
    const float* __restrict__ X, float* __restrict__ R, ConvGeo g, int s16,
    int syn_chunks, int tiles_x, int tiles_y, int rows_per_wave, CxScales sc) {
  using Dm = CxDims<K>;
  constexpr int MT = Dm::MT, NI = Dm::NI, TW = Dm::TW, PW = Dm::PW;
  // tile height: the waves' code rows y0-(K-1) .. y0-(K-1)+8*rows-1 reach the
  // image rows y0 .. y0+TH-1 completely
  const int TH = kCxSynWaves * rows_per_wave - (K - 1);
  extern __shared__ __attribute__((aligned(16))) char cx_lds[];
  char* lds = cx_lds;
  const int pitch = s16 + 8;                       // elements per slot row
  const int plane = Dm::SLOTS * pitch;             // elements per plane
  const uint16_t* Dh = reinterpret_cast<const uint16_t*>(lds);
  const uint16_t* Dl = Dh + plane;
  float* priv = reinterpret_cast<float*>(lds + (size_t)plane * 4);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  // Block -> tile, XCD aware (workgroups are dealt round-robin to the 8 XCDs,
  // one L2 each): vertically adjacent tiles share K-1 code rows of all atoms,
  // so a strip (image, tile column) stays on one XCD and its tiles run back to
  // back: block L is on XCD L % 8, slots L / 8 walk the tiles of strip
  // (slot / tiles_y) * 8 + XCD.  (Measured before: 895 MB of HBM-side reads
  // per launch for 290 MB of code maps.)
  // F16: Y enters as sigma_Y Y, the kernels as sigma_D D; the reconstruction
  // comes back by 1 / (sigma_Y sigma_D) before the image is subtracted
  if (F16) cx_clear_words(sc.y_zero, sc.images);
  // blockIdx.y = image channel: its own kernel planes, its own plane of the
  // images; the code maps are shared
  const int channel = blockIdx.y;
  syn_image += (int64_t)channel * syn_chunks * 2 * plane;
  const int64_t strips = g.b * tiles_x;
  const int64_t slot = (int64_t)blockIdx.x >> 3;
  const int64_t strip = (slot / tiles_y) * 8 + (blockIdx.x & 7);
  if (strip >= strips) return;                     // whole block (grid padding)
  const int tile_y = (int)(slot % tiles_y);
  const int tile_x = (int)(strip % tiles_x);
  const int64_t img = strip / tiles_x;
  float y_scale = 1.f, unscale = 1.f;
  if (F16) {
    float inv_y;
    cx_scale_of_bits(sc.y_in ? sc.y_in[img] : 0u, &y_scale, &inv_y);
    unscale = inv_y * sc.dscale[1];
  }
  for (int i = tid; i < kCxSynWaves * TH * PW; i += 512) priv[i] = 0.f;
  const int x0 = tile_x * TW, y0 = tile_y * TH;
  float* mine = priv + wave * TH * PW;
  const int64_t map = (int64_t)g.ch * g.cw;
  const float* Yimg = Y + img * g.s * map;
  const int nks = s16 / 16;

  // Y of this image as a buffer resource: a lane's operand is 8 atoms of one
  // code position, i.e. 8 dword loads `map` elements apart; a column outside
  // the code map gets an out-of-range offset, which reads as zero.
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)Yimg, 0, (int)((int64_t)g.s * map * 4), 0x00020000);
  const unsigned map4 = (unsigned)(map * 4);

  // The work of a wave is a flat sequence of batches q = ((row, column pass),
  // 2 or 4 K steps of 16 atoms); the operand loads of batch q+1 are in flight while
  // batch q runs on the matrix pipe (two register buffers).
  constexpr int kBatch = (MT * NI > 6) ? 2 : 4;    // K steps per batch
  constexpr int PASSES = 2 / NI;                   // column passes per row
  const int nbat = (nks + kBatch - 1) / kBatch;
  const int total = rows_per_wave * PASSES * nbat;

  // Odd tiles walk their code rows bottom-up: the K-1 rows a tile shares
  // with the tile below / above are then read by both at about the same time
  // (the end of an even tile and of the odd tile under it, the start of an
  // odd tile and of the even tile under it) and the second read hits the L2
  // -- they are on one XCD, see the block -> tile map above.
  auto row_of = [&](int q) {
    const int step = q / (PASSES * nbat);
    const int ur = (tile_y & 1) ? rows_per_wave - 1 - step : step;
    const int w = (tile_y & 1) ? kCxSynWaves - 1 - wave : wave;
    return y0 - (K - 1) + w + kCxSynWaves * ur;
  };
  auto col0_of = [&](int q) {
    return x0 - (K - 1) + 32 * NI * ((q / nbat) % PASSES);
  };
  // Kernel sets whose planes would leave no room for a sensible tile (128
  // kernels of 16x16: 139 KB) go through the LDS in `syn_chunks` chunks of s16
  // atoms; the private tiles accumulate over the chunks.
  int atom0 = 0;                                   // first atom of the chunk
  auto issue = [&](int q, float (&dst)[NI][kBatch][8]) {
    const int u = row_of(q);
    if (u < 0 || u >= g.ch) return;                // whole wave
    const int bt = q % nbat;
    const bool ragged = atom0 + s16 > g.s;         // some K step lacks atoms
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int v = col0_of(q) + 32 * ni + l31;
      const bool vok = v >= 0 && v < g.cw;
      const unsigned voff = vok ? (unsigned)(8 * half) * map4 +
                                      (unsigned)(u * g.cw + v) * 4u
                                : 0x80000000u;
#pragma unroll
      for (int kk = 0; kk < kBatch; ++kk) {
        const int ks = bt * kBatch + kk;
        if (ks >= nks) break;
        const bool guard = ragged && atom0 + ks * 16 + 16 > g.s;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          unsigned vo = voff;
          if (guard && atom0 + ks * 16 + 8 * half + j >= g.s)
            vo = 0x80000000u;
          dst[ni][kk][j] = __builtin_bit_cast(
              float, __builtin_amdgcn_raw_buffer_load_b32(
                         yrs, vo, (unsigned)(atom0 + ks * 16 + j) * map4, 0));
        }
      }
    }
  };

  f32x16 acc[MT][NI];
  auto compute = [&](int q, const float (&src)[NI][kBatch][8]) {
    const int u = row_of(q);
    if (u < 0 || u >= g.ch) return;                // whole wave
    const int bt = q % nbat;
    if (bt == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][ni][r] = 0.f;
    }
#pragma unroll
    for (int kk = 0; kk < kBatch; ++kk) {
      const int ks = bt * kBatch + kk;
      if (ks >= nks) break;
      uint4 bh[NI], bl[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        cx_split8<F16>(src[ni][kk], y_scale, bh[ni], bl[ni]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int off = (32 * mt + l31) * pitch + ks * 16 + 8 * half;
        const uint4 ah = *reinterpret_cast<const uint4*>(Dh + off);
        const uint4 al = *reinterpret_cast<const uint4*>(Dl + off);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mt][ni] = cx_mfma<F16>(ah, bh[ni], acc[mt][ni]);
          acc[mt][ni] = cx_mfma<F16>(ah, bl[ni], acc[mt][ni]);
          acc[mt][ni] = cx_mfma<F16>(al, bh[ni], acc[mt][ni]);
        }
      }
    }
    if (bt != nbat - 1) return;
    // col2im into this wave's copy of the tile.  Register (mt, r) of a lane
    // holds tap (dy, dx) at code column v: it belongs to pixel column v + dx.
    // The dx sum is formed in registers first: the register is rotated by dx
    // lanes inside its 32-lane half (ds_bpermute), lanes >= dx keep it for
    // their own column, lanes < dx for the column 32 further right.  That
    // leaves 2 * HROWS LDS updates per pass instead of one per tap (measured:
    // one ds_add_f32 per tap cost 100 cycles each and 63% of the kernel).
    // Rows without a tap hold exact zeros (zero operand rows), so only the
    // pixel range is tested.
    const int pyb = u - y0 + half * Dm::HROWS;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int pxb = col0_of(q) + 32 * ni + l31 - x0;
      float* base = mine + pyb * PW + pxb;
#pragma unroll
      for (int dy = 0; dy < Dm::HROWS; ++dy) {
        // plain read-modify-write: within one pass no two lanes of the wave
        // touch the same pixel (the half-waves sit on different image rows,
        // main and spill columns are disjoint), and a lane's spill column of
        // one column tile is the same lane's main column of the next, so
        // program order per lane is all the ordering needed.  (Not volatile:
        // a volatile access loses the LDS address space, becomes a flat
        // load / store and drains vmcnt -- every outstanding code-map access.)
        const bool row_ok = (unsigned)(pyb + dy) < (unsigned)TH;
        const bool main_ok = row_ok && (unsigned)pxb < (unsigned)TW;
        const bool spill_ok =
            row_ok && l31 < K - 1 && (unsigned)(pxb + 32) < (unsigned)TW;
        float* pm = base + dy * PW;
        float* ps = base + dy * PW + 32;
        float old_main = 0.f, old_spill = 0.f;
        if (main_ok) old_main = *pm;
        if (spill_ok) old_spill = *ps;
        float main_sum = 0.f, spill_sum = 0.f;
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
          const int idx = dy * K + dx;                 // compile time
          float rot = acc[idx / 16][ni][idx % 16];
          if (dx != 0) {
            const int from = ((l31 - dx) & 31) | (lane & 32);
            rot = __builtin_bit_cast(
                float, __builtin_amdgcn_ds_bpermute(
                           from * 4, __builtin_bit_cast(int, rot)));
          }
          if (l31 >= dx)
            main_sum = add_rn(main_sum, rot);
          else
            spill_sum = add_rn(spill_sum, rot);
        }
        if (main_ok) *pm = add_rn(old_main, main_sum);
        if (spill_ok) *ps = add_rn(old_spill, spill_sum);
        // the half-waves of LATER passes do touch these pixels: keep the
        // compiler from moving their reads above these writes (the LDS
        // itself serves one wave's accesses in order)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  float buf0[NI][kBatch][8], buf1[NI][kBatch][8];
  for (int chunk = 0; chunk < syn_chunks; ++chunk) {
    atom0 = chunk * s16;
    if (chunk != 0) __syncthreads();               // planes of the last chunk
    {
      const int n16 = plane / 4;                   // 2 planes * 2 B / 16 B
      const uint4* src = reinterpret_cast<const uint4*>(syn_image) +
                         (int64_t)chunk * n16;
      uint4* dst = reinterpret_cast<uint4*>(lds);
      for (int i = tid; i < n16; i += 512) dst[i] = src[i];
    }
    __syncthreads();
    issue(0, buf0);
    for (int q = 0; q < total; q += 2) {
      if (q + 1 < total) issue(q + 1, buf1);
      compute(q, buf0);
      if (q + 2 < total) issue(q + 2, buf0);
      if (q + 1 < total) compute(q + 1, buf1);
    }
  }
  __syncthreads();
  float r_max = 0.f;
  for (int e = tid; e < TH * TW; e += 512) {
    const int py = e / TW, px = e % TW;
    const int y = y0 + py, x = x0 + px;
    if (y >= g.H || x >= g.W) continue;
    float sum = priv[py * PW + px];
#pragma unroll
    for (int w = 1; w < kCxSynWaves; ++w)
      sum = add_rn(sum, priv[w * TH * PW + py * PW + px]);
    if (F16) sum *= unscale;
    const int64_t i = ((img * g.c + channel) * g.H + y) * (int64_t)g.W + x;
    const float rv = mul_rn(mask_at(g, y, x), sub_rn(sum, X[i]));
    R[i] = rv;
    r_max = fmaxf(r_max, fabsf(rv));
  }
  if (F16 && sc.r_out) {
    __syncthreads();                               // priv is free now
    cx_publish_max_word(r_max, sc.r_out + img,
                        reinterpret_cast<unsigned*>(priv));
  }
}

// -------------------------------------------------------------- analysis
struct __attribute__((packed, aligned(2))) CxUnaligned16 {
  uint32_t x, y, z, w;
};

// MA: 32-atom tiles per block (atom chunk AC = 32 * MA).  FAST: the common
// case (FISTA, soft threshold, no early stopping) with every option folded at
// compile time; the other instantiation reads them from ProxParams.
// SHIFT = 4 or 2: the window is kept in that many copies, copy c shifted by c
// pixels, and a lane reads its 8 pixels from copy (start mod SHIFT) at an
// 8-byte (4 copies) or 4-byte (2 copies) ALIGNED position -- the unaligned
// 16-byte read costs 52-64 LDS cycles per wave against 4
// (tools/micro/lds_unaligned.hip), and with one atom tile per block (colour
// images: 32 atoms, 6 MFMAs per tap row) four of them per tap row made the
// kernel LDS-bound five times over.  SHIFT = 1: one copy, unaligned reads.
// Chosen by the plan: as many copies as fit without costing a resident block.
template <int MA, bool FAST, bool F16, int SHIFT>
__global__ __launch_bounds__(256) void conv_analysis_x3_kernel(
    const float* __restrict__ R, const uint16_t* __restrict__ ana_image,
    float* __restrict__ Y, float* __restrict__ C, ConvGeo g, int tiles_v,
    int tiles_u, int chunks, int ana_rows, ProxParams pp, CxScales sc) {
  constexpr int AC = 32 * MA;
  extern __shared__ __attribute__((aligned(16))) char cx_lds[];
  char* lds = cx_lds;
  const int k = g.kh;
  const int plane = k * AC * 16;                   // elements
  const int rows = ana_rows + k - 1;
  const int win = rows * kCxAnaPitch;              // elements per window plane
  // per image channel: kernel planes [hi | lo], then window planes [hi | lo]
  // (SHIFT: 4 copies of [hi | lo | 32 elements of padding], which also puts
  // the copies on disjoint banks)
  const int wcopy = SHIFT > 1 ? 2 * win + 32 : 2 * win;
  const int wchan = SHIFT * wcopy;                 // elements per channel
  uint16_t* Dh = reinterpret_cast<uint16_t*>(lds);
  uint16_t* Dl = Dh + plane;
  uint16_t* Rh = Dh + 2 * plane * g.c;
  uint16_t* Rl = Rh + win;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  // Block -> tile, XCD aware.  Workgroups are dealt round-robin to the 8 XCDs
  // (one L2 each); the tiles_v tiles of one band (8 code rows of one atom
  // chunk) share their boundary cache lines, because code rows are not
  // 128-byte aligned, so a band stays on one XCD: block L runs on XCD L % 8,
  // and slots L / 8 walk the tiles of band (slot / tiles_v) * 8 + XCD.
  // F16: the window enters as sigma_R R, the kernels as sigma_D D; the
  // gradient step takes eta / (sigma_R sigma_D) -- a power-of-two factor
  // commutes with the rounding of eta * G
  if (F16) cx_clear_words(sc.r_zero, sc.images);
  float y_max = 0.f;                               // of the next iterate
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile_v = slot % tiles_v;
  const int64_t band = (int64_t)(slot / tiles_v) * 8 + xcd;
  if (band >= (int64_t)tiles_u * chunks * g.b) return;   // whole block
  const int chunk = (int)(band % chunks);
  const int tile_u = (int)((band / chunks) % tiles_u);
  const int64_t img = band / ((int64_t)chunks * tiles_u);
  float r_scale = 1.f, eta = pp.eta;
  if (F16) {
    float inv_r;
    cx_scale_of_bits(sc.r_in[img], &r_scale, &inv_r);
    eta = pp.eta * (inv_r * sc.dscale[1]);
  }
  const int u0 = tile_u * ana_rows, v0 = tile_v * kCxStrip;
  {
    const uint4* src = reinterpret_cast<const uint4*>(
        ana_image + (int64_t)chunk * g.c * 2 * plane);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    const int n16 = g.c * (plane / 4);
    for (int i = tid; i < n16; i += 256) dst[i] = src[i];
    for (int channel = 0; channel < g.c; ++channel) {
      const float* Rimg = R + (img * g.c + channel) * g.H * (int64_t)g.W;
      uint16_t* wh = Rh + channel * wchan;
      for (int e = tid; e < win; e += 256) {
        const int ry = e / kCxAnaPitch, rx = e % kCxAnaPitch;
        const int y = u0 + ry, x = v0 + rx;
        const float v =
            (y < g.H && x < g.W) ? Rimg[(int64_t)y * g.W + x] : 0.f;
        uint16_t hb, lb;
        cx_split1<F16>(F16 ? v * r_scale : v, hb, lb);
        if (SHIFT > 1) {
          // pixel x of a row sits at position x - c of copy c; the first c
          // pixels of the window land in the padding of the copy before
          // (never read: reads end at position 78 of a row of 88)
#pragma unroll
          for (int c = 0; c < SHIFT; ++c) {
            wh[c * wcopy + e - c] = hb;
            wh[c * wcopy + win + e - c] = lb;
          }
        } else {
          wh[e] = hb;
          wh[win + e] = lb;
        }
      }
    }
  }
  __syncthreads();
  const int64_t map = (int64_t)g.ch * g.cw;
  const unsigned map4 = (unsigned)(map * 4);
  const int code_bytes = (int)((int64_t)g.s * map * 4);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(Y + img * g.s * map), 0, code_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(C + img * g.s * map), 0, code_bytes, 0x00020000);
  // out of place (ProxParams): never the lines that are read
  const __amdgpu_buffer_rsrc_t yws = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(pp.y_out + img * g.s * map), 0, code_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t cws = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(pp.c_out + img * g.s * map), 0, code_bytes, 0x00020000);
  const bool ragged = (g.s % AC) != 0;
  double local = 0.0;
  for (int pass = 0; pass < ana_rows / 4; ++pass) {
    const int lu = wave + 4 * pass;
    const int u = u0 + lu;
    if (u >= g.ch) continue;                       // whole wave
    const bool fista = FAST ? true : (pp.fista != 0);
    const bool early = FAST ? false : (pp.delta_sum != nullptr);
    // Accumulator tile t = (atom tile t >> 1, column tile t & 1): lane l31 is
    // the code column, register r the atom (r & 3) + 8 (r >> 2) + 4 half, so
    // a wave access is two 128-byte row segments.  (Measured alternative:
    // atoms on the lanes and 8- or 16-byte accesses along the row -- a quarter
    // of the instructions but 32 cache lines per instruction -- is 45%
    // slower; staging the tile through LDS for 16-byte accesses along the
    // row, which pays off in the subspace epilogue where rows are 128-byte
    // aligned, measures no gain here.)  Buffer addressing: lane offset + a
    // scalar offset per register;
    // positions outside the code map get an out-of-range offset (loads give
    // 0, stores are dropped).
    auto tile_offset = [&](int t) -> unsigned {
      const int v = v0 + 32 * (t & 1) + l31;
      const int a0 = chunk * AC + 32 * (t >> 1) + 4 * half;
      return v < g.cw ? (unsigned)a0 * map4 + (unsigned)(u * g.cw + v) * 4u
                      : 0x80000000u;
    };
    auto atoms_left = [&](int t) -> int {          // rows rr < this exist
      return ragged ? g.s - (chunk * AC + 32 * (t >> 1) + 4 * half) : 64;
    };
    auto load_tile = [&](int t, float (&yv)[16], float (&cv)[16]) {
      const unsigned lane_off = tile_offset(t);
      const int left = atoms_left(t);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2);
        const unsigned vo = rr < left ? lane_off : 0x80000000u;
        yv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
            yrs, vo, (unsigned)rr * map4, 0));
        if (fista)
          cv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
              crs, vo, (unsigned)rr * map4, 0));
      }
    };
    auto finish_tile = [&](int t, const float (&yv)[16],
                           const float (&cv)[16], const f32x16& tile) {
      const unsigned lane_off = tile_offset(t);
      const int left = atoms_left(t);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2);
        const unsigned vo = rr < left ? lane_off : 0x80000000u;
        const float p = sub_rn(yv[r], mul_rn(eta, tile[r]));
        const float c = FAST ? shrink(p, pp.cutoff, VTC_SOFT)
                             : shrink(p, pp.cutoff, pp.mode);
        float d;
        if (fista) {
          d = sub_rn(c, cv[r]);
          const float y1 = add_rn(c, mul_rn(pp.beta, d));
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y1), yws, vo,
                                                (unsigned)rr * map4, 0);
          if (F16) y_max = fmaxf(y_max, fabsf(y1));
        } else {
          d = sub_rn(c, yv[r]);
          if (F16) y_max = fmaxf(y_max, fabsf(c));
        }
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(c), cws, vo,
                                              (unsigned)rr * map4, 0);
        if (early && vo != 0x80000000u) local += (double)(fabsf(d) / pp.eta);
      }
    };
    float yA[16], cA[16], yB[16], cB[16];
    load_tile(0, yA, cA);
    f32x16 acc[MA][2];
#pragma unroll
    for (int ma = 0; ma < MA; ++ma)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ma][ni][r] = 0.f;
    // K index = (image channel, dy, 16 dx)
    for (int cdy = 0; cdy < g.c * k; ++cdy) {
      const int channel = cdy / k, dy = cdy - channel * k;
      const int wbase = channel * wchan, dbase = channel * 2 * plane;
      uint4 bh[2], bl[2];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (SHIFT == 4) {
          // start = 32 ni + l31 + 8 half: copy (start & 3) = (l31 & 3), at
          // position start - (start & 3): 8-byte aligned
          const uint16_t* bp = Rh + wbase + (l31 & 3) * wcopy +
                               (lu + dy) * kCxAnaPitch + 32 * ni + (l31 & ~3) +
                               8 * half;
          const uint2 h0 = *reinterpret_cast<const uint2*>(bp);
          const uint2 h1 = *reinterpret_cast<const uint2*>(bp + 4);
          const uint2 l0 = *reinterpret_cast<const uint2*>(bp + win);
          const uint2 l1 = *reinterpret_cast<const uint2*>(bp + win + 4);
          bh[ni] = make_uint4(h0.x, h0.y, h1.x, h1.y);
          bl[ni] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        } else if (SHIFT == 2) {
          // copy (l31 & 1), position start - (start & 1): 4-byte aligned, four
          // dword reads per plane (ds_read2_b32 pairs)
          const uint32_t* bp = reinterpret_cast<const uint32_t*>(
              Rh + wbase + (l31 & 1) * wcopy + (lu + dy) * kCxAnaPitch +
              32 * ni + (l31 & ~1) + 8 * half);
          const uint32_t* lp = bp + win / 2;
          bh[ni] = make_uint4(bp[0], bp[1], bp[2], bp[3]);
          bl[ni] = make_uint4(lp[0], lp[1], lp[2], lp[3]);
        } else {
          const int off =
              wbase + (lu + dy) * kCxAnaPitch + 32 * ni + l31 + 8 * half;
          const CxUnaligned16 h =
              *reinterpret_cast<const CxUnaligned16*>(Rh + off);
          const CxUnaligned16 l =
              *reinterpret_cast<const CxUnaligned16*>(Rl + off);
          bh[ni] = __builtin_bit_cast(uint4, h);
          bl[ni] = __builtin_bit_cast(uint4, l);
        }
      }
#pragma unroll
      for (int ma = 0; ma < MA; ++ma) {
        const int off = dbase + ((dy * 2 + half) * AC + 32 * ma + l31) * 8;
        const uint4 ah = *reinterpret_cast<const uint4*>(Dh + off);
        const uint4 al = *reinterpret_cast<const uint4*>(Dl + off);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[ma][ni] = cx_mfma<F16>(ah, bh[ni], acc[ma][ni]);
          acc[ma][ni] = cx_mfma<F16>(ah, bl[ni], acc[ma][ni]);
          acc[ma][ni] = cx_mfma<F16>(al, bh[ni], acc[ma][ni]);
        }
      }
    }
    // gradient step, threshold, extrapolation on the accumulator tiles, as
    // a pipeline over the 2 * MA tiles: the Y / code loads of tile t+1 are
    // issued before tile t is finished and stored (a buffer store orders
    // later loads behind it), those of tile 0 before the matrix loop.
    // Buffer addressing: lane offset (atom a0, row u, column v) + a scalar
    // offset per register; positions outside the code map get an
    // out-of-range offset (loads give 0, stores are dropped).
    load_tile(1, yB, cB);
    finish_tile(0, yA, cA, acc[0][0]);
    if (MA == 2) {
      load_tile(2, yA, cA);
      finish_tile(1, yB, cB, acc[0][1]);
      load_tile(3, yB, cB);
      finish_tile(2, yA, cA, acc[MA - 1][0]);
      finish_tile(3, yB, cB, acc[MA - 1][1]);
    } else {
      finish_tile(1, yB, cB, acc[0][1]);
    }
  }
  if (pp.delta_sum) {
    const double w = wave_sum(local);
    if ((tid & 63) == 0) atomicAdd(pp.delta_sum, w);
  }
  if (F16 && sc.y_out) {
    __syncthreads();                               // the window is free now
    cx_publish_max_word(y_max, sc.y_out + img, reinterpret_cast<unsigned*>(Rh));
  }
}

// ------------------------------------------------ fused iteration kernel
// One conv-FISTA iteration moves the code maps (36 MB per 256x256 image at
// 128 kernels) five times when synthesis and analysis are separate launches:
// the analysis reads Y and the codes and writes both, the synthesis reads the
// new Y once more -- with a 1.6x halo overlap.  Here the synthesis rides on the
// analysis epilogue: the new Y tile is still in the accumulator layout (lane =
// code column, register = atom), which -- with the atoms of a 16-atom k-step
// taken in the order (j & 3) + 8 (j >> 2) + 4 half, baked into the packed
// synthesis operand -- IS the B-operand layout of Q[tap, pos] = sum_s D[s, tap]
// Y'[s, pos].
//
// Blocks are persistent (one per CU, fixed chunk of 64 atoms: the operand
// planes of both products go to LDS once) and walk items of 8 code rows x 32
// code columns; wave w owns code row w of the item.  Per item a wave
//   * forms the analysis product for its row (two 32x32 accumulator tiles),
//   * finishes atom tile 0 (gradient step, threshold, extrapolation; Y' and the
//     codes go to memory), issues the Y / code loads of the NEXT item's tile 0
//     into the registers just freed, and feeds Y' to the synthesis product;
//     the same for atom tile 1 -- every code-map load has most of an item of
//     work between issue and use,
//   * folds Q (col2im, as in conv_synth_x3_kernel) into its private 11 x 42
//     window in LDS.
// The block then adds the 8 windows in wave order and writes ONE partial
// reconstruction tile (18 x 42) to its own slot of a small buffer;
// conv_partial_reduce_kernel sums, for every pixel, the <= 12 partial tiles
// that cover it in a fixed order and forms the next residual.  The residual
// window of the next item is fetched a whole item ahead and the private
// windows are double-buffered, which leaves one barrier per item.  Per
// iteration the code maps are read twice and written twice, nothing else of
// that size moves.
// FISTA, soft threshold, no early stopping (the configuration of every
// training run); other options take the two-kernel path.
template <int K>
struct CxFused {
  static constexpr int ROWS = 8;                    // code rows per item
  static constexpr int COLS = 32;                   // code columns per item
  static constexpr int AC = 64;                     // atoms per chunk
  static constexpr int WIN_PITCH = 48;              // residual window row, bf16
  static constexpr int WP = 48;                     // private window row, floats
  static constexpr int TH = ROWS + K - 1;           // partial tile rows
  static constexpr int TW = COLS + K - 1;           // partial tile columns
  static constexpr int SYN_PITCH = AC + 8;          // elements per slot row
  // partial tiles start on 128-byte lines (neighbouring tiles come from
  // different blocks)
  static constexpr int TILE_STRIDE = (TH * TW + 31) / 32 * 32;
  static constexpr int WIN_ELEMS = TH * WIN_PITCH;  // one plane of one window
  static constexpr int WIN_REGS = (WIN_ELEMS + 511) / 512;
  // The window operand of a lane starts at pixel l31 + 8 half: a 16-byte LDS
  // read at 2-byte alignment, which costs 52-64 LDS cycles per wave against 4
  // for an aligned one (tools/micro/lds_unaligned.hip).  The window is kept
  // in 4 copies, copy c shifted by c pixels, so that every lane reads two
  // 8-byte aligned halves from the copy c = start & 3.  Copy stride = 2 planes
  // + 32 elements: the 4 copies fall on disjoint LDS banks.
  static constexpr int WIN_COPY = 2 * WIN_ELEMS + 32;
  static constexpr int WIN_BUF = 4 * WIN_COPY;      // one window buffer, elements
  static constexpr size_t ana_bytes = (size_t)2 * K * AC * 16 * 2;
  static constexpr size_t syn_bytes =
      (size_t)2 * CxDims<K>::SLOTS * SYN_PITCH * 2;
  static constexpr size_t win_bytes = (size_t)2 * WIN_BUF * 2;
  static constexpr size_t priv_bytes = (size_t)2 * ROWS * K * WP * 4;
  static constexpr size_t lds = ana_bytes + syn_bytes + win_bytes + priv_bytes;
  static_assert(COLS + 15 < WIN_PITCH && TW <= WP && WIN_PITCH % 4 == 0,
                "window pitches");
};

// synp image (uint16): [chunk][plane][slot][AC + 8]: element a of a slot row
// holds D[chunk*64 + 32 (a>>5) + 16 ((a>>4)&1) + 4 ((a>>3)&1) + (a&3) +
// 8 ((a>>2)&1)][tap(slot)] -- the k order of the accumulator layout.
template <bool F16>
__global__ void conv_x3_pack_synp_kernel(const float* __restrict__ D,
                                         uint16_t* __restrict__ synp, int s,
                                         int k, int slots, int chunks,
                                         const float* __restrict__ dscale) {
  const float sigma = F16 ? dscale[0] : 1.f;
  const int taps = k * k;
  const int pitch = 64 + 8;
  const int64_t plane = (int64_t)slots * pitch;
  const int64_t total = (int64_t)chunks * plane;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int chunk = (int)(e / plane);
    const int rem = (int)(e % plane);
    const int slot = rem / pitch, a = rem % pitch;
    float v = 0.f;
    const int t = cx_slot_tap(slot, k);
    if (a < 64 && t >= 0) {
      const int j = a & 7;
      const int atom = chunk * 64 + 32 * (a >> 5) + 16 * ((a >> 4) & 1) +
                       4 * ((a >> 3) & 1) + (j & 3) + 8 * (j >> 2);
      if (atom < s) v = D[(int64_t)atom * taps + t];
    }
    uint16_t* hi = synp + (int64_t)chunk * 2 * plane + rem;
    cx_split1<F16>(v * sigma, hi[0], hi[plane]);
  }
}

// position of an item: image, first code row, first code column
struct CxItem {
  int img, u0, v0;
  bool valid;
};

// Code maps of the fused path ("fragment order").  Between the launches Y and
// the codes are kept as the kernel's accumulator tiles: the 16 registers x 64
// lanes of (item, chunk, code row, atom tile) form one 4 KB block,
//   block = ((((img * tiles_u + tu) * tiles_v + tv) * chunks + chunk) * 8 + row)
//           * 2 + atom tile
//   float (r, lane) of a block at  (r >> 2) * 256 + lane * 4 + (r & 3),
// so a tile moves with four 16-byte accesses per lane, 1 KB contiguous per
// instruction (a quarter of the instructions of dword-per-lane accesses to the
// (s, ch, cw) layout: a wave holds at most 64 vector-memory operations in
// flight, and 128 per item made it wait on its own stores), and every block is
// owned by one wave: no cache line is shared between blocks.  (The maps are
// updated in place and the L2s of the 8 XCDs are not coherent within a
// launch: with the (s, ch, cw) layout, whose rows are not multiples of 128
// bytes, blocks on different XCDs read-modify-writing parts of one line lost
// updates -- a ~10 % chance per launch of a wrong 64-byte piece.)
// The caller's (b, s, ch, cw) layout appears at the two ends only: initial
// codes are converted once, the last launch writes its codes in that layout.
// The momentum iterate Y is not stored at all: y_k = c_k + beta_(k-1) (c_k -
// c_(k-1)) is recomputed, with the very operations that formed it, from the
// last two code iterates -- three passes over the code maps per iteration
// (read c_k, read c_(k-1), write c_(k+1) where c_(k-1) was) instead of four.
struct CxMaps {
  const float* cur;    // c_k, fragment order
  float* old;          // c_(k-1) in, c_(k+1) out, fragment order
  float* user_codes;   // last launch: the caller's (b, s, ch, cw); else null
  float beta_prev;     // beta_(k-1); 0 in the first iteration (y_0 = c_0)
};

// floats of one fragment-order set of code maps
static size_t cx_frag_floats(const ConvGeo& g, int chunks, int rows, int cols) {
  return (size_t)g.b * ceil_div(g.ch, rows) * ceil_div(g.cw, cols) * chunks *
         rows * 2 * 1024;
}

// RAGGED: the atom count is not a multiple of 64 (the last chunk is partial)
// STAMP: per-section s_memtime sums (diagnostics, VTC_CONV_STAMPS=1)
template <int K, bool RAGGED, bool STAMP, bool F16>
__global__ __launch_bounds__(512) void conv_fused_x3_kernel(
    const float* __restrict__ R, const uint16_t* __restrict__ ana_image,
    const uint16_t* __restrict__ synp_image, CxMaps M,
    float* __restrict__ partial, ConvGeo g, int tiles_v, int tiles_u,
    int chunks, ProxParams pp, int do_synth, unsigned frag_bytes,
    unsigned long long* stamps, CxScales sc) {
  using Dm = CxDims<K>;
  using F = CxFused<K>;
  constexpr int AC = F::AC, MT = Dm::MT, WP = F::WP, WPITCH = F::WIN_PITCH;
  unsigned long long st_prev = 0, st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto stamp = [&](int slot_) {
    if (!STAMP) return;
    unsigned long long now;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
    st_acc[slot_] += now - st_prev;
    st_prev = now;
  };
  if (STAMP)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
  extern __shared__ __attribute__((aligned(16))) char cx_lds[];
  const int plane = K * AC * 16;                    // analysis plane, elements
  uint16_t* Dh = reinterpret_cast<uint16_t*>(cx_lds);
  uint16_t* Dl = Dh + plane;
  const int splane = Dm::SLOTS * F::SYN_PITCH;      // synthesis plane
  uint16_t* Sh = Dl + plane;
  uint16_t* Sl = Sh + splane;
  uint16_t* Win = Sl + splane;              // [buffer][copy][plane][TH][WPITCH]
  float* Priv = reinterpret_cast<float*>(Win + 2 * F::WIN_BUF);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // wave = code row of the item; scalar, so that everything derived from it
  // (block addresses of the code maps) stays in scalar registers
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  // Block -> (XCD, chunk, rank).  Workgroups go round-robin over the 8 XCDs;
  // every strip of one band of rows is given to blocks of the SAME XCD: the
  // residual windows of neighbouring strips overlap, and so do their partial
  // tiles' readers.
  const int xcd = blockIdx.x & 7;
  const int local = blockIdx.x >> 3;                // index within the XCD
  const int chunk = local % chunks;
  const int rank = local / chunks;                  // among the blocks of
  const int nper = (int)(gridDim.x >> 3) / chunks;  // this (XCD, chunk)
  // bands of this XCD: band = 8 t + xcd, t = 0 .. bands_here - 1
  const int bands = g.b * tiles_u;
  const int bands_here = (bands - xcd + 7) / 8;
  const int items = bands_here * tiles_v;           // host: fits 31 bits
  auto decode = [&](int j) -> CxItem {
    CxItem it;
    it.valid = j < items;
    const int jj = it.valid ? j : 0;
    const int band = (jj / tiles_v) * 8 + xcd;
    it.v0 = (jj % tiles_v) * F::COLS;
    it.u0 = (band % tiles_u) * F::ROWS;
    it.img = band / tiles_u;
    return it;
  };
  // F16 (see the top of the file): the residual windows enter as sigma_R R,
  // the kernels as sigma_D D, and the gradient step takes
  // eta / (sigma_R sigma_D); the new iterate enters the synthesis product
  // scaled per code column and the tap sums are scaled back before the fold
  float eta = pp.eta, inv_d = 1.f;                  // eta: of the current item
  if (F16) {
    inv_d = sc.dscale[1];
    cx_clear_words(sc.r_zero, sc.images);
  }
  // residual scale of an item's image and the step that undoes it
  auto item_scale = [&](const CxItem& it, float* rs, float* eta_eff) {
    *rs = 1.f;
    *eta_eff = pp.eta;
    if (F16 && it.valid) {
      // a SCALAR load (wave-uniform index): a vector load here would be the
      // youngest vector-memory operation of the wave, and waiting for it
      // would drain every code-map access in flight
      const int img_u = __builtin_amdgcn_readfirstlane(it.img);
      float inv_r;
      cx_scale_of_bits(sc.r_in[img_u], rs, &inv_r);
      *eta_eff = pp.eta * (inv_r * inv_d);
    }
  };
  CxItem cur = decode(rank);
  if (!cur.valid) return;                           // whole block
  float rs_cur, eta_cur, rs_nxt, eta_nxt;
  item_scale(cur, &rs_cur, &eta_cur);

  // residual window of an item: global -> registers -> bf16 hi / lo planes
  float wreg[F::WIN_REGS];
  auto window_fetch = [&](const CxItem& it) {
    if (!it.valid) return;
    const float* Rimg = R + (int64_t)it.img * g.H * g.W;
#pragma unroll
    for (int q = 0; q < F::WIN_REGS; ++q) {
      const int e = tid + 512 * q;
      const int ry = e / WPITCH, rx = e % WPITCH;
      const int y = it.u0 + ry, x = it.v0 + rx;
      wreg[q] = (e < F::WIN_ELEMS && y < g.H && x < g.W)
                    ? Rimg[y * (int)g.W + x] : 0.f;
    }
  };
  auto window_store = [&](int buf, float r_scale) {
    uint16_t* Wb = Win + buf * F::WIN_BUF;
#pragma unroll
    for (int q = 0; q < F::WIN_REGS; ++q) {
      const int e = tid + 512 * q;
      if (e < F::WIN_ELEMS) {
        uint16_t hb, lb;
        cx_split1<F16>(F16 ? wreg[q] * r_scale : wreg[q], hb, lb);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          // pixel x of a row sits at position x - c of copy c.  The first c
          // pixels of a row land on positions 45 .. 47 of the row before (or
          // in the padding of the copy before): reads end at position 43,
          // nothing there is ever used.
          Wb[c * F::WIN_COPY + e - c] = hb;
          Wb[c * F::WIN_COPY + F::WIN_ELEMS + e - c] = lb;
        }
      }
    }
  };
  {
    const uint4* src = reinterpret_cast<const uint4*>(
        ana_image + (int64_t)chunk * 2 * plane);
    uint4* dst = reinterpret_cast<uint4*>(Dh);
    for (int i = tid; i < plane / 4; i += 512) dst[i] = src[i];
    if (do_synth) {
      const uint4* ssrc = reinterpret_cast<const uint4*>(
          synp_image + (int64_t)chunk * 2 * splane);
      uint4* sdst = reinterpret_cast<uint4*>(Sh);
      for (int i = tid; i < splane / 4; i += 512) sdst[i] = ssrc[i];
    }
    window_fetch(cur);
    window_store(0, rs_cur);
  }
  __syncthreads();
  stamp(0);

  // fragment-order block of (item, this chunk, this wave's row, atom tile):
  // byte offset, wave-uniform (the scalar offset of the buffer accesses; the
  // lane part is lane * 16 for every access)
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
      (void*)M.old, 0, (int)frag_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)M.cur, 0, (int)frag_bytes, 0x00020000);
  const unsigned lane16 = (unsigned)lane * 16u;
  auto block_of = [&](const CxItem& it, int ma) -> unsigned {
    const int item_index =
        (it.img * tiles_u + it.u0 / F::ROWS) * tiles_v + it.v0 / F::COLS;
    return ((((unsigned)item_index * (unsigned)chunks + (unsigned)chunk) *
                 F::ROWS + (unsigned)wave) * 2u + (unsigned)ma) * 4096u;
  };
  auto load_tile = [&](const CxItem& it, int ma, float (&yv)[16],
                       float (&cv)[16]) {
    if (!it.valid || it.u0 + wave >= g.ch) return;  // wave-uniform
    const unsigned blk = block_of(it, ma);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const cx_u32x4 y4 = __builtin_amdgcn_raw_buffer_load_b128(
          ors, lane16 + 1024u * k4, blk, 0);
      const cx_u32x4 c4 = __builtin_amdgcn_raw_buffer_load_b128(
          crs, lane16 + 1024u * k4, blk, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        yv[4 * k4 + j] = __uint_as_float(y4[j]);
        cv[4 * k4 + j] = __uint_as_float(c4[j]);
      }
    }
  };
  // proximal step on atom tile ma of the current item, in registers (yv holds
  // c_(k-1), cv holds c_k): the new Y replaces the gradient in `tile` (zero
  // where outside the problem, for the synthesis), the new codes replace the
  // old ones in cv
  // early stopping only: sum of |c' - c| / eta of the item (added to the
  // global sum per item, so that nothing is carried through the item loop)
  float stop_sum = 0.f;
  auto prox_mode = [&](auto mode_tag, const CxItem& it, int ma,
                       const float (&yv)[16], float (&cv)[16], f32x16& tile) {
    constexpr int MODE = decltype(mode_tag)::value;
    const bool inside = it.v0 + l31 < g.cw;          // the row is (row_ok)
    const int left = g.s - (chunk * AC + 32 * ma + 4 * half);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2);
      // beta = 0 (ISTA, the first iterations) takes the iterate itself, as
      // the reference does: 0 * (inf - inf) would turn a diverged entry to NaN
      const float y = M.beta_prev != 0.f
                          ? add_rn(cv[r], mul_rn(M.beta_prev,
                                                 sub_rn(cv[r], yv[r])))
                          : cv[r];
      const float p = sub_rn(y, mul_rn(eta, tile[r]));
      const float c = shrink(p, pp.cutoff, MODE);
      const float d = sub_rn(c, cv[r]);
      const float y1 = pp.beta != 0.f ? add_rn(c, mul_rn(pp.beta, d)) : c;
      const bool ok = inside && (!RAGGED || rr < left);
      if (pp.delta_sum && ok) stop_sum += fabsf(d) / pp.eta;
      cv[r] = ok ? c : 0.f;
      tile[r] = ok ? y1 : 0.f;
    }
  };
  // (ISTA is the same recursion with every beta = 0; the threshold is one of
  // four compile-time variants behind a uniform branch)
  auto prox_tile = [&](const CxItem& it, int ma, const float (&yv)[16],
                       float (&cv)[16], f32x16& tile) {
    switch (pp.mode) {
      case VTC_SOFT:
        prox_mode(std::integral_constant<int, VTC_SOFT>{}, it, ma, yv, cv, tile);
        break;
      case VTC_SOFT_NONNEG:
        prox_mode(std::integral_constant<int, VTC_SOFT_NONNEG>{}, it, ma, yv,
                  cv, tile);
        break;
      case VTC_HARD:
        prox_mode(std::integral_constant<int, VTC_HARD>{}, it, ma, yv, cv, tile);
        break;
      default:
        prox_mode(std::integral_constant<int, VTC_HARD_NONNEG>{}, it, ma, yv,
                  cv, tile);
        break;
    }
  };
  auto store_tile = [&](const CxItem& it, int ma, const f32x16& tile,
                        const float (&cv)[16]) {
    if (M.user_codes) {
      // last launch: codes only, in the caller's layout (dword per lane:
      // lane = column, register = atom)
      const int64_t map = (int64_t)g.ch * g.cw;
      const unsigned map4 = (unsigned)(map * 4);
      const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(M.user_codes + it.img * g.s * map), 0,
          (int)((int64_t)g.s * map * 4), 0x00020000);
      const int u = it.u0 + wave, v = it.v0 + l31;
      const int a0 = chunk * AC + 32 * ma + 4 * half;
      const unsigned off = v < g.cw ? (unsigned)a0 * map4 +
                                          (unsigned)(u * (int)g.cw + v) * 4u
                                    : 0x80000000u;
      const int left = g.s - a0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2);
        __builtin_amdgcn_raw_buffer_store_b32(
            __float_as_uint(cv[r]), crs,
            (!RAGGED || rr < left) ? off : 0x80000000u, (unsigned)rr * map4, 0);
      }
      return;
    }
    const unsigned blk = block_of(it, ma);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      cx_u32x4 c4;
#pragma unroll
      for (int j = 0; j < 4; ++j) c4[j] = __float_as_uint(cv[4 * k4 + j]);
      // block offset in the vector offset, scalar offset 0: with a register
      // there hipcc leaves out the wait state a 16-byte store needs before a
      // VALU instruction overwrites its data registers (epi_prox.h)
      __builtin_amdgcn_raw_buffer_store_b128(c4, ors,
                                             lane16 + 1024u * k4 + blk, 0, 0);
    }
  };
  f32x16 Q[MT];
  // Q += D[chunk atoms of tile ma, taps]^T * Y' (Y' as the B operand).  The
  // LDS reads of operand tile mt+1 are issued before the products of tile mt.
  auto synth_tile = [&](int ma, const f32x16& yn, float col_scale) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float v8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v8[j] = yn[8 * ks + j];
      uint4 bh, bl;
      cx_split8<F16>(v8, col_scale, bh, bl);
      const int base = l31 * F::SYN_PITCH + 32 * ma + 16 * ks + 8 * half;
      uint4 ah = *reinterpret_cast<const uint4*>(Sh + base);
      uint4 al = *reinterpret_cast<const uint4*>(Sl + base);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        uint4 ah_n = ah, al_n = al;
        if (mt + 1 < MT) {
          const int off = base + 32 * (mt + 1) * F::SYN_PITCH;
          ah_n = *reinterpret_cast<const uint4*>(Sh + off);
          al_n = *reinterpret_cast<const uint4*>(Sl + off);
        }
        __builtin_amdgcn_sched_barrier(0);           // keep the reads up here
        Q[mt] = cx_mfma<F16>(ah, bh, Q[mt]);
        Q[mt] = cx_mfma<F16>(ah, bl, Q[mt]);
        Q[mt] = cx_mfma<F16>(al, bh, Q[mt]);
        __builtin_amdgcn_sched_barrier(0);
        ah = ah_n;
        al = al_n;
      }
    }
  };
  // col2im of Q (taps x 32 code columns) into the wave's window: register
  // (mt, r) of lane half h holds tap idx + h * HTAPS at code column l31, i.e.
  // pixel (dy + h * HROWS, l31 + dx); the dx sum is formed by lane rotation
  // first (see conv_synth_x3_kernel).  Within one fold no two lanes touch the
  // same pixel, and nobody else touches this window until the barrier.
  auto fold = [&](float* mine) {
    // (opaque to the optimiser: otherwise the K - 1 rotation addresses are
    // hoisted out of the item loop and held in registers the loop needs)
    int lane4 = l31 * 4;
    asm volatile("" : "+v"(lane4));
    const int half_bit = (lane & 32) * 4;
    const int pyb = half * Dm::HROWS;
    float* base = mine + pyb * WP + l31;
#pragma unroll
    for (int dy = 0; dy < Dm::HROWS; ++dy) {
      const bool row_ok = (pyb + dy) < K;
      const bool spill_ok = row_ok && l31 < K - 1;
      float* pm = base + dy * WP;
      float* ps = base + dy * WP + 32;
      float main_sum = 0.f, spill_sum = 0.f;
#pragma unroll
      for (int dx = 0; dx < K; ++dx) {
        const int idx = dy * K + dx;                 // compile time
        float rot = Q[idx / 16][idx % 16];
        if (dx != 0) {
          const int from4 = ((lane4 - 4 * dx) & 124) | half_bit;
          rot = __builtin_bit_cast(
              float, __builtin_amdgcn_ds_bpermute(
                         from4, __builtin_bit_cast(int, rot)));
        }
        if (l31 >= dx)
          main_sum = add_rn(main_sum, rot);
        else
          spill_sum = add_rn(spill_sum, rot);
      }
      // the only fold of the item covers the whole window: plain stores
      if (row_ok) *pm = main_sum;
      if (spill_ok) *ps = spill_sum;
    }
  };

  // analysis operands of tap row dy: the two window planes (8 consecutive
  // pixels per lane, unaligned) and the kernel planes of both atom tiles
  struct AnaOps {
    uint4 bh, bl, ah0, al0, ah1, al1;
  };
  const int win_start = l31 + 8 * half;             // first pixel of the lane
  const int win_lane = (win_start & 3) * F::WIN_COPY + (win_start & ~3);
  auto ana_load = [&](const uint16_t* Wb, int dy) -> AnaOps {
    AnaOps o;
    const uint16_t* bp = Wb + win_lane + (wave + dy) * WPITCH;
    const uint2 h0 = *reinterpret_cast<const uint2*>(bp);
    const uint2 h1 = *reinterpret_cast<const uint2*>(bp + 4);
    const uint2 l0 = *reinterpret_cast<const uint2*>(bp + F::WIN_ELEMS);
    const uint2 l1 = *reinterpret_cast<const uint2*>(bp + F::WIN_ELEMS + 4);
    o.bh = make_uint4(h0.x, h0.y, h1.x, h1.y);
    o.bl = make_uint4(l0.x, l0.y, l1.x, l1.y);
    const int off = ((dy * 2 + half) * AC + l31) * 8;
    o.ah0 = *reinterpret_cast<const uint4*>(Dh + off);
    o.al0 = *reinterpret_cast<const uint4*>(Dl + off);
    o.ah1 = *reinterpret_cast<const uint4*>(Dh + off + 32 * 8);
    o.al1 = *reinterpret_cast<const uint4*>(Dl + off + 32 * 8);
    return o;
  };

  float yA[16], cA[16], yB[16], cB[16];
  load_tile(cur, 0, yA, cA);
  load_tile(cur, 1, yB, cB);
  CxItem nxt = decode(rank + nper);
  item_scale(nxt, &rs_nxt, &eta_nxt);
  window_fetch(nxt);
  for (int i = 0; cur.valid; ++i) {
    const int buf = i & 1;
    const bool row_ok = cur.u0 + wave < g.ch;        // wave-uniform
    const uint16_t* Wb = Win + buf * F::WIN_BUF;
    eta = eta_cur;
    // The window of the NEXT item goes to the other buffer now: it was last
    // read before the previous barrier, and everything this wave has in
    // flight (the next window, this item's code maps) was issued most of an
    // item ago -- the one full wait on memory per item costs nothing here.
    window_store(buf ^ 1, rs_nxt);
    const CxItem nn = decode(rank + (i + 2) * nper);
    float rs_nn, eta_nn;                             // used two items from now
    item_scale(nn, &rs_nn, &eta_nn);
    if (row_ok) {
      f32x16 acc[2];
#pragma unroll
      for (int ma = 0; ma < 2; ++ma)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ma][r] = 0.f;
      // the LDS reads of tap row dy+1 are in flight under the products of dy
      AnaOps ops = ana_load(Wb, 0);
#pragma unroll
      for (int dy = 0; dy < K; ++dy) {
        AnaOps nx = ops;
        if (dy + 1 < K) nx = ana_load(Wb, dy + 1);
        __builtin_amdgcn_sched_barrier(0);           // keep the reads up here
        acc[0] = cx_mfma<F16>(ops.ah0, ops.bh, acc[0]);
        acc[1] = cx_mfma<F16>(ops.ah1, ops.bh, acc[1]);
        acc[0] = cx_mfma<F16>(ops.ah0, ops.bl, acc[0]);
        acc[1] = cx_mfma<F16>(ops.ah1, ops.bl, acc[1]);
        acc[0] = cx_mfma<F16>(ops.al0, ops.bh, acc[0]);
        acc[1] = cx_mfma<F16>(ops.al1, ops.bh, acc[1]);
        __builtin_amdgcn_sched_barrier(0);
        ops = nx;
      }
      window_fetch(nn);                              // for the item after next
      stamp(1);
      if (do_synth) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) Q[mt][r] = 0.f;
      }
      // Both proximal steps before any store: while loads and stores are
      // mixed in flight every wait on a load drains the stores as well.  The
      // products of tile 0's synthesis run under the arithmetic of tile 1;
      // then stores, then the loads of the next item into the freed registers
      // (most of an item of work between their issue and their use).
      prox_tile(cur, 0, yA, cA, acc[0]);
      // F16: one power of two per code column (= lane, both halves) and atom
      // tile, from the 32 values of the column in that tile; the tap sums are
      // carried from the first tile's scale to the second's (a power of two,
      // exact) and scaled back after it
      float col_s0 = 1.f, col_u0 = 1.f;
      auto column_scale = [&](const f32x16& t, float* cs, float* cu) {
        float m = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(t[r]));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        cx_scale_of_bits(__float_as_uint(m), cs, cu);
      };
      if (do_synth) {
        if (F16) column_scale(acc[0], &col_s0, &col_u0);
        synth_tile(0, acc[0], col_s0);
      }
      prox_tile(cur, 1, yB, cB, acc[1]);
      if (pp.delta_sum) {
        const double w = wave_sum((double)stop_sum);
        if (lane == 0) atomicAdd(pp.delta_sum, w);
        stop_sum = 0.f;
      }
      stamp(2);
      store_tile(cur, 0, acc[0], cA);
      store_tile(cur, 1, acc[1], cB);
      stamp(7);
      load_tile(nxt, 0, yA, cA);
      load_tile(nxt, 1, yB, cB);
      stamp(8);
      if (do_synth) {
        float col_s1 = 1.f, col_u1 = 1.f;
        if (F16) {
          column_scale(acc[1], &col_s1, &col_u1);
          const float carry = col_s1 * col_u0;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) Q[mt][r] *= carry;
        }
        synth_tile(1, acc[1], col_s1);
        if (F16) {
          const float back = col_u1 * inv_d;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) Q[mt][r] *= back;
        }
      }
      stamp(3);
      if (do_synth) fold(Priv + (buf * F::ROWS + wave) * K * WP);
      stamp(4);
    } else {
      window_fetch(nn);
      load_tile(nxt, 0, yA, cA);
      load_tile(nxt, 1, yB, cB);
    }
    __syncthreads();
    stamp(5);
    if (do_synth) {
      // partial tile = the windows of the waves that had a code row, added in
      // wave order; window w covers pixel rows w .. w + K - 1 of the tile
      const float* pw = Priv + buf * F::ROWS * K * WP;
      float* out = partial +
                   ((((int64_t)(cur.img * tiles_u + cur.u0 / F::ROWS) * chunks +
                      chunk) * tiles_v) + cur.v0 / F::COLS) * F::TILE_STRIDE;
      for (int e = tid; e < F::TH * F::TW; e += 512) {
        const int py = e / F::TW, px = e % F::TW;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < F::ROWS; ++w) {
          const int ry = py - w;
          if (ry >= 0 && ry < K && cur.u0 + w < g.ch)
            sum = add_rn(sum, pw[(w * K + ry) * WP + px]);
        }
        out[e] = sum;
      }
    }
    stamp(6);
    cur = nxt;
    nxt = nn;
    rs_cur = rs_nxt;
    eta_cur = eta_nxt;
    rs_nxt = rs_nn;
    eta_nxt = eta_nn;
  }
  if (STAMP && lane == 0) {
    for (int q = 0; q < 9; ++q) atomicAdd(stamps + q, st_acc[q]);
    atomicAdd(stamps + 9, 1ull);
  }
}

// (b, s, ch, cw) -> fragment order (positions outside the maps stay as they
// are: the destination is zeroed beforehand).  One thread per destination
// float4 = 4 atoms (a0 + 8 k + 4 half + j, j = 0..3) of one position.
__global__ void conv_to_fragments_kernel(const float* __restrict__ src,
                                         float* __restrict__ dst, ConvGeo g,
                                         int tiles_u, int tiles_v, int chunks,
                                         int rows, int cols) {
  const int64_t total = (int64_t)g.b * tiles_u * tiles_v * chunks * rows * 2 * 256;
  const int64_t map = (int64_t)g.ch * g.cw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), k4 = (int)((i >> 6) & 3);
    int64_t rest = i >> 8;
    const int ma = (int)(rest & 1); rest >>= 1;
    const int row = (int)(rest % rows); rest /= rows;
    const int chunk = (int)(rest % chunks); rest /= chunks;
    const int tv = (int)(rest % tiles_v); rest /= tiles_v;
    const int tu = (int)(rest % tiles_u);
    const int64_t img = rest / tiles_u;
    const int u = tu * rows + row, v = tv * cols + (lane & 31);
    if (u >= g.ch || v >= g.cw) continue;
    const int a0 = chunk * 64 + 32 * ma + 8 * k4 + 4 * (lane >> 5);
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* p = src + (img * g.s + a0) * map + (int64_t)u * g.cw + v;
    if (a0 + 0 < g.s) out.x = p[0];
    if (a0 + 1 < g.s) out.y = p[map];
    if (a0 + 2 < g.s) out.z = p[2 * map];
    if (a0 + 3 < g.s) out.w = p[3 * map];
    reinterpret_cast<float4*>(dst)[i] = out;
  }
}

// fragment order -> (b, s, ch, cw)
__global__ void conv_from_fragments_kernel(const float* __restrict__ src,
                                           float* __restrict__ dst, ConvGeo g,
                                           int tiles_u, int tiles_v,
                                           int chunks, int rows, int cols) {
  const int64_t total = (int64_t)g.b * tiles_u * tiles_v * chunks * rows * 2 * 256;
  const int64_t map = (int64_t)g.ch * g.cw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), k4 = (int)((i >> 6) & 3);
    int64_t rest = i >> 8;
    const int ma = (int)(rest & 1); rest >>= 1;
    const int row = (int)(rest % rows); rest /= rows;
    const int chunk = (int)(rest % chunks); rest /= chunks;
    const int tv = (int)(rest % tiles_v); rest /= tiles_v;
    const int tu = (int)(rest % tiles_u);
    const int64_t img = rest / tiles_u;
    const int u = tu * rows + row, v = tv * cols + (lane & 31);
    if (u >= g.ch || v >= g.cw) continue;
    const int a0 = chunk * 64 + 32 * ma + 8 * k4 + 4 * (lane >> 5);
    const float4 in = reinterpret_cast<const float4*>(src)[i];
    float* p = dst + (img * g.s + a0) * map + (int64_t)u * g.cw + v;
    if (a0 + 0 < g.s) p[0] = in.x;
    if (a0 + 1 < g.s) p[map] = in.y;
    if (a0 + 2 < g.s) p[2 * map] = in.z;
    if (a0 + 3 < g.s) p[3 * map] = in.w;
  }
}

// residual = mask * (sum of the partial tiles covering the pixel - image)
// grid = (blocks per image, images); r_max_out (may be null): the per-image
// words that receive max |R| (CxScales)
template <int K>
__global__ __launch_bounds__(1024) void conv_partial_reduce_kernel(
    const float* __restrict__ partial, const float* __restrict__ X,
    float* __restrict__ R, ConvGeo g, int tiles_v, int tiles_u, int chunks,
    unsigned* r_max_out) {
  using F = CxFused<K>;
  __shared__ unsigned red[16];
  float r_max = 0.f;
  const int64_t img = blockIdx.y;
  const unsigned per_image = (unsigned)g.H * (unsigned)g.W;
  for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < per_image;
       p += gridDim.x * blockDim.x) {
    const unsigned W = (unsigned)g.W;
    const int y = (int)(p / W), x = (int)(p - (unsigned)y * W);
    int tu_lo = (y - F::TH + F::ROWS) / F::ROWS;    // ceil((y - TH + 1) / ROWS)
    if (y - F::TH + 1 <= 0) tu_lo = 0;
    int tu_hi = y / F::ROWS;
    if (tu_hi > tiles_u - 1) tu_hi = tiles_u - 1;
    int tv_lo = (x - F::TW + F::COLS) / F::COLS;
    if (x - F::TW + 1 <= 0) tv_lo = 0;
    int tv_hi = x / F::COLS;
    if (tv_hi > tiles_v - 1) tv_hi = tiles_v - 1;
    float sum = 0.f;
    for (int tu = tu_lo; tu <= tu_hi; ++tu)
      for (int c = 0; c < chunks; ++c)
        for (int tv = tv_lo; tv <= tv_hi; ++tv) {
          const float* tile = partial + (((img * tiles_u + tu) * chunks + c) *
                                             (int64_t)tiles_v + tv) *
                                            F::TILE_STRIDE;
          sum = add_rn(sum, tile[(y - tu * F::ROWS) * F::TW +
                                 (x - tv * F::COLS)]);
        }
    const int64_t i = img * (int64_t)per_image + p;
    const float rv = mul_rn(mask_at(g, y, x), sub_rn(sum, X[i]));
    R[i] = rv;
    r_max = fmaxf(r_max, fabsf(rv));
  }
  if (r_max_out) cx_publish_max_word(r_max, r_max_out + img, red);
}

// ---------------------------------------------- dictionary gradient (a8/a9)
// dD[s, dy, dx] = sum_{image, u, v} C[s, u, v] * r[u + dy, v + dx]
// (dict_update_rules/convolutional/sc_steepest_descent.py:60-65) as a
// contraction over code positions: M = atoms, N = taps (an accumulator tile =
// 2 tap rows x 16 dx), K = 16 consecutive code columns of one code row.
//   A: a lane holds 8 consecutive codes of one atom, straight from HBM (two
//      16-byte buffer loads) and split to bf16 hi / lo in registers;
//   B: a lane holds 8 consecutive residual pixels of window row u + dy,
//      starting at column v + dx: the same unaligned 16-byte LDS read as in the
//      analysis kernel.
// Blocks are persistent: each walks its share of (image, 8 code rows, 64 code
// columns) items with the partial dD of 4 x 32 atoms in registers (one atom
// tile per wave), writes one slab at the end; the slabs are summed in a fixed
// order.  Atoms beyond 128 are handled by further passes (grid.y).
constexpr int kCxGradRows = 8;

template <int K>
__global__ __launch_bounds__(256) void conv_grad_x3_kernel(
    const float* __restrict__ R, const float* __restrict__ C,
    float* __restrict__ slabs, ConvGeo g, int tiles_v, int tiles_u,
    int64_t items) {
  constexpr int NT = (K + 1) / 2;                  // accumulator tiles (2 dy each)
  constexpr int WROWS = kCxGradRows + 2 * NT;      // window rows incl. padding
  // The window operand of a lane starts at pixel 16 ks + 8 half + dx: a
  // 16-byte LDS read at 2-byte alignment, 52-64 LDS cycles per wave against 4
  // aligned (tools/micro/lds_unaligned.hip) -- 12 such reads per 18 MFMAs made
  // this kernel LDS-bound 4x over.  As in the fused iteration kernel the
  // window is kept in 4 copies, copy c shifted by c pixels, and a lane reads
  // two 8-byte aligned halves from copy dx & 3; the copy stride is padded to
  // 16 or 48 (mod 64) dwords so that the copies sit on disjoint banks.
  constexpr int PLANE = WROWS * kCxAnaPitch;       // elements
  constexpr int kPadDw =
      (PLANE % 64 == 16 || PLANE % 64 == 48) ? 0 : (16 - PLANE % 64 + 64) % 64;
  constexpr int COPY = 2 * PLANE + 2 * kPadDw;     // hi plane, lo plane, pad
  __shared__ __attribute__((aligned(16))) uint16_t Wc[4 * COPY + 8];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int atom0 = (blockIdx.y * 4 + wave) * 32;  // this wave's atom tile
  const int channel = blockIdx.z;                  // image channel
  const int atom = atom0 + l31;
  const bool atom_ok = atom < g.s;
  const int64_t map = (int64_t)g.ch * g.cw;
  const unsigned map4 = (unsigned)(map * 4);
  const int dyi = l31 >> 4, dx = l31 & 15;

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

  for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
    const int tile_v = (int)(item % tiles_v);
    const int tile_u = (int)((item / tiles_v) % tiles_u);
    const int64_t img = item / ((int64_t)tiles_v * tiles_u);
    const int u0 = tile_u * kCxGradRows, v0 = tile_v * kCxStrip;
    __syncthreads();                               // window of the last item
    {
      const float* Rimg = R + (img * g.c + channel) * g.H * (int64_t)g.W;
      for (int e = tid; e < PLANE; e += 256) {
        const int ry = e / kCxAnaPitch, rx = e % kCxAnaPitch;
        const int y = u0 + ry, x = v0 + rx;
        const float v = (y < g.H && x < g.W) ? Rimg[(int64_t)y * g.W + x] : 0.f;
        const __bf16 h = (__bf16)v;
        const uint16_t hb = cx_bits(h);
        const uint16_t lb = cx_bits((__bf16)(v - (float)h));
        // pixel x of a row sits at position x - c of copy c; the first c
        // pixels of a row land on the last positions of the row (or plane)
        // before, which no read reaches (reads end at position 78 of 88)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          Wc[8 + c * COPY + e - c] = hb;
          Wc[8 + c * COPY + PLANE + e - c] = lb;
        }
      }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(C + img * g.s * map), 0, (int)((int64_t)g.s * map * 4),
        0x00020000);
    for (int lu = 0; lu < kCxGradRows; ++lu) {
      const int u = u0 + lu;
      if (u >= g.ch) break;                        // whole block
#pragma unroll
      for (int ks = 0; ks < kCxStrip / 16; ++ks) {
        const int v = v0 + 16 * ks + 8 * half;     // first of this lane's 8
        float a[8];
        {
          const unsigned off =
              atom_ok ? (unsigned)atom * map4 + (unsigned)(u * g.cw + v) * 4u
                      : 0x80000000u;
          const cx_u32x4 lo4 =
              __builtin_amdgcn_raw_buffer_load_b128(crs, off, 0, 0);
          const cx_u32x4 hi4 =
              __builtin_amdgcn_raw_buffer_load_b128(crs, off, 16, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            a[i] = (v + i < g.cw) ? __uint_as_float(lo4[i]) : 0.f;
            a[4 + i] = (v + 4 + i < g.cw) ? __uint_as_float(hi4[i]) : 0.f;
          }
        }
        uint4 ah, al;
        cx_split8<false>(a, 1.f, ah, al);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const uint16_t* bp = Wc + 8 + (dx & 3) * COPY +
                               (lu + 2 * nt + dyi) * kCxAnaPitch + 16 * ks +
                               8 * half + (dx & ~3);
          const uint2 h0 = *reinterpret_cast<const uint2*>(bp);
          const uint2 h1 = *reinterpret_cast<const uint2*>(bp + 4);
          const uint2 l0 = *reinterpret_cast<const uint2*>(bp + PLANE);
          const uint2 l1 = *reinterpret_cast<const uint2*>(bp + PLANE + 4);
          const uint4 bh = make_uint4(h0.x, h0.y, h1.x, h1.y);
          const uint4 bl = make_uint4(l0.x, l0.y, l1.x, l1.y);
          acc[nt] = cx_mfma<false>(ah, bh, acc[nt]);
          acc[nt] = cx_mfma<false>(ah, bl, acc[nt]);
          acc[nt] = cx_mfma<false>(al, bh, acc[nt]);
        }
      }
    }
  }
  // this block's partial sums: slab[blockIdx.x][atom][channel][tap]
  float* slab = slabs + ((int64_t)blockIdx.x * g.s * g.c + channel) * (K * K);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int dy = 2 * nt + dyi;
    if (dy >= K || dx >= K) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a_out = atom0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (a_out < g.s)
        slab[(int64_t)a_out * g.c * (K * K) + dy * K + dx] = acc[nt][r];
    }
  }
}

// ------------------------------------------------------------------ host
struct CxPlan {
  int k, s16, syn_chunks, slots, AC, chunks;
  int syn_rows;      // code rows per wave of the synthesis kernel
  int ana_rows;      // code rows per block of the analysis kernel
  int ana_shift;     // window copies (conv_analysis_x3_kernel<.., SHIFT>): 1, 2, 4
  size_t syn_image_bytes, ana_image_bytes;
  size_t syn_lds, ana_lds;
  int th, tw;
  // fused iteration kernel (conv_fused_x3_kernel): 0 when not applicable
  size_t synp_image_bytes, partial_bytes, fused_lds;
  size_t padded_bytes;   // one fragment-order set of code maps
};

static int cx_compute_units() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount,
                              dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;
  }
  return cus;
}

// Code rows per wave of the synthesis kernel.  One block per CU is resident
// (the operand planes fill most of the LDS), a block's work grows with the
// rows, its halo overhead shrinks with them: take the count that minimises
// rounds(blocks / CUs) * rows, the larger one on a tie, within the LDS budget.
// (configs[4], b = 8: 4 rows -> 624 blocks = 3 rounds, 5 rows -> 480 blocks =
// 2 rounds; measured 0.070 -> 0.063 ms per image-iteration.)
static int cx_pick_rows(const ConvGeo& g, int k, int tw, int pw,
                        size_t image_bytes) {
  const int cus = cx_compute_units();
  const int64_t tiles_x = ceil_div(g.W, tw);
  int best = 0;
  int64_t best_cost = 0;
  for (int rows = 2; rows <= kCxSynMaxRows; ++rows) {
    const int th = kCxSynWaves * rows - (k - 1);
    if (th < 1) continue;
    const size_t lds = image_bytes +
                       (size_t)kCxSynWaves * th * pw * sizeof(float);
    if (lds > 150 * 1024) break;
    const int64_t blocks = tiles_x * ceil_div(g.H, th) * g.b;
    const int64_t cost = ceil_div(blocks, cus) * rows;
    if (best == 0 || cost <= best_cost) {
      best = rows;
      best_cost = cost;
    }
  }
  return best;   // 0: not even the smallest tile fits
}

template <int K>
static void cx_fill_plan(const ConvGeo& g, CxPlan* p) {
  using Dm = CxDims<K>;
  p->k = K;
  // atoms per LDS-resident chunk of the synthesis planes: everything when the
  // planes stay under 100 KB (every 11x11 set up to 128 kernels), else the
  // fewest equal chunks that do
  {
    const int all16 = (g.s + 15) / 16 * 16;
    int parts = 1, per = all16;
    while ((size_t)2 * Dm::SLOTS * (per + 8) * 2 > 100 * 1024 && per > 16) {
      ++parts;
      per = ((all16 / 16 + parts - 1) / parts) * 16;
    }
    p->s16 = per;
    p->syn_chunks = (all16 + per - 1) / per;
  }
  p->slots = Dm::SLOTS;
  // atoms per analysis block: 64, or 32 when the kernel planes of all image
  // channels would not leave room for the windows (3 channels of 11x11)
  p->AC = (g.s > 32 && (size_t)g.c * 2 * K * 64 * 16 * 2 <= 100 * 1024) ? 64
                                                                         : 32;
  p->chunks = (g.s + p->AC - 1) / p->AC;
  const size_t syn_channel_bytes = (size_t)2 * Dm::SLOTS * (p->s16 + 8) * 2;
  p->syn_image_bytes = (size_t)g.c * p->syn_chunks * syn_channel_bytes;
  p->ana_image_bytes = (size_t)p->chunks * g.c * 2 * K * p->AC * 16 * 2;
  p->syn_rows = cx_pick_rows(g, K, Dm::TW, Dm::PW, syn_channel_bytes);
  p->th = kCxSynWaves * p->syn_rows - (K - 1);
  p->syn_lds = syn_channel_bytes +
               (size_t)kCxSynWaves * p->th * Dm::PW * sizeof(float);
  // Code rows per analysis block.  Two blocks per CU are resident; with few
  // rounds of blocks the shorter tile fills them better, with many the taller
  // one reloads the kernel planes half as often (same-box A/B: configs[4],
  // b = 8: 10.44 vs 10.60 ms; 512x512 images: 34.87 vs 34.44 ms).
  {
    const int64_t blocks8 = ceil_div(g.cw, kCxStrip) *
                            ceil_div(g.ch, kCxAnaMaxRows) * p->chunks * g.b;
    p->ana_rows = blocks8 < (int64_t)16 * cx_compute_units() ? 4 : 8;
  }
  {
    const size_t planes = (size_t)2 * K * p->AC * 16 * 2;
    const size_t win = (size_t)(p->ana_rows + K - 1) * kCxAnaPitch;
    const size_t plain = g.c * (planes + 2 * win * 2);
    const size_t shifted = g.c * (planes + 4 * (2 * win + 32) * 2);
    // aligned window copies when they fit and do not cost a resident block
    const size_t budget = 150 * 1024, lds = 160 * 1024;
    const size_t two = g.c * (planes + 2 * (2 * win + 32) * 2);
    const size_t blocks_plain = lds / plain < 2 ? lds / plain : 2;
    p->ana_shift = 1;
    p->ana_lds = plain;
    if (shifted <= budget && lds / shifted >= blocks_plain) {
      p->ana_shift = 4;
      p->ana_lds = shifted;
    } else if (two <= budget && lds / two >= blocks_plain) {
      p->ana_shift = 2;
      p->ana_lds = two;
    }
  }
  p->tw = Dm::TW;
  using F = CxFused<K>;
  p->synp_image_bytes = p->partial_bytes = p->fused_lds = 0;
  p->padded_bytes = 0;
  if (g.c == 1 && p->AC == 64 && K <= 11 && F::lds <= 160 * 1024) {
    p->fused_lds = F::lds;
    p->synp_image_bytes = (size_t)p->chunks * F::syn_bytes;
    p->partial_bytes = (size_t)g.b * ceil_div(g.ch, F::ROWS) * p->chunks *
                       ceil_div(g.cw, F::COLS) * F::TILE_STRIDE * sizeof(float);
    p->padded_bytes =
        cx_frag_floats(g, p->chunks, F::ROWS, F::COLS) * sizeof(float);
    // 32-bit byte offsets into the fragment-order maps and the partial tiles
    if (p->padded_bytes >= ((size_t)1 << 32) ||
        p->partial_bytes >= ((size_t)1 << 32)) {
      p->fused_lds = 0;
      p->synp_image_bytes = p->partial_bytes = p->padded_bytes = 0;
    }
  }
}

// Geometries this path covers: stride 1, square kernels of the sizes
// instantiated below, and operand planes that fit the 160 KiB LDS (up to 4
// image channels of 11x11 or 16x16 kernels).  Several channels: the synthesis
// runs per channel (grid.y), the analysis takes (channel, dy, 16 dx) as its K
// index; the fused iteration kernel is single-channel.
static bool cx_plan(const ConvGeo& g, CxPlan* p) {
  if (g.c < 1 || g.c > 8 || g.sv != 1 || g.sh != 1 || g.kh != g.kw)
    return false;
  switch (g.kh) {
    case 5: cx_fill_plan<5>(g, p); break;
    case 8: cx_fill_plan<8>(g, p); break;
    case 11: cx_fill_plan<11>(g, p); break;
    case 16: cx_fill_plan<16>(g, p); break;
    default: return false;
  }
  // 32-bit byte offsets within one image's code maps (buffer addressing)
  const int64_t code_bytes =
      (int64_t)p->s16 * p->syn_chunks * g.ch * g.cw * 4;
  return p->syn_rows > 0 && p->syn_lds <= 150 * 1024 &&
         p->ana_lds <= 150 * 1024 &&
         g.b <= 65535 && code_bytes < (int64_t)0x7fffffff;
}

static size_t cx_image_bytes(const CxPlan& p) {
  return align_up(p.syn_image_bytes, 256) + align_up(p.ana_image_bytes, 256);
}

// extra workspace of the fused iteration kernel
static size_t cx_fused_bytes(const CxPlan& p) {
  return align_up(p.synp_image_bytes, 256) + align_up(p.partial_bytes, 256) +
         2 * align_up(p.padded_bytes, 256);
}

template <int K, bool RAGGED, bool F16>
static int cx_launch_fused_k(const float* R, const uint16_t* ana,
                             const uint16_t* synp, const CxMaps& maps,
                             float* partial, const float* X, float* R_next,
                             const ConvGeo& g, const CxPlan& p,
                             const ProxParams& pp, bool do_synth,
                             const CxScales& sc, hipStream_t st) {
  using F = CxFused<K>;
  const int tiles_v = (int)ceil_div(g.cw, F::COLS);
  const int tiles_u = (int)ceil_div(g.ch, F::ROWS);
  static unsigned long long attr_set = 0;
  if (first_use_on_this_device(&attr_set)) {
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(
            conv_fused_x3_kernel<K, RAGGED, false, F16>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(
            conv_fused_x3_kernel<K, RAGGED, true, F16>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  // persistent blocks, one per CU: the same number for every (XCD, chunk)
  const int64_t bands = (int64_t)g.b * tiles_u;
  const int64_t items_per_xcd = ceil_div(bands, 8) * tiles_v;
  int64_t nper = cx_compute_units() / (8 * p.chunks);
  if (nper < 1) nper = 1;
  if (nper > items_per_xcd) nper = items_per_xcd;
  const int64_t blocks = 8 * nper * p.chunks;
  static const bool want_stamps = getenv("VTC_CONV_STAMPS") != nullptr;
  unsigned long long* stamps_dev = nullptr;
  if (want_stamps) {
    VTC_HIP_CHECK(hipMalloc(&stamps_dev, 80));
    VTC_HIP_CHECK(hipMemsetAsync(stamps_dev, 0, 80, st));
  }
  if (stamps_dev)
    hipLaunchKernelGGL((conv_fused_x3_kernel<K, RAGGED, true, F16>),
                       dim3((unsigned)blocks), dim3(512), F::lds, st, R, ana,
                       synp, maps, partial, g, tiles_v, tiles_u, p.chunks, pp,
                       do_synth ? 1 : 0, (unsigned)p.padded_bytes, stamps_dev,
                       sc);
  else
    hipLaunchKernelGGL((conv_fused_x3_kernel<K, RAGGED, false, F16>),
                       dim3((unsigned)blocks), dim3(512), F::lds, st, R, ana,
                       synp, maps, partial, g, tiles_v, tiles_u, p.chunks, pp,
                       do_synth ? 1 : 0, (unsigned)p.padded_bytes, nullptr,
                       sc);
  VTC_LAUNCH_CHECK();
  if (stamps_dev) {
    unsigned long long host[10];
    VTC_HIP_CHECK(hipMemcpyAsync(host, stamps_dev, 80, hipMemcpyDeviceToHost, st));
    VTC_HIP_CHECK(hipStreamSynchronize(st));
    VTC_HIP_CHECK(hipFree(stamps_dev));
    const char* names[9] = {"setup", "window+analysis", "prox A, B + synth A",
                            "synth B", "fold", "barrier", "partial-out",
                            "stores", "loads"};
    fprintf(stderr, "[vtc conv stamps] do_synth=%d\n", do_synth ? 1 : 0);
    for (int q = 0; q < 9; ++q)
      fprintf(stderr, "[vtc conv stamps] %-20s %8.0f ticks/wave\n", names[q],
              (double)host[q] / (double)host[9]);
  }
  if (do_synth) {
    // Blocks of 1024 pixels: all blocks of an image finish together and each
    // adds one atomic to the image's maximum word (f16 mode) -- with 256-pixel
    // blocks the ~300 read-modify-writes per word queued for 30 us behind a
    // 13 us kernel (a look at the word first does not help: every block
    // starts while it is still zero).
    int64_t rblocks = ceil_div((int64_t)g.H * g.W, 1024);
    if (rblocks > 4096) rblocks = 4096;
    hipLaunchKernelGGL(conv_partial_reduce_kernel<K>,
                       dim3((unsigned)rblocks, (unsigned)g.b), dim3(1024), 0,
                       st, partial, X, R_next, g, tiles_v, tiles_u, p.chunks,
                       F16 ? sc.r_out : nullptr);
    VTC_LAUNCH_CHECK();
  }
  return VTC_OK;
}

// one fused iteration: (R, Y, C) -> (Y', C'), and R' unless it is the last one
// sc.dscale != null selects the f16 split (CxScales)
static int cx_launch_fused(const float* R, const uint16_t* ana,
                           const uint16_t* synp, const CxMaps& maps,
                           float* partial, const float* X, float* R_next,
                           const ConvGeo& g, const CxPlan& p,
                           const ProxParams& pp, bool do_synth,
                           const CxScales& sc, hipStream_t st) {
  const bool f16 = sc.dscale != nullptr;
#define VTC_CX_FUSED_T(KK, RG)                                                 \
  (f16 ? cx_launch_fused_k<KK, RG, true>(R, ana, synp, maps, partial, X,       \
                                         R_next, g, p, pp, do_synth, sc, st)   \
       : cx_launch_fused_k<KK, RG, false>(R, ana, synp, maps, partial, X,      \
                                          R_next, g, p, pp, do_synth, sc, st))
#define VTC_CX_FUSED(KK)                                                       \
  case KK:                                                                     \
    return (g.s % 64) ? VTC_CX_FUSED_T(KK, true) : VTC_CX_FUSED_T(KK, false)
  switch (p.k) {
    VTC_CX_FUSED(5);
    VTC_CX_FUSED(8);
    VTC_CX_FUSED(11);
  }
#undef VTC_CX_FUSED
#undef VTC_CX_FUSED_T
  set_error("conv bf16x3: kernel size not instantiated");
  return VTC_ERR_UNSUPPORTED;
}

// dscale != null: f16 operands of sigma_D D (dscale = {sigma_D, 1 / sigma_D},
// already on the device)
static int cx_pack(const float* D, const ConvGeo& g, const CxPlan& p,
                   uint16_t* syn, uint16_t* ana, const float* dscale,
                   hipStream_t st) {
  if (dscale)
    hipLaunchKernelGGL(conv_x3_pack_kernel<true>, dim3(256, (unsigned)g.c),
                       dim3(256), 0, st, D, syn, ana, g.s, p.k, p.s16,
                       p.syn_chunks, p.slots, p.AC, p.chunks, dscale);
  else
    hipLaunchKernelGGL(conv_x3_pack_kernel<false>, dim3(256, (unsigned)g.c),
                       dim3(256), 0, st, D, syn, ana, g.s, p.k, p.s16,
                       p.syn_chunks, p.slots, p.AC, p.chunks, dscale);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

template <int K, bool F16>
static int cx_launch_synth_k(const float* Y, const uint16_t* syn,
                             const float* X, float* R, const ConvGeo& g,
                             const CxPlan& p, const CxScales& sc,
                             hipStream_t st) {
  using Dm = CxDims<K>;
  const int tiles_x = (int)ceil_div(g.W, Dm::TW);
  const int tiles_y = (int)ceil_div(g.H, p.th);
  static unsigned long long attr_set = 0;
  if (first_use_on_this_device(&attr_set)) {
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(conv_synth_x3_kernel<K, F16>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  const int64_t blocks = ceil_div(g.b * tiles_x, 8) * 8 * tiles_y;
  if (blocks > 0x7fffffffLL) {
    set_error("conv bf16x3: too many tiles");
    return VTC_ERR_INVALID_ARGUMENT;
  }
  hipLaunchKernelGGL((conv_synth_x3_kernel<K, F16>),
                     dim3((unsigned)blocks, (unsigned)g.c), dim3(512),
                     p.syn_lds, st, Y, syn, X, R, g, p.s16, p.syn_chunks,
                     tiles_x, tiles_y, p.syn_rows, sc);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

static int cx_launch_synth(const float* Y, const uint16_t* syn, const float* X,
                           float* R, const ConvGeo& g, const CxPlan& p,
                           const CxScales& sc, hipStream_t st) {
  const bool f16 = sc.dscale != nullptr;
#define VTC_CX_SYNTH(KK)                                                      \
  case KK:                                                                    \
    return f16 ? cx_launch_synth_k<KK, true>(Y, syn, X, R, g, p, sc, st)      \
               : cx_launch_synth_k<KK, false>(Y, syn, X, R, g, p, sc, st)
  switch (p.k) {
    VTC_CX_SYNTH(5);
    VTC_CX_SYNTH(8);
    VTC_CX_SYNTH(11);
    VTC_CX_SYNTH(16);
  }
#undef VTC_CX_SYNTH
  set_error("conv bf16x3: kernel size not instantiated");
  return VTC_ERR_UNSUPPORTED;
}

template <int MA, bool FAST, bool F16, int SHIFT>
static int cx_launch_analysis_m(const float* R, const uint16_t* ana, float* Y,
                                float* C, const ConvGeo& g, const CxPlan& p,
                                const ProxParams& pp, const CxScales& sc,
                                hipStream_t st) {
  const int tiles_v = (int)ceil_div(g.cw, kCxStrip);
  const int tiles_u = (int)ceil_div(g.ch, p.ana_rows);
  static unsigned long long attr_set = 0;
  if (first_use_on_this_device(&attr_set)) {
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(
            conv_analysis_x3_kernel<MA, FAST, F16, SHIFT>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  const int64_t bands = (int64_t)tiles_u * p.chunks * g.b;
  const int64_t blocks = ceil_div(bands, 8) * 8 * tiles_v;
  if (blocks > 0x7fffffffLL) {
    set_error("conv bf16x3: too many tiles");
    return VTC_ERR_INVALID_ARGUMENT;
  }
  hipLaunchKernelGGL((conv_analysis_x3_kernel<MA, FAST, F16, SHIFT>),
                     dim3((unsigned)blocks), dim3(256), p.ana_lds, st, R, ana,
                     Y, C, g, tiles_v, tiles_u, p.chunks, p.ana_rows, pp, sc);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

static int cx_launch_analysis(const float* R, const uint16_t* ana, float* Y,
                              float* C, const ConvGeo& g, const CxPlan& p,
                              const ProxParams& pp, const CxScales& sc,
                              hipStream_t st) {
  const bool fast = pp.fista && pp.mode == VTC_SOFT && !pp.delta_sum;
  const bool f16 = sc.dscale != nullptr;
#define VTC_CX_ANA_S(MA_, FAST_, F16_)                                        \
  (p.ana_shift == 4                                                           \
       ? cx_launch_analysis_m<MA_, FAST_, F16_, 4>(R, ana, Y, C, g, p, pp, sc, \
                                                   st)                        \
       : p.ana_shift == 2                                                     \
             ? cx_launch_analysis_m<MA_, FAST_, F16_, 2>(R, ana, Y, C, g, p,  \
                                                         pp, sc, st)          \
             : cx_launch_analysis_m<MA_, FAST_, F16_, 1>(R, ana, Y, C, g, p,  \
                                                         pp, sc, st))
#define VTC_CX_ANA(MA_, FAST_)                                                \
  (f16 ? VTC_CX_ANA_S(MA_, FAST_, true) : VTC_CX_ANA_S(MA_, FAST_, false))
  if (p.AC == 64) return fast ? VTC_CX_ANA(2, true) : VTC_CX_ANA(2, false);
  return fast ? VTC_CX_ANA(1, true) : VTC_CX_ANA(1, false);
#undef VTC_CX_ANA
#undef VTC_CX_ANA_S
}

// blocks of the gradient kernel (and slabs of its output)
static int cx_grad_blocks(const ConvGeo& g) {
  const int64_t items = ceil_div(g.cw, kCxStrip) *
                        ceil_div(g.ch, kCxGradRows) * g.b;
  const int64_t want = (int64_t)4 * cx_compute_units();
  return (int)(items < want ? items : want);
}

template <int K>
static int cx_launch_grad_k(const float* R, const float* C, float* slabs,
                            const ConvGeo& g, hipStream_t st) {
  const int tiles_v = (int)ceil_div(g.cw, kCxStrip);
  const int tiles_u = (int)ceil_div(g.ch, kCxGradRows);
  const int64_t items = (int64_t)tiles_v * tiles_u * g.b;
  hipLaunchKernelGGL(conv_grad_x3_kernel<K>,
                     dim3((unsigned)cx_grad_blocks(g),
                          (unsigned)ceil_div(g.s, 128), (unsigned)g.c),
                     dim3(256), 0, st, R, C, slabs, g, tiles_v, tiles_u, items);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

static int cx_launch_grad(const float* R, const float* C, float* slabs,
                          const ConvGeo& g, const CxPlan& p, hipStream_t st) {
  switch (p.k) {
    case 5: return cx_launch_grad_k<5>(R, C, slabs, g, st);
    case 8: return cx_launch_grad_k<8>(R, C, slabs, g, st);
    case 11: return cx_launch_grad_k<11>(R, C, slabs, g, st);
    case 16: return cx_launch_grad_k<16>(R, C, slabs, g, st);
  }
  set_error("conv bf16x3: kernel size not instantiated");
  return VTC_ERR_UNSUPPORTED;
}

}  // namespace vtc
