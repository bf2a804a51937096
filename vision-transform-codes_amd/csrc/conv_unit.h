// Unit-stride, square-kernel specialisations of the convolutional kernels
// (BASELINE configs[4]: 128 kernels of 11x11 on 256x256 images, stride 1).
//
// With stride 1 every thread of a wave walks the SAME kernel taps, so the
// taps are wave-uniform and come through scalar loads (SGPR operands of the
// FMAs); the only vector-side traffic is one row segment of the LDS window per
// (kernel, tap row), reused for K taps x 8 outputs.  Both kernels are then
// bound by the f32 FMA rate:
//   synthesis  recon[y, x..x+7]  += code_row[s][y-dy][x-dx .. ] * D[s][dy][dx]
//   analysis   g[s][p, q..q+7]   += resid_row[p+dy][q+dx .. ]   * D[s][dy][dx]
// A block covers a 32 x 64 tile (pixels resp. code positions), a thread a
// strip of 8 along x.  Exact f32, fixed summation order.
#pragma once

namespace vtc {

constexpr int kUnitTY = 32;
constexpr int kUnitTX = 64;
constexpr int kUnitStrip = 8;
constexpr int kUnitSynChunk = 4;   // kernels staged per pass (synthesis)
constexpr int kUnitAnaChunk = 4;   // kernels accumulated per pass (analysis)

template <int K>
struct UnitDims {
  static constexpr int WY = kUnitTY + K - 1;
  static constexpr int WX = kUnitTX + K - 1;
  static constexpr int WP = (WX + 3) / 4 * 4;   // LDS row pitch, floats
  static constexpr int SEG = kUnitStrip + K - 1;
  static constexpr int NSEG4 = (SEG + 3) / 4;   // float4 loads per segment
  static constexpr int KP = (K + 3) / 4 * 4;    // tap row pitch in LDS
  static constexpr int NT4 = KP / 4;
  static_assert(kUnitStrip * 7 + 4 * NSEG4 <= WP, "segment overruns the row");
};

// One (kernel, tap-row) step of the direct convolution for a strip of 8
// outputs: `seg` holds SEG consecutive window values, `taps` one row of K taps.
// FLIP: synthesis walks the taps against the segment (x - dx), analysis with
// it (q + dx).
template <int K, bool FLIP, int NSEG4, int NT4>
__device__ __forceinline__ void strip_fma(const float4 (&seg4)[NSEG4],
                                          const float4 (&tap4)[NT4],
                                          float (&acc)[kUnitStrip]) {
  float seg[4 * NSEG4], tap[4 * NT4];
#pragma unroll
  for (int u = 0; u < NSEG4; ++u) {
    seg[4 * u + 0] = seg4[u].x; seg[4 * u + 1] = seg4[u].y;
    seg[4 * u + 2] = seg4[u].z; seg[4 * u + 3] = seg4[u].w;
  }
#pragma unroll
  for (int u = 0; u < NT4; ++u) {
    tap[4 * u + 0] = tap4[u].x; tap[4 * u + 1] = tap4[u].y;
    tap[4 * u + 2] = tap4[u].z; tap[4 * u + 3] = tap4[u].w;
  }
#pragma unroll
  for (int dx = 0; dx < K; ++dx)
#pragma unroll
    for (int j = 0; j < kUnitStrip; ++j)
      acc[j] = fmaf(seg[FLIP ? j + (K - 1) - dx : j + dx], tap[dx], acc[j]);
}

// ------------------------------------------------------------- synthesis
template <int K>
__global__ __launch_bounds__(256) void conv_synth_unit_kernel(
    const float* __restrict__ codes, const float* __restrict__ D,
    const float* __restrict__ images, float* __restrict__ out, ConvGeo g,
    int tiles_x, int groups, int per_group) {
  // groups == 1: out = mask * (synthesis - images)   (the residual)
  // groups  > 1: out[group] = partial synthesis over this group's kernels;
  //              the analysis kernel adds the groups up (fixed order) and
  //              forms the residual while staging its window.  Splitting the
  //              kernel sum over blocks is what fills the chip at small b.
  using U = UnitDims<K>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Wl = lds;                                        // [chunk][WY][WP]
  float* Tl = lds + kUnitSynChunk * U::WY * U::WP;        // [chunk][K][KP]
  const int tid = threadIdx.x;
  const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x % tiles_x;
  const int chan = blockIdx.y / groups, grp = blockIdx.y % groups;
  const int64_t img = blockIdx.z;
  const int s_begin = grp * per_group;
  const int s_end = (s_begin + per_group < g.s) ? s_begin + per_group : g.s;
  const int y0 = tile_y * kUnitTY, x0 = tile_x * kUnitTX;
  const int ty = tid >> 3, tx = tid & 7;
  const int y = y0 + ty, xs = x0 + tx * kUnitStrip;
  const int64_t map = (int64_t)g.ch * g.cw;
  float acc[kUnitStrip];
#pragma unroll
  for (int j = 0; j < kUnitStrip; ++j) acc[j] = 0.f;

  // step idx = sl * K + dy: code row y - dy sits at window row ty + (K-1) - dy,
  // the strip needs window columns 8tx .. 8tx + SEG - 1
  auto fetch = [&](int idx, float4 (&seg)[U::NSEG4], float4 (&tap)[U::NT4]) {
    const int sl = idx / K, dy = idx % K;
    const float* row = Wl + ((size_t)sl * U::WY + ty + (K - 1) - dy) * U::WP +
                       tx * kUnitStrip;
#pragma unroll
    for (int u = 0; u < U::NSEG4; ++u)
      seg[u] = *reinterpret_cast<const float4*>(row + 4 * u);
    const float* trow = Tl + (size_t)idx * U::KP;
#pragma unroll
    for (int u = 0; u < U::NT4; ++u)
      tap[u] = *reinterpret_cast<const float4*>(trow + 4 * u);
  };

  for (int s0 = s_begin; s0 < s_end; s0 += kUnitSynChunk) {
    const int ns = (s_end - s0 < kUnitSynChunk) ? s_end - s0 : kUnitSynChunk;
    __syncthreads();
    // window origin in code coordinates: (y0 - (K-1), x0 - (K-1))
    for (int e = tid; e < ns * U::WY * U::WP; e += 256) {
      const int sl = e / (U::WY * U::WP), rem = e % (U::WY * U::WP);
      const int i = rem / U::WP, j = rem % U::WP;
      const int p = y0 - (K - 1) + i, q = x0 - (K - 1) + j;
      float v = 0.f;
      if (j < U::WX && p >= 0 && p < g.ch && q >= 0 && q < g.cw)
        v = codes[(img * g.s + s0 + sl) * map + (int64_t)p * g.cw + q];
      Wl[e] = v;
    }
    for (int e = tid; e < ns * K * U::KP; e += 256) {
      const int row = e / U::KP, dx = e % U::KP;   // row = sl * K + dy
      const int sl = row / K, dy = row % K;
      Tl[e] = (dx < K) ? D[((int64_t)(s0 + sl) * g.c + chan) * (K * K) +
                           dy * K + dx]
                       : 0.f;
    }
    __syncthreads();
    // software pipeline over the ns*K steps: fetch step i+1 while step i
    // multiplies; two named register sets keep every index static
    const int steps = ns * K;
    float4 segA[U::NSEG4], tapA[U::NT4], segB[U::NSEG4], tapB[U::NT4];
    fetch(0, segA, tapA);
    for (int i = 0; i < steps; i += 2) {
      if (i + 1 < steps) fetch(i + 1, segB, tapB);
      strip_fma<K, true>(segA, tapA, acc);
      if (i + 2 < steps) fetch(i + 2, segA, tapA);
      if (i + 1 < steps) strip_fma<K, true>(segB, tapB, acc);
    }
  }
  if (y < g.H) {
#pragma unroll
    for (int j = 0; j < kUnitStrip; ++j) {
      const int x = xs + j;
      if (x < g.W) {
        const int64_t i = ((img * g.c + chan) * g.H + y) * (int64_t)g.W + x;
        if (groups == 1)
          out[i] = mul_rn(mask_at(g, y, x), sub_rn(acc[j], images[i]));
        else
          out[(int64_t)grp * g.b * g.c * g.H * g.W + i] = acc[j];
      }
    }
  }
}

// -------------------------------------------------------------- analysis
template <int K>
__global__ __launch_bounds__(256) void conv_analysis_unit_kernel(
    const float* __restrict__ residual, const float* __restrict__ images,
    int syn_groups, const float* __restrict__ D, float* __restrict__ Y,
    float* __restrict__ C, ConvGeo g, int tiles_q, int per_group,
    ProxParams pp) {
  // syn_groups == 1: `residual` is the masked residual.
  // syn_groups  > 1: `residual` holds syn_groups partial syntheses; the
  //                  residual mask * (sum - images) is formed here.
  // blockIdx.z: this block's share of the kernels (independent outputs).
  using U = UnitDims<K>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Rl = lds;                                   // [c][WY][WP]
  float* Tl = lds + (size_t)g.c * U::WY * U::WP;     // [per_group][c][K][KP]
  const int tid = threadIdx.x;
  const int tile_p = blockIdx.x / tiles_q, tile_q = blockIdx.x % tiles_q;
  const int64_t img = blockIdx.y;
  const int s_begin = blockIdx.z * per_group;
  const int s_end = (s_begin + per_group < g.s) ? s_begin + per_group : g.s;
  const int p0 = tile_p * kUnitTY, q0 = tile_q * kUnitTX;
  const int64_t group_stride = (int64_t)g.b * g.c * g.H * g.W;
  for (int e = tid; e < g.c * U::WY * U::WP; e += 256) {
    const int chan = e / (U::WY * U::WP), rem = e % (U::WY * U::WP);
    const int i = rem / U::WP, j = rem % U::WP;
    const int yy = p0 + i, xx = q0 + j;
    float v = 0.f;
    if (j < U::WX && yy < g.H && xx < g.W) {
      const int64_t at = ((img * g.c + chan) * g.H + yy) * (int64_t)g.W + xx;
      if (syn_groups == 1) {
        v = residual[at];
      } else {
        float sum = residual[at];
        for (int grp = 1; grp < syn_groups; ++grp)
          sum = add_rn(sum, residual[grp * group_stride + at]);
        v = mul_rn(mask_at(g, yy, xx), sub_rn(sum, images[at]));
      }
    }
    Rl[e] = v;
  }
  // taps of this block's kernels, rows padded to KP; kernels past s_end (the
  // last chunk may be short) are zero and their outputs are discarded
  const int rows_per_kernel = g.c * K;
  for (int e = tid; e < per_group * rows_per_kernel * U::KP; e += 256) {
    const int row = e / U::KP, dx = e % U::KP;
    const int sk = s_begin + row / rows_per_kernel;
    Tl[e] = (dx < K && sk < s_end)
                ? D[(int64_t)sk * rows_per_kernel * K +
                    (row % rows_per_kernel) * K + dx]
                : 0.f;
  }
  __syncthreads();
  const int tp = tid >> 3, tq = tid & 7;
  const int p = p0 + tp, qs = q0 + tq * kUnitStrip;
  const int64_t map = (int64_t)g.ch * g.cw;
  double local = 0.0;
  // step idx = chan * K + dy: residual rows p + dy
  auto fetch_seg = [&](int idx, float4 (&seg)[U::NSEG4]) {
    const int chan = idx / K, dy = idx % K;
    const float* row = Rl + ((size_t)chan * U::WY + tp + dy) * U::WP +
                       tq * kUnitStrip;
#pragma unroll
    for (int u = 0; u < U::NSEG4; ++u)
      seg[u] = *reinterpret_cast<const float4*>(row + 4 * u);
  };
  const int steps = g.c * K;
  for (int s0 = s_begin; s0 < s_end; s0 += kUnitAnaChunk) {
    float acc[kUnitAnaChunk][kUnitStrip];
#pragma unroll
    for (int a = 0; a < kUnitAnaChunk; ++a)
#pragma unroll
      for (int j = 0; j < kUnitStrip; ++j) acc[a][j] = 0.f;
    const float* tchunk = Tl + (size_t)(s0 - s_begin) * rows_per_kernel * U::KP;
    float4 segA[U::NSEG4], segB[U::NSEG4];
    fetch_seg(0, segA);
    for (int i = 0; i < steps; i += 2) {
      if (i + 1 < steps) fetch_seg(i + 1, segB);
#pragma unroll
      for (int a = 0; a < kUnitAnaChunk; ++a) {
        float4 tap[U::NT4];
        const float* trow = tchunk + ((size_t)a * rows_per_kernel + i) * U::KP;
#pragma unroll
        for (int u = 0; u < U::NT4; ++u)
          tap[u] = *reinterpret_cast<const float4*>(trow + 4 * u);
        strip_fma<K, false>(segA, tap, acc[a]);
      }
      if (i + 2 < steps) fetch_seg(i + 2, segA);
      if (i + 1 < steps) {
#pragma unroll
        for (int a = 0; a < kUnitAnaChunk; ++a) {
          float4 tap[U::NT4];
          const float* trow =
              tchunk + ((size_t)a * rows_per_kernel + i + 1) * U::KP;
#pragma unroll
          for (int u = 0; u < U::NT4; ++u)
            tap[u] = *reinterpret_cast<const float4*>(trow + 4 * u);
          strip_fma<K, false>(segB, tap, acc[a]);
        }
      }
    }
    if (p < g.ch) {
#pragma unroll
      for (int a = 0; a < kUnitAnaChunk; ++a) {
        if (s0 + a >= s_end) continue;
#pragma unroll
        for (int j = 0; j < kUnitStrip; ++j) {
          const int q = qs + j;
          if (q >= g.cw) continue;
          const int64_t idx =
              (img * g.s + s0 + a) * map + (int64_t)p * g.cw + q;
          const float yv = Y[idx];
          const float c = shrink(sub_rn(yv, mul_rn(pp.eta, acc[a][j])),
                                 pp.cutoff, pp.mode);
          float d;
          if (pp.fista) {
            d = sub_rn(c, C[idx]);
            pp.y_out[idx] = add_rn(c, mul_rn(pp.beta, d));
          } else {
            d = sub_rn(c, yv);
          }
          pp.c_out[idx] = c;
          if (pp.delta_sum) local += (double)(fabsf(d) / pp.eta);
        }
      }
    }
  }
  if (pp.delta_sum) {
    const double wsum = wave_sum(local);
    if ((tid & 63) == 0) atomicAdd(pp.delta_sum, wsum);
  }
}

// ----------------------------------------------------------------- dispatch
static inline bool unit_geometry(const ConvGeo& g) {
  return g.sv == 1 && g.sh == 1 && g.kh == g.kw &&
         (g.kh == 5 || g.kh == 8 || g.kh == 11 || g.kh == 16);
}

// How many blocks share the kernels of one tile: enough blocks to give every
// CU several, kernels per block a multiple of the staging chunk.
static inline int unit_groups(const ConvGeo& g, int tiles, int* per_group) {
  int64_t want = ceil_div(1536, (int64_t)tiles * g.b);
  const int64_t cap = ceil_div(g.s, 4 * kUnitSynChunk);
  if (want > cap) want = cap;
  if (want > 8) want = 8;
  if (want < 1) want = 1;
  int per = (int)ceil_div(g.s, want);
  per = (per + kUnitSynChunk - 1) / kUnitSynChunk * kUnitSynChunk;
  *per_group = per;
  return (int)ceil_div(g.s, per);
}

template <int K>
static int launch_synth_unit_k(const float* codes, const float* D,
                               const float* images, float* out,
                               const ConvGeo& g, int groups, int per_group,
                               hipStream_t st) {
  using U = UnitDims<K>;
  const size_t lds = (size_t)kUnitSynChunk *
                     (U::WY * U::WP + K * U::KP) * sizeof(float);
  auto kernel = conv_synth_unit_kernel<K>;
  static unsigned long long configured = 0;
  if (lds > 64 * 1024 && first_use_on_this_device(&configured)) {
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  const int tiles_x = (int)ceil_div(g.W, kUnitTX);
  const int tiles_y = (int)ceil_div(g.H, kUnitTY);
  hipLaunchKernelGGL(kernel,
                     dim3((unsigned)(tiles_x * tiles_y),
                          (unsigned)(g.c * groups), (unsigned)g.b),
                     dim3(256), lds, st, codes, D, images, out, g, tiles_x,
                     groups, per_group);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

template <int K>
static int launch_analysis_unit_k(const float* residual, const float* images,
                                  int syn_groups, const float* D, float* Y,
                                  float* C, const ConvGeo& g,
                                  const ProxParams& pp, hipStream_t st) {
  using U = UnitDims<K>;
  const int tiles_q = (int)ceil_div(g.cw, kUnitTX);
  const int tiles_p = (int)ceil_div(g.ch, kUnitTY);
  int per_group = 0;
  const int groups = unit_groups(g, tiles_p * tiles_q, &per_group);
  const size_t lds = ((size_t)g.c * U::WY * U::WP +
                      (size_t)per_group * g.c * K * U::KP) * sizeof(float);
  auto kernel = conv_analysis_unit_kernel<K>;
  if (lds > 64 * 1024)
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(conv_analysis_unit_kernel<K>,
                     dim3((unsigned)(tiles_p * tiles_q), (unsigned)g.b,
                          (unsigned)groups),
                     dim3(256), lds, st, residual, images, syn_groups, D, Y, C,
                     g, tiles_q, per_group, pp);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

// The analysis window of all channels must fit LDS for the specialisation.
static inline bool unit_analysis_fits(const ConvGeo& g) {
  const size_t wy = kUnitTY + g.kh - 1, wx = (kUnitTX + g.kw - 1 + 3) / 4 * 4;
  const size_t kp = (g.kw + 3) / 4 * 4;
  // window of all channels + the taps of at most ceil(s/1) kernels of a group
  // (groups only shrink this): keep well inside the 160 KiB
  return ((size_t)g.c * wy * wx + (size_t)g.s * g.c * g.kh * kp) *
             sizeof(float) <= 140 * 1024;
}

// groups == 1 -> `out` receives the masked residual; otherwise the partial
// syntheses (groups * b*c*H*W floats).
static int launch_synth_unit(const float* codes, const float* D,
                             const float* images, float* out, const ConvGeo& g,
                             int groups, int per_group, hipStream_t st) {
  switch (g.kh) {
    case 5:
      return launch_synth_unit_k<5>(codes, D, images, out, g, groups,
                                    per_group, st);
    case 8:
      return launch_synth_unit_k<8>(codes, D, images, out, g, groups,
                                    per_group, st);
    case 11:
      return launch_synth_unit_k<11>(codes, D, images, out, g, groups,
                                     per_group, st);
    default:
      return launch_synth_unit_k<16>(codes, D, images, out, g, groups,
                                     per_group, st);
  }
}

static int launch_analysis_unit(const float* residual, const float* images,
                                int syn_groups, const float* D, float* Y,
                                float* C, const ConvGeo& g,
                                const ProxParams& pp, hipStream_t st) {
  switch (g.kh) {
    case 5:
      return launch_analysis_unit_k<5>(residual, images, syn_groups, D, Y, C,
                                       g, pp, st);
    case 8:
      return launch_analysis_unit_k<8>(residual, images, syn_groups, D, Y, C,
                                       g, pp, st);
    case 11:
      return launch_analysis_unit_k<11>(residual, images, syn_groups, D, Y, C,
                                        g, pp, st);
    default:
      return launch_analysis_unit_k<16>(residual, images, syn_groups, D, Y, C,
                                        g, pp, st);
  }
}

}  // namespace vtc
