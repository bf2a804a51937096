// Convolutional ISTA/FISTA and the convolutional dictionary update.
//
// Restates
//   analysis_transforms/convolutional/ista_fista.py:141-193
//   dict_update_rules/convolutional/sc_steepest_descent.py:54-72
//   dict_update_rules/convolutional/sc_cheap_quadratic_descent.py:59-79
//   utils/convolutions.py:14-24
// Four routes, chosen per geometry and precision in vtc_conv_ista_fista:
//   conv_x3.h     one channel, stride 1, square kernels 5/8/11/16, VTC_BF16X3:
//                 both convolutions as split-bf16 MFMA contractions
//   conv_unit.h   stride 1, square kernels, VTC_F32: direct f32 kernels with
//                 scalar taps
//   conv_patch.h  strides > 1 with few covering positions, VTC_F32: exact-f32
//                 MFMA contractions over an im2col copy of the residual
//   this file     everything else: direct f32 kernels, the code window / the
//                 residual window of a tile staged in LDS
// plus the convolutional dictionary gradient / apply kernels.  None writes an
// im2col buffer for stride 1.  The VTC_F32 routes are exact f32 with fixed
// summation orders (on gfx950 the f32-input MFMA runs at the f32 VALU rate, so
// for a direct exact-f32 convolution the matrix cores buy nothing).
//
// Index conventions (b image, s kernel, c channel, (p,q) code position,
// (y,x) padded-image pixel, (dy,dx) kernel tap):
//   synthesis  recon[b,c,y,x]  = sum_{s,p,q} codes[b,s,p,q] D[s,c,y-p*sv,x-q*sh]
//   analysis   g[b,s,p,q]      = sum_{c,dy,dx} r[b,c,p*sv+dy,q*sh+dx] D[s,c,dy,dx]
//   gradient   dD[s,c,dy,dx]   = sum_{b,p,q} codes[b,s,p,q] r[b,c,p*sv+dy,q*sh+dx]
#include "common.h"

#include <stdlib.h>
#include <type_traits>
#include "gemm_f32.h"
#include "fc_fused.h"

#include <vector>

namespace vtc {

struct ConvGeo {
  int64_t b;
  int c, H, W, s, kh, kw, sv, sh, ch, cw;
  int has_pad, lead_v, trail_v, lead_h, trail_h;
};

static inline int code_dim(int padded, int kernel, int stride) {
  // utils/convolutions.py:14-15 : 1 + ceil((padded - kernel) / stride)
  const int span = padded - kernel;
  return 1 + (span >= 0 ? (span + stride - 1) / stride : -((-span) / stride));
}

static int make_geo(const vtc_conv_geometry* g, ConvGeo* out) {
  VTC_REQUIRE(g, "conv: null geometry");
  VTC_REQUIRE(g->b >= 0 && g->c > 0 && g->h > 0 && g->w > 0 && g->s > 0 &&
                  g->kh > 0 && g->kw > 0 && g->stride_v > 0 && g->stride_h > 0,
              "conv: non-positive dimension");
  VTC_REQUIRE(g->h >= g->kh && g->w >= g->kw, "conv: kernel exceeds image");
  ConvGeo o;
  o.b = g->b;
  o.c = g->c; o.H = g->h; o.W = g->w; o.s = g->s; o.kh = g->kh; o.kw = g->kw;
  o.sv = g->stride_v; o.sh = g->stride_h;
  o.ch = code_dim(g->h, g->kh, g->stride_v);
  o.cw = code_dim(g->w, g->kw, g->stride_h);
  o.has_pad = g->has_padding;
  o.lead_v = g->pad_lead_v; o.trail_v = g->pad_trail_v;
  o.lead_h = g->pad_lead_h; o.trail_h = g->pad_trail_h;
  // conv_transpose2d of the codes must give back exactly (H, W); otherwise
  // the reference fails in `mask * (recon - images)` with a shape error.
  if ((o.ch - 1) * o.sv + o.kh != o.H || (o.cw - 1) * o.sh + o.kw != o.W) {
    set_error("conv: padded image %dx%d is not (code-1)*stride+kernel "
              "(%dx%d); the reference raises a size mismatch here",
              o.H, o.W, (o.ch - 1) * o.sv + o.kh, (o.cw - 1) * o.sh + o.kw);
    return VTC_ERR_INVALID_ARGUMENT;
  }
  *out = o;
  return VTC_OK;
}

// utils/convolutions.py:17-24, including the `-0:` quirk: a trailing pad of 0
// blanks the whole axis.
__device__ __forceinline__ float mask_at(const ConvGeo& g, int y, int x) {
  if (!g.has_pad) return 1.f;
  const bool row_ok = (y >= g.lead_v) && (g.trail_v != 0) &&
                      (y < g.H - g.trail_v);
  const bool col_ok = (x >= g.lead_h) && (g.trail_h != 0) &&
                      (x < g.W - g.trail_h);
  return (row_ok && col_ok) ? 1.f : 0.f;
}

// ------------------------------------------------------------ synthesis
// Block: 32x32 output pixels of one (image, channel); thread: 4 pixels in x.
// Per chunk of kernels the needed code window and the kernels' taps for this
// channel are staged in LDS (window zero-filled outside the code map).
constexpr int kSynTile = 32;

struct SynPlan {
  int wy, wx;       // window extent in code rows / cols
  int ty_max, tx_max;
  int s_chunk;
  size_t lds_bytes;
};

static SynPlan plan_synthesis(const ConvGeo& g) {
  SynPlan p;
  p.ty_max = (g.kh - 1) / g.sv;
  p.tx_max = (g.kw - 1) / g.sh;
  p.wy = (kSynTile - 1) / g.sv + p.ty_max + 2;
  p.wx = (kSynTile - 1) / g.sh + p.tx_max + 2;
  const size_t per_kernel = (size_t)p.wy * p.wx + (size_t)g.kh * g.kw;
  size_t chunk = (48 * 1024 / sizeof(float)) / per_kernel;
  if (chunk < 1) chunk = 1;
  if (chunk > (size_t)g.s) chunk = g.s;
  p.s_chunk = (int)chunk;
  p.lds_bytes = chunk * per_kernel * sizeof(float);
  return p;
}

__global__ __launch_bounds__(256) void conv_synth_residual_kernel(
    const float* __restrict__ codes, const float* __restrict__ D,
    const float* __restrict__ images, float* __restrict__ residual, ConvGeo g,
    int wy, int wx, int ty_max, int tx_max, int s_chunk, int tiles_x) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Wl = lds;                                   // [s_chunk][wy][wx]
  float* Kl = lds + (size_t)s_chunk * wy * wx;       // [s_chunk][kh][kw]
  const int tid = threadIdx.x;
  const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x % tiles_x;
  const int chan = blockIdx.y;
  const int64_t img = blockIdx.z;
  const int y0 = tile_y * kSynTile, x0 = tile_x * kSynTile;
  const int p_lo = y0 / g.sv - ty_max, q_lo = x0 / g.sh - tx_max;
  const int y = y0 + (tid >> 3);
  const int xb = x0 + (tid & 7) * 4;
  const int ry = y % g.sv, py = y / g.sv - p_lo;
  int rx[4], qx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rx[j] = (xb + j) % g.sh;
    qx[j] = (xb + j) / g.sh - q_lo;
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t map = (int64_t)g.ch * g.cw;
  const int taps = g.kh * g.kw;
  for (int s0 = 0; s0 < g.s; s0 += s_chunk) {
    const int ns = (g.s - s0 < s_chunk) ? g.s - s0 : s_chunk;
    __syncthreads();
    for (int e = tid; e < ns * wy * wx; e += 256) {
      const int sl = e / (wy * wx), rem = e % (wy * wx);
      const int p = p_lo + rem / wx, q = q_lo + rem % wx;
      float v = 0.f;
      if (p >= 0 && p < g.ch && q >= 0 && q < g.cw)
        v = codes[(img * g.s + s0 + sl) * map + (int64_t)p * g.cw + q];
      Wl[e] = v;
    }
    for (int e = tid; e < ns * taps; e += 256) {
      const int sl = e / taps, tap = e % taps;
      Kl[e] = D[((int64_t)(s0 + sl) * g.c + chan) * taps + tap];
    }
    __syncthreads();
    if (y < g.H) {
      for (int sl = 0; sl < ns; ++sl) {
        const float* w = Wl + (size_t)sl * wy * wx;
        const float* k = Kl + (size_t)sl * taps;
        for (int t = 0; t <= ty_max; ++t) {
          const int dy = ry + t * g.sv;
          if (dy >= g.kh) break;
          const float* wrow = w + (py - t) * wx;
          const float* krow = k + dy * g.kw;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            for (int u = 0; u <= tx_max; ++u) {
              const int dx = rx[j] + u * g.sh;
              if (dx >= g.kw) break;
              acc[j] = fmaf(wrow[qx[j] - u], krow[dx], acc[j]);
            }
          }
        }
      }
    }
  }
  if (y < g.H) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = xb + j;
      if (x < g.W) {
        const int64_t i = ((img * g.c + chan) * g.H + y) * (int64_t)g.W + x;
        residual[i] = mul_rn(mask_at(g, y, x), sub_rn(acc[j], images[i]));
      }
    }
  }
}

// -------------------------------------------------------------- analysis
// Block: a tp x tq tile of code positions of one image (tp*tq <= 256, one
// position per thread), kernels in chunks of 16 accumulators per thread.  The
// residual window of the tile (all channels) sits in LDS; kernel taps come
// from a transposed copy Kt[(c,dy,dx)][s] so that the 16 taps a thread needs
// are 16 consecutive floats at a wave-uniform address (scalar loads).
constexpr int kAnaAcc = 16;

struct AnaPlan {
  int tp, tq, wy, wx;
  size_t lds_bytes;
};

static AnaPlan plan_analysis(const ConvGeo& g) {
  AnaPlan p;
  p.tp = 16;
  p.tq = 16;
  for (;;) {
    p.wy = (p.tp - 1) * g.sv + g.kh;
    p.wx = (p.tq - 1) * g.sh + g.kw;
    p.lds_bytes = (size_t)g.c * p.wy * p.wx * sizeof(float);
    if (p.lds_bytes <= 40 * 1024 || (p.tp == 1 && p.tq == 1)) break;
    if (p.tp >= p.tq && p.tp > 1) p.tp /= 2; else p.tq /= 2;
  }
  return p;
}

// y_out / c_out: where the two-kernel analysis routes write the new momentum
// iterate and the new codes.  Never the buffers they read: the maps are in
// the caller's (b, s, ch, cw) layout, whose rows are not multiples of 128
// bytes, so blocks on different XCDs share cache lines -- and a line that two
// XCDs both read and partly rewrite within one launch can lose one of the
// updates (the L2s are not coherent within a launch).  Lines
// that are only written merge correctly.
struct ProxParams {
  float eta, cutoff, beta;
  int mode, fista;
  double* delta_sum;
  float* y_out;
  float* c_out;
};

}  // namespace vtc
#include "conv_unit.h"
#include "conv_x3.h"
#include "conv_patch.h"
namespace vtc {

__global__ __launch_bounds__(256) void conv_analysis_prox_kernel(
    const float* __restrict__ residual, const float* __restrict__ Kt,
    float* __restrict__ Y, float* __restrict__ C, ConvGeo g, int tp, int tq,
    int wy, int wx, int tiles_q, ProxParams pp) {
  extern __shared__ __attribute__((aligned(16))) float Rl[];  // [c][wy][wx]
  const int tid = threadIdx.x;
  const int tile_p = blockIdx.x / tiles_q, tile_q = blockIdx.x % tiles_q;
  const int64_t img = blockIdx.y;
  const int p0 = tile_p * tp, q0 = tile_q * tq;
  const int y0 = p0 * g.sv, x0 = q0 * g.sh;
  for (int e = tid; e < g.c * wy * wx; e += 256) {
    const int chan = e / (wy * wx), rem = e % (wy * wx);
    const int y = y0 + rem / wx, x = x0 + rem % wx;
    float v = 0.f;
    if (y < g.H && x < g.W)
      v = residual[((img * g.c + chan) * g.H + y) * (int64_t)g.W + x];
    Rl[e] = v;
  }
  __syncthreads();
  const int lp = tid / tq, lq = tid % tq;
  const int p = p0 + lp, q = q0 + lq;
  const bool active = (tid < tp * tq) && p < g.ch && q < g.cw;
  const int taps = g.kh * g.kw;
  const int64_t map = (int64_t)g.ch * g.cw;
  double local = 0.0;
  for (int s0 = 0; s0 < g.s; s0 += kAnaAcc) {
    float acc[kAnaAcc];
#pragma unroll
    for (int i = 0; i < kAnaAcc; ++i) acc[i] = 0.f;
    if (active) {
      for (int chan = 0; chan < g.c; ++chan) {
        const float* win = Rl + ((size_t)chan * wy + lp * g.sv) * wx + lq * g.sh;
        for (int dy = 0; dy < g.kh; ++dy) {
          for (int dx = 0; dx < g.kw; ++dx) {
            const float rv = win[dy * wx + dx];
            const float* kt =
                Kt + ((int64_t)(chan * taps + dy * g.kw + dx)) * g.s + s0;
            if (s0 + kAnaAcc <= g.s) {
#pragma unroll
              for (int i = 0; i < kAnaAcc; ++i) acc[i] = fmaf(rv, kt[i], acc[i]);
            } else {
#pragma unroll
              for (int i = 0; i < kAnaAcc; ++i)
                if (s0 + i < g.s) acc[i] = fmaf(rv, kt[i], acc[i]);
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < kAnaAcc; ++i) {
        if (s0 + i >= g.s) continue;
        const int64_t idx = (img * g.s + s0 + i) * map + (int64_t)p * g.cw + q;
        const float yv = Y[idx];
        const float c = shrink(sub_rn(yv, mul_rn(pp.eta, acc[i])), pp.cutoff,
                               pp.mode);
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
  if (pp.delta_sum) {
    const double w = wave_sum(local);
    if ((tid & 63) == 0) atomicAdd(pp.delta_sum, w);
  }
}

// Kt[(c*taps + tap)][s] = D[s][c][tap]
__global__ void transpose_kernels_kernel(const float* __restrict__ D,
                                         float* __restrict__ Kt, int s,
                                         int ctaps) {
  const int64_t total = (int64_t)s * ctaps;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / s, col = i % s;  // row = (c,tap), col = kernel
    Kt[i] = D[col * ctaps + row];
  }
}

// ---------------------------------------------------------- dict gradient
// Block: one tp x tq tile of positions of one image -> partial gradient for
// every (s, c, dy, dx), written to slab [tile-slot][s*c*taps]; blocks stride
// over (image, tile) pairs so that the number of slabs stays bounded, and the
// slabs are summed in a fixed order afterwards.
// Thread <-> (c, dy, dx) tap (looping if there are more taps than threads);
// kernels in chunks of 16 accumulators.
__global__ __launch_bounds__(256) void conv_grad_kernel(
    const float* __restrict__ residual, const float* __restrict__ codes,
    float* __restrict__ slabs, ConvGeo g, int tp, int tq, int wy, int wx,
    int tiles_q, int64_t tiles_per_image, int64_t total_tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Rl = lds;                                  // [c][wy][wx]
  float* Cl = lds + (size_t)g.c * wy * wx;          // [kAnaAcc][tp*tq]
  const int tid = threadIdx.x;
  const int taps = g.kh * g.kw;
  const int ctaps = g.c * taps;
  const int64_t map = (int64_t)g.ch * g.cw;
  const int npos = tp * tq;
  float* slab = slabs + (int64_t)blockIdx.x * g.s * ctaps;
  for (int s0 = 0; s0 < g.s; s0 += kAnaAcc) {
    for (int tap0 = 0; tap0 < ctaps; tap0 += 256) {
      const int tap = tap0 + tid;
      const bool tap_ok = tap < ctaps;
      const int chan = tap_ok ? tap / taps : 0;
      const int dy = tap_ok ? (tap % taps) / g.kw : 0;
      const int dx = tap_ok ? (tap % taps) % g.kw : 0;
      float acc[kAnaAcc];
#pragma unroll
      for (int i = 0; i < kAnaAcc; ++i) acc[i] = 0.f;
      for (int64_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int64_t img = tile / tiles_per_image;
        const int trem = (int)(tile % tiles_per_image);
        const int p0 = (trem / tiles_q) * tp, q0 = (trem % tiles_q) * tq;
        const int y0 = p0 * g.sv, x0 = q0 * g.sh;
        __syncthreads();
        for (int e = tid; e < g.c * wy * wx; e += 256) {
          const int cc = e / (wy * wx), rem = e % (wy * wx);
          const int y = y0 + rem / wx, x = x0 + rem % wx;
          float v = 0.f;
          if (y < g.H && x < g.W)
            v = residual[((img * g.c + cc) * g.H + y) * (int64_t)g.W + x];
          Rl[e] = v;
        }
        for (int e = tid; e < kAnaAcc * npos; e += 256) {
          const int i = e / npos, pos = e % npos;
          const int p = p0 + pos / tq, q = q0 + pos % tq;
          float v = 0.f;
          if (s0 + i < g.s && p < g.ch && q < g.cw)
            v = codes[(img * g.s + s0 + i) * map + (int64_t)p * g.cw + q];
          Cl[e] = v;
        }
        __syncthreads();
        if (tap_ok) {
          const float* win = Rl + ((size_t)chan * wy + dy) * wx + dx;
          for (int lp = 0; lp < tp; ++lp) {
            for (int lq = 0; lq < tq; ++lq) {
              const float rv = win[lp * g.sv * wx + lq * g.sh];
              const float* cl = Cl + lp * tq + lq;
#pragma unroll
              for (int i = 0; i < kAnaAcc; ++i)
                acc[i] = fmaf(cl[i * npos], rv, acc[i]);
            }
          }
        }
      }
      if (tap_ok) {
#pragma unroll
        for (int i = 0; i < kAnaAcc; ++i)
          if (s0 + i < g.s) slab[(int64_t)(s0 + i) * ctaps + tap] = acc[i];
      }
    }
  }
}

static int grad_blocks(int64_t total_tiles) {
  return (int)(total_tiles < 512 ? total_tiles : 512);
}

// ------------------------------------------------------------ dict apply
// Single block: the dictionary is tiny (s*c*kh*kw floats).
__device__ float block_sum_1024(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float total = 0.f;
  for (int i = 0; i < 16; ++i) total += red[i];
  return total;
}

__global__ __launch_bounds__(1024) void conv_apply_kernel(
    float* __restrict__ D, const float* __restrict__ G,
    const float* __restrict__ hess, float batch_f, float stepsize, float lowest,
    int normalize, int64_t s, int64_t ke, float* __restrict__ scaled) {
  __shared__ float red[16];
  const int64_t total = s * ke;
  // g = G / b [ / (h + lowest) ]; norms of D and g
  float sq_d = 0.f, sq_g = 0.f;
  for (int64_t i = threadIdx.x; i < total; i += 1024) {
    float gv = G[i] / batch_f;
    if (hess) gv = gv / add_rn(hess[i / ke], lowest);
    scaled[i] = gv;
    sq_g = fmaf(gv, gv, sq_g);
    const float dv = D[i];
    sq_d = fmaf(dv, dv, sq_d);
  }
  const float norm_d = sqrtf(block_sum_1024(sq_d, red));
  const float norm_g = sqrtf(block_sum_1024(sq_g, red));
  const float ratio = norm_d / norm_g;  // sc_steepest_descent.py:68
  for (int64_t i = threadIdx.x; i < total; i += 1024)
    D[i] = sub_rn(D[i], mul_rn(stepsize, mul_rn(scaled[i], ratio)));
  if (!normalize) return;
  __syncthreads();
  // per-kernel l2 normalisation, one wave per kernel in turn
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t k = wave; k < s; k += 16) {
    float sq = 0.f;
    for (int64_t j = lane; j < ke; j += 64) {
      const float v = D[k * ke + j];
      sq = fmaf(v, v, sq);
    }
    const float nrm = sqrtf(wave_sum(sq));
    for (int64_t j = lane; j < ke; j += 64) D[k * ke + j] = D[k * ke + j] / nrm;
  }
}

static size_t conv_x3_image_bytes(const ConvGeo& g) {
  CxPlan p;
  return cx_plan(g, &p) ? cx_image_bytes(p) + cx_fused_bytes(p) : 0;
}

// f16x3 scale state (x3_scale.h, CxScales): {sigma_D, 1 / sigma_D} and, per
// image, two words each for max |R| and max |Y|
static size_t cx_state_words(const ConvGeo& g) {
  return 64 + 4 * (size_t)g.b;
}

static size_t conv_inference_ws(const ConvGeo& g) {
  const size_t code_elems = (size_t)g.b * g.s * g.ch * g.cw;
  const size_t img_elems = (size_t)g.b * g.c * g.H * g.W;
  return 3 * align_up(code_elems * sizeof(float), 256) +      // Y, Y', C'
         align_up(8 * img_elems * sizeof(float), 256) +       // residual or
                                                              // <= 8 partials
         align_up((size_t)g.s * g.c * g.kh * g.kw * 4, 256) + // Kt
         conv_x3_image_bytes(g) +                             // bf16x3 operands
         patch_workspace_bytes(g) +                           // im2col, Q
         align_up(cx_state_words(g) * sizeof(unsigned), 256) + // f16x3 scales
         256;
}

static int launch_synthesis(const float* codes, const float* D,
                            const float* images, float* residual,
                            const ConvGeo& g, hipStream_t st) {
  if (unit_geometry(g))
    return launch_synth_unit(codes, D, images, residual, g, 1, g.s, st);
  const SynPlan sp = plan_synthesis(g);
  const int tiles_x = (int)ceil_div(g.W, kSynTile);
  const int tiles_y = (int)ceil_div(g.H, kSynTile);
  if (sp.lds_bytes > 64 * 1024)
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(conv_synth_residual_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.lds_bytes));
  hipLaunchKernelGGL(conv_synth_residual_kernel,
                     dim3((unsigned)(tiles_x * tiles_y), (unsigned)g.c,
                          (unsigned)g.b),
                     dim3(256), sp.lds_bytes, st, codes, D, images, residual,
                     g, sp.wy, sp.wx, sp.ty_max, sp.tx_max, sp.s_chunk,
                     tiles_x);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

}  // namespace vtc

using namespace vtc;

extern "C" int vtc_conv_code_dims(const vtc_conv_geometry* g, int32_t* code_h,
                                  int32_t* code_w) {
  VTC_REQUIRE(g && code_h && code_w, "vtc_conv_code_dims: null pointer");
  VTC_REQUIRE(g->stride_v > 0 && g->stride_h > 0, "vtc_conv_code_dims: stride");
  *code_h = code_dim(g->h, g->kh, g->stride_v);
  *code_w = code_dim(g->w, g->kw, g->stride_h);
  return VTC_OK;
}

extern "C" int vtc_conv_x3_supported(const vtc_conv_geometry* geom) {
  ConvGeo g;
  CxPlan p;
  if (!geom || make_geo(geom, &g) != VTC_OK) return 0;
  return cx_plan(g, &p) ? 1 : 0;
}

extern "C" size_t vtc_conv_ista_fista_workspace_bytes(
    const vtc_conv_geometry* geom) {
  ConvGeo g;
  if (make_geo(geom, &g) != VTC_OK || g.b == 0) return 256;
  return conv_inference_ws(g);
}

extern "C" int vtc_conv_ista_fista(
    const float* images_padded, const float* dictionary,
    const float* initial_codes, float* codes, const vtc_conv_geometry* geom,
    float stepsize, float sparsity_weight, int num_iters, int variant,
    int threshold, float early_stopping_epsilon, int precision,
    void* workspace, size_t workspace_bytes, int* iters_run, void* stream) {
  VTC_REQUIRE((geom && geom->b == 0) || (images_padded && dictionary && codes),
              "vtc_conv_ista_fista: null pointer");
  VTC_REQUIRE(precision == VTC_F32 || precision == VTC_BF16X3 ||
                  precision == VTC_F16X3,
              "vtc_conv_ista_fista: precision must be VTC_F32, VTC_F16X3 or "
              "VTC_BF16X3");
  ConvGeo g;
  int rc = make_geo(geom, &g);
  if (rc != VTC_OK) return rc;
  CxPlan xp;
  const bool x3 = (precision == VTC_BF16X3 || precision == VTC_F16X3);
  const bool f16 = (precision == VTC_F16X3);
  if (x3 && !cx_plan(g, &xp)) {
    set_error("vtc_conv_ista_fista: the split modes cover stride 1 and square "
              "kernels of 5, 8, 11 or 16 whose planes fit the LDS (see "
              "vtc_conv_x3_supported)");
    return VTC_ERR_UNSUPPORTED;
  }
  VTC_REQUIRE(variant == VTC_ISTA || variant == VTC_FISTA,
              "vtc_conv_ista_fista: variant must be ista or fista");
  VTC_REQUIRE(threshold >= VTC_SOFT && threshold <= VTC_HARD_NONNEG,
              "vtc_conv_ista_fista: unknown threshold mode");
  VTC_REQUIRE(num_iters >= 1, "vtc_conv_ista_fista: num_iters >= 1");
  if (iters_run) *iters_run = 0;
  if (g.b == 0) return VTC_OK;
  if (!workspace || workspace_bytes < conv_inference_ws(g)) {
    set_error("vtc_conv_ista_fista: workspace too small");
    return VTC_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const size_t code_elems = (size_t)g.b * g.s * g.ch * g.cw;
  const size_t img_elems = (size_t)g.b * g.c * g.H * g.W;
  const int ctaps = g.c * g.kh * g.kw;
  Carver ws(workspace);
  float* Ybuf = ws.take<float>(code_elems);
  float* Yalt = ws.take<float>(code_elems);   // out-of-place targets of the
  float* Calt = ws.take<float>(code_elems);   // two-kernel routes (ProxParams)
  float* residual = ws.take<float>(8 * img_elems);
  float* Kt = ws.take<float>((size_t)g.s * ctaps);
  uint16_t* syn_image = nullptr;
  uint16_t* ana_image = nullptr;
  uint16_t* synp_image = nullptr;   // fused iteration kernel (conv_x3.h)
  float* partial = nullptr;
  float* Cfrag1 = nullptr;          // the last two code iterates of that
  float* Cfrag0 = nullptr;          // kernel, fragment order (CxMaps)
  // f16x3: power-of-two scales of the operands (conv_x3.h, CxScales)
  unsigned* state = ws.take<unsigned>(cx_state_words(g));
  float* dscale = f16 ? reinterpret_cast<float*>(state) : nullptr;
  unsigned* r_slot[2] = {state + 64, state + 64 + g.b};
  unsigned* y_slot[2] = {state + 64 + 2 * g.b, state + 64 + 3 * g.b};
  const int images = (int)g.b;
  if (f16) {
    VTC_HIP_CHECK(hipMemsetAsync(state, 0,
                                 cx_state_words(g) * sizeof(unsigned), st));
    hipLaunchKernelGGL(cx_array_scale_kernel, dim3(1), dim3(1024), 0, st,
                       dictionary, (int64_t)g.s * g.c * g.kh * g.kw, dscale);
    VTC_LAUNCH_CHECK();
    if (initial_codes) {
      hipLaunchKernelGGL(cx_image_max_kernel, dim3(64, (unsigned)g.b),
                         dim3(256), 0, st, initial_codes,
                         (int64_t)g.s * g.ch * g.cw, y_slot[0]);
      VTC_LAUNCH_CHECK();
    }
  }
  if (x3) {
    syn_image = ws.take<uint16_t>(xp.syn_image_bytes / 2);
    ana_image = ws.take<uint16_t>(xp.ana_image_bytes / 2);
    rc = cx_pack(dictionary, g, xp, syn_image, ana_image, dscale, st);
    // fused iteration kernel (kernels up to 11x11, more than 32 of them)
    static const bool no_fused = getenv("VTC_CONV_NO_FUSED") != nullptr;
    if (rc == VTC_OK && xp.fused_lds != 0 && !no_fused) {
      synp_image = ws.take<uint16_t>(xp.synp_image_bytes / 2);
      partial = ws.take<float>(xp.partial_bytes / sizeof(float));
      Cfrag1 = ws.take<float>(xp.padded_bytes / sizeof(float));
      Cfrag0 = ws.take<float>(xp.padded_bytes / sizeof(float));
      if (f16)
        hipLaunchKernelGGL(conv_x3_pack_synp_kernel<true>, dim3(256),
                           dim3(256), 0, st, dictionary, synp_image, g.s, xp.k,
                           xp.slots, xp.chunks, dscale);
      else
        hipLaunchKernelGGL(conv_x3_pack_synp_kernel<false>, dim3(256),
                           dim3(256), 0, st, dictionary, synp_image, g.s, xp.k,
                           xp.slots, xp.chunks, dscale);
      VTC_LAUNCH_CHECK();
    }
    if (rc != VTC_OK) return rc;
  }
  const bool patch_path = !x3 && patch_geometry(g);
  float* patches = nullptr;
  float* contributions = nullptr;
  if (patch_path) {
    const size_t elems = (size_t)g.b * g.ch * g.cw * ctaps;
    patches = ws.take<float>(elems);
    contributions = ws.take<float>(elems);
  }
  double* delta_sum = ws.take<double>(1);
  const bool fista = (variant == VTC_FISTA);
  float* Y = fista ? Ybuf : codes;
  const size_t code_bytes = code_elems * sizeof(float);
  if (initial_codes) {
    VTC_HIP_CHECK(hipMemcpyAsync(codes, initial_codes, code_bytes,
                                 hipMemcpyDeviceToDevice, st));
    if (fista)
      VTC_HIP_CHECK(hipMemcpyAsync(Y, initial_codes, code_bytes,
                                   hipMemcpyDeviceToDevice, st));
  } else {
    VTC_HIP_CHECK(hipMemsetAsync(codes, 0, code_bytes, st));
    if (fista) VTC_HIP_CHECK(hipMemsetAsync(Y, 0, code_bytes, st));
  }
  hipLaunchKernelGGL(transpose_kernels_kernel,
                     dim3((unsigned)ceil_div((int64_t)g.s * ctaps, 256)),
                     dim3(256), 0, st, dictionary, Kt, g.s, ctaps);
  VTC_LAUNCH_CHECK();

  const AnaPlan ap = plan_analysis(g);
  const int tiles_p = (int)ceil_div(g.ch, ap.tp);
  const int tiles_q = (int)ceil_div(g.cw, ap.tq);
  const bool unit_path = unit_geometry(g) && unit_analysis_fits(g);
  int syn_per_group = g.s;
  const int syn_groups =
      unit_path ? unit_groups(g, (int)(ceil_div(g.W, kUnitTX) *
                                       ceil_div(g.H, kUnitTY)),
                              &syn_per_group)
                : 1;
  const float eta = stepsize;
  const float cutoff = sparsity_weight * stepsize;
  const float eps = early_stopping_epsilon;
  std::vector<float> betas;
  fista_betas(num_iters, &betas);
  int done = 0;
  // two-kernel routes: (Y, C) -> (y_out, c_out), buffers swapped per iteration
  float* Cin = codes;
  float* Cout = Calt;
  float* Yout = Yalt;
  const float* frag_latest = nullptr;   // fused path: newest codes (fragments)
  for (int k = 0; k < num_iters; ++k) {
    if (eps >= 0.f)
      VTC_HIP_CHECK(hipMemsetAsync(delta_sum, 0, sizeof(double), st));
    if (!fista) Y = Cin;                        // ISTA iterates on the codes
    ProxParams pp{eta, cutoff, fista ? betas[k] : 0.f, threshold,
                  fista ? 1 : 0, eps >= 0.f ? delta_sum : nullptr,
                  Yout, Cout};
    if (x3 && synp_image) {
      // synthesis fused into the analysis epilogue: the residual of the first
      // iteration from the stand-alone synthesis (on the caller's layout),
      // the later ones from the partial tiles of the previous fused launch.
      // Between the launches the last two code iterates live in fragment
      // order (conv_x3.h, CxMaps; Y is recomputed from them); the last launch
      // writes the codes in the caller's layout.
      if (k == 0) {
        CxScales first{dscale, nullptr, r_slot[0], nullptr, y_slot[0], nullptr,
                       nullptr, images};
        rc = cx_launch_synth(codes, syn_image, images_padded, residual, g, xp,
                             first, st);
        if (rc != VTC_OK) return rc;
        VTC_HIP_CHECK(hipMemsetAsync(Cfrag1, 0, xp.padded_bytes, st));
        VTC_HIP_CHECK(hipMemsetAsync(Cfrag0, 0, xp.padded_bytes, st));
        if (initial_codes) {
          const int tu = (int)ceil_div(g.ch, 8), tv = (int)ceil_div(g.cw, 32);
          hipLaunchKernelGGL(conv_to_fragments_kernel, dim3(4096), dim3(256),
                             0, st, initial_codes, Cfrag0, g, tu, tv, xp.chunks,
                             8, 32);
          VTC_LAUNCH_CHECK();
          VTC_HIP_CHECK(hipMemcpyAsync(Cfrag1, Cfrag0, xp.padded_bytes,
                                       hipMemcpyDeviceToDevice, st));
        }
      }
      // with early stopping the last launch is not known in advance: the
      // codes are converted when the loop ends
      const bool last = k + 1 == num_iters && eps < 0.f;
      CxMaps maps;
      maps.cur = (k & 1) ? Cfrag1 : Cfrag0;          // c_k
      maps.old = (k & 1) ? Cfrag0 : Cfrag1;          // c_(k-1) in, c_(k+1) out
      maps.user_codes = last ? codes : nullptr;
      maps.beta_prev = (fista && k > 0) ? betas[k - 1] : 0.f;
      // the launch reads max |R_k| and clears the slot into which the
      // residual kernel behind it leaves max |R_(k+1)|
      CxScales sc{dscale, r_slot[k & 1], r_slot[(k + 1) & 1],
                  r_slot[(k + 1) & 1], nullptr, nullptr, nullptr, images};
      rc = cx_launch_fused(residual, ana_image, synp_image, maps, partial,
                           images_padded, residual, g, xp, pp,
                           k + 1 < num_iters, sc, st);
      if (rc != VTC_OK) return rc;
      frag_latest = maps.old;
    } else if (x3) {
      // synthesis: reads max |Y_k|, leaves max |R_k|, clears the slot of
      // max |Y_(k+1)|; analysis: reads max |R_k|, leaves max |Y_(k+1)|, clears
      // the slot of max |R_(k+1)|
      CxScales syn_sc{dscale, nullptr, r_slot[k & 1], nullptr, y_slot[k & 1],
                      nullptr, y_slot[(k + 1) & 1], images};
      rc = cx_launch_synth(Y, syn_image, images_padded, residual, g, xp,
                           syn_sc, st);
      if (rc != VTC_OK) return rc;
      CxScales ana_sc{dscale, r_slot[k & 1], nullptr, r_slot[(k + 1) & 1],
                      nullptr, y_slot[(k + 1) & 1], nullptr, images};
      rc = cx_launch_analysis(residual, ana_image, Y, Cin, g, xp, pp, ana_sc,
                              st);
      if (rc != VTC_OK) return rc;
    } else if (patch_path) {
      // strides > 1: both convolutions as exact-f32 patch contractions
      rc = patch_synthesis(Y, dictionary, images_padded, residual,
                           contributions, g, st);
      if (rc != VTC_OK) return rc;
      rc = patch_analysis(residual, dictionary, Y, Cin, patches, g, pp, st);
      if (rc != VTC_OK) return rc;
    } else if (unit_path) {
      // stride-1 square kernels: scalar-tap kernels, the kernel sum of the
      // synthesis split over `syn_groups` blocks per tile
      rc = launch_synth_unit(Y, dictionary, images_padded, residual, g,
                             syn_groups, syn_per_group, st);
      if (rc != VTC_OK) return rc;
      rc = launch_analysis_unit(residual, images_padded, syn_groups,
                                dictionary, Y, Cin, g, pp, st);
      if (rc != VTC_OK) return rc;
    } else {
      rc = launch_synthesis(Y, dictionary, images_padded, residual, g, st);
      if (rc != VTC_OK) return rc;
      hipLaunchKernelGGL(conv_analysis_prox_kernel,
                         dim3((unsigned)(tiles_p * tiles_q), (unsigned)g.b),
                         dim3(256), ap.lds_bytes, st, residual, Kt, Y, Cin,
                         g, ap.tp, ap.tq, ap.wy, ap.wx, tiles_q, pp);
      VTC_LAUNCH_CHECK();
    }
    if (!(x3 && synp_image)) {
      // what was written becomes what is read
      float* t = Cin; Cin = Cout; Cout = t;
      if (fista) { t = Y; Y = Yout; Yout = t; }
    }
    done = k + 1;
    if (eps >= 0.f) {
      double total = 0.0;
      VTC_HIP_CHECK(hipMemcpyAsync(&total, delta_sum, sizeof(double),
                                   hipMemcpyDeviceToHost, st));
      VTC_HIP_CHECK(hipStreamSynchronize(st));
      const float mean = (float)(total / (double)code_elems);
      if (mean < eps && k > 0) break;
    }
  }
  if (x3 && synp_image && eps >= 0.f) {
    const int tu = (int)ceil_div(g.ch, 8), tv = (int)ceil_div(g.cw, 32);
    hipLaunchKernelGGL(conv_from_fragments_kernel, dim3(4096), dim3(256), 0,
                       st, frag_latest, codes, g, tu, tv, xp.chunks, 8, 32);
    VTC_LAUNCH_CHECK();
  } else if (Cin != codes) {
    VTC_HIP_CHECK(hipMemcpyAsync(codes, Cin, code_bytes,
                                 hipMemcpyDeviceToDevice, st));
  }
  if (iters_run) *iters_run = done;
  return VTC_OK;
}

// residual = mask * (conv_transpose2d(codes, D) - images_padded): the masked
// reconstruction error (the trainer's validation metrics crop the padding,
// training/sparse_coding.py:185-195, which is what the mask zeroes).
extern "C" int vtc_conv_residual(const float* images_padded,
                                 const float* dictionary, const float* codes,
                                 float* residual,
                                 const vtc_conv_geometry* geom, void* stream) {
  VTC_REQUIRE(images_padded && dictionary && codes && residual,
              "vtc_conv_residual: null pointer");
  ConvGeo g;
  int rc = make_geo(geom, &g);
  if (rc != VTC_OK) return rc;
  if (g.b == 0) return VTC_OK;
  return launch_synthesis(codes, dictionary, images_padded, residual, g,
                          as_stream(stream));
}

extern "C" size_t vtc_conv_dict_gradient_workspace_bytes(
    const vtc_conv_geometry* geom) {
  ConvGeo g;
  if (make_geo(geom, &g) != VTC_OK || g.b == 0) return 256;
  const AnaPlan ap = plan_analysis(g);
  const int64_t tiles =
      g.b * ceil_div(g.ch, ap.tp) * ceil_div(g.cw, ap.tq);
  size_t slab_count = (size_t)grad_blocks(tiles);
  if (patch_gradient_small(g) && slab_count < (size_t)g.b) slab_count = g.b;
  size_t extra = 0;
  CxPlan xp;
  if (cx_plan(g, &xp)) {                            // bf16x3 route
    if ((size_t)cx_grad_blocks(g) > slab_count) slab_count = cx_grad_blocks(g);
    extra = cx_image_bytes(xp);
  }
  // strided geometries: the residual comes from the patch contraction
  // (conv_patch.h), which needs its per-position contributions Q
  if (patch_geometry(g))
    extra += align_up((size_t)g.b * g.ch * g.cw * g.c * g.kh * g.kw *
                          sizeof(float), 256);
  return align_up((size_t)g.b * g.c * g.H * g.W * sizeof(float), 256) +
         align_up(slab_count * g.s * g.c * g.kh * g.kw * sizeof(float), 256) +
         extra;
}

extern "C" int vtc_conv_dict_gradient(const float* images_padded,
                                      const float* dictionary,
                                      const float* codes, float* grad_sum,
                                      const vtc_conv_geometry* geom,
                                      int precision, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  VTC_REQUIRE(images_padded && dictionary && codes && grad_sum,
              "vtc_conv_dict_gradient: null pointer");
  VTC_REQUIRE(precision == VTC_F32 || precision == VTC_BF16X3 ||
                  precision == VTC_F16X3,
              "vtc_conv_dict_gradient: precision must be VTC_F32, VTC_F16X3 "
              "or VTC_BF16X3");
  ConvGeo g;
  int rc = make_geo(geom, &g);
  if (rc != VTC_OK) return rc;
  VTC_REQUIRE(g.b > 0, "vtc_conv_dict_gradient: empty batch");
  CxPlan xp;
  const bool x3 = (precision == VTC_BF16X3 || precision == VTC_F16X3);
  if (x3 && !cx_plan(g, &xp)) {
    set_error("vtc_conv_dict_gradient: no bf16x3 route for this geometry "
              "(see vtc_conv_x3_supported)");
    return VTC_ERR_UNSUPPORTED;
  }
  if (!workspace ||
      workspace_bytes < vtc_conv_dict_gradient_workspace_bytes(geom)) {
    set_error("vtc_conv_dict_gradient: workspace too small");
    return VTC_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const AnaPlan ap = plan_analysis(g);
  const int tiles_q = (int)ceil_div(g.cw, ap.tq);
  const int64_t tiles_per_image = ceil_div(g.ch, ap.tp) * tiles_q;
  const int64_t total_tiles = g.b * tiles_per_image;
  const int blocks = grad_blocks(total_tiles);
  const int64_t dict_elems = (int64_t)g.s * g.c * g.kh * g.kw;
  Carver ws(workspace);
  float* residual = ws.take<float>((size_t)g.b * g.c * g.H * g.W);
  if (x3) {
    // residual and gradient on the matrix cores (conv_x3.h); the slab sum and
    // everything after it are the same as on the f32 route
    const int xblocks = cx_grad_blocks(g);
    float* xslabs = ws.take<float>((size_t)(xblocks > blocks ? xblocks
                                                              : blocks) *
                                   dict_elems);
    uint16_t* syn_image = ws.take<uint16_t>(xp.syn_image_bytes / 2);
    uint16_t* ana_image = ws.take<uint16_t>(xp.ana_image_bytes / 2);
    // (the gradient stays on the bf16 split: a single product, no iteration
    // to amplify its 2^-17, and its tests hold 5e-6 on the updated kernels)
    const CxScales none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                        nullptr, 0};
    rc = cx_pack(dictionary, g, xp, syn_image, ana_image, nullptr, st);
    if (rc != VTC_OK) return rc;
    rc = cx_launch_synth(codes, syn_image, images_padded, residual, g, xp,
                         none, st);
    if (rc != VTC_OK) return rc;
    rc = cx_launch_grad(residual, codes, xslabs, g, xp, st);
    if (rc != VTC_OK) return rc;
    return launch_slab_reduce(xslabs, xblocks, dict_elems, grad_sum, st);
  }
  float* slabs = ws.take<float>(
      (size_t)(patch_gradient_small(g) && blocks < g.b ? g.b : blocks) *
      dict_elems);
  if (patch_geometry(g)) {
    // (the direct synthesis kernel spends ~30 integer instructions per FMA on
    // strided geometries: 135 us against 19 at the reference's example size)
    float* Q = ws.take<float>((size_t)g.b * g.ch * g.cw * g.c * g.kh * g.kw);
    rc = patch_synthesis(codes, dictionary, images_padded, residual, Q, g, st);
  } else {
    rc = launch_synthesis(codes, dictionary, images_padded, residual, g, st);
  }
  if (rc != VTC_OK) return rc;
  if (patch_gradient_small(g)) {
    // one slab per image (the workspace holds at least b of them)
    rc = patch_gradient_slabs(residual, codes, slabs, g, st);
    if (rc != VTC_OK) return rc;
    return launch_slab_reduce(slabs, (int)g.b, dict_elems, grad_sum, st);
  }
  const size_t lds = ap.lds_bytes +
                     (size_t)kAnaAcc * ap.tp * ap.tq * sizeof(float);
  hipLaunchKernelGGL(conv_grad_kernel, dim3((unsigned)blocks), dim3(256), lds,
                     st, residual, codes, slabs, g, ap.tp, ap.tq, ap.wy, ap.wx,
                     tiles_q, tiles_per_image, total_tiles);
  VTC_LAUNCH_CHECK();
  return launch_slab_reduce(slabs, blocks, dict_elems, grad_sum, st);
}

extern "C" int vtc_conv_dict_apply(float* dictionary, const float* grad_sum,
                                   const float* hessian_diagonal,
                                   int64_t global_batch, float stepsize,
                                   float lowest_code_val, int normalize,
                                   int64_t s, int64_t kernel_elems,
                                   float* scratch, void* stream) {
  VTC_REQUIRE(dictionary && grad_sum && scratch,
              "vtc_conv_dict_apply: null pointer");
  VTC_REQUIRE(s > 0 && kernel_elems > 0 && global_batch > 0,
              "vtc_conv_dict_apply: bad sizes");
  hipLaunchKernelGGL(conv_apply_kernel, dim3(1), dim3(1024), 0,
                     as_stream(stream), dictionary, grad_sum, hessian_diagonal,
                     (float)global_batch, stepsize, lowest_code_val, normalize,
                     s, kernel_elems, scratch);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}
