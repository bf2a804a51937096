// Strided convolutions as patch contractions (the reference's own conv
// geometry: 16x16 kernels, stride 8, examples/train_convolutional_sparse_
// coding.py:25-39; its tests use the same, tests/ista_fista_2.py:16-24).
//
// With stride s and kernel k every pixel is covered by at most
// ceil(kh/sv) * ceil(kw/sh) code positions (4 for k = 2 s), so an im2col copy
// of the residual is only that many times the image and both convolutions of
// analysis_transforms/convolutional/ista_fista.py:152-155 become plain
// exact-f32 MFMA contractions (gemm_f32.h) over "patches":
//
//   synthesis  Q[pos, t]  = sum_s Y[s, pos] D[s, t]          per image
//              recon[y,x] = sum over the covering positions of Q[pos, t(y,x)]
//              residual   = mask * (recon - X)                (col2im kernel)
//   analysis   P[pos, t]  = residual[pos * stride + t]        (im2col kernel)
//              G[s, pos]  = sum_t D[s, t] P[pos, t]   + proximal epilogue
//
// pos = (image, p, q), t = (channel, dy, dx).  The direct kernels of conv.hip
// spend ~30 integer instructions per FMA on this geometry and give the
// analysis 45 blocks for the whole chip: 366 us per iteration against ~40 us
// here for the example geometry (b = 5).  Fixed summation orders throughout.
#pragma once

namespace vtc {

static bool patch_geometry(const ConvGeo& g) {
  const int64_t cover = (int64_t)ceil_div(g.kh, g.sv) * ceil_div(g.kw, g.sh);
  const int64_t ctaps = (int64_t)g.c * g.kh * g.kw;
  return (g.sv > 1 || g.sh > 1) && cover <= 16 && ctaps <= 8192 &&
         g.b <= 65535 &&
         (int64_t)g.b * g.ch * g.cw < ((int64_t)1 << 31);  // 32-bit positions
}

static size_t patch_workspace_bytes(const ConvGeo& g) {
  if (!patch_geometry(g)) return 0;
  const size_t elems = (size_t)g.b * g.ch * g.cw * g.c * g.kh * g.kw;
  return 2 * align_up(elems * sizeof(float), 256);   // P and Q
}

// P[pos][t] = R[img][c][p*sv + dy][q*sh + dx]
__global__ void conv_im2col_kernel(const float* __restrict__ R,
                                   float* __restrict__ P, ConvGeo g) {
  const int ctaps = g.c * g.kh * g.kw;
  const int64_t map = (int64_t)g.ch * g.cw;
  const int64_t total = g.b * map * ctaps;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (total < ((int64_t)1 << 31)) {
    // 32-bit index arithmetic (five 64-bit divisions per element were a fifth
    // of an iteration at the reference's example geometry)
    const unsigned cw = (unsigned)g.cw, ch = (unsigned)g.ch;
    const unsigned kw = (unsigned)g.kw, kh = (unsigned)g.kh;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
         i < (unsigned)total; i += (unsigned)stride) {
      const unsigned pos = i / (unsigned)ctaps, t = i - pos * (unsigned)ctaps;
      const unsigned dx = t % kw, dy = (t / kw) % kh, chan = t / (kw * kh);
      const unsigned q = pos % cw, pc = pos / cw;
      const unsigned p = pc % ch, img = pc / ch;
      P[i] = R[(((int64_t)img * g.c + chan) * g.H + p * g.sv + dy) *
                   (int64_t)g.W + q * g.sh + dx];
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += stride) {
    const int t = (int)(i % ctaps);
    const int64_t pos = i / ctaps;
    const int dx = t % g.kw, dy = (t / g.kw) % g.kh, chan = t / (g.kw * g.kh);
    const int q = (int)(pos % g.cw), p = (int)((pos / g.cw) % g.ch);
    const int64_t img = pos / map;
    P[i] = R[((img * g.c + chan) * g.H + p * g.sv + dy) * (int64_t)g.W +
             q * g.sh + dx];
  }
}

// residual[img][c][y][x] = mask * (sum_{covering (p,q), ascending} Q - X)
__global__ void conv_col2im_residual_kernel(const float* __restrict__ Q,
                                            const float* __restrict__ X,
                                            float* __restrict__ R, ConvGeo g) {
  const int ctaps = g.c * g.kh * g.kw;
  const int64_t total = g.b * g.c * (int64_t)g.H * g.W;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += stride) {
    int x, y, chan;
    int64_t img;
    if (total < ((int64_t)1 << 31)) {               // 32-bit index arithmetic
      const unsigned i32 = (unsigned)i, W = (unsigned)g.W, H = (unsigned)g.H;
      const unsigned row = i32 / W, plane = row / H;
      x = (int)(i32 - row * W);
      y = (int)(row - plane * H);
      chan = (int)(plane % (unsigned)g.c);
      img = plane / (unsigned)g.c;
    } else {
      x = (int)(i % g.W);
      y = (int)((i / g.W) % g.H);
      chan = (int)((i / ((int64_t)g.W * g.H)) % g.c);
      img = i / ((int64_t)g.W * g.H * g.c);
    }
    int p_lo = (y - g.kh + g.sv) / g.sv;          // ceil((y - kh + 1) / sv)
    if (y - g.kh + 1 <= 0) p_lo = 0;
    int p_hi = y / g.sv;
    if (p_hi > g.ch - 1) p_hi = g.ch - 1;
    int q_lo = (x - g.kw + g.sh) / g.sh;
    if (x - g.kw + 1 <= 0) q_lo = 0;
    int q_hi = x / g.sh;
    if (q_hi > g.cw - 1) q_hi = g.cw - 1;
    float sum = 0.f;
    for (int p = p_lo; p <= p_hi; ++p)
      for (int q = q_lo; q <= q_hi; ++q) {
        const int t = (chan * g.kh + (y - p * g.sv)) * g.kw + (x - q * g.sh);
        sum = add_rn(sum, Q[((img * g.ch + p) * g.cw + q) * (int64_t)ctaps + t]);
      }
    R[i] = mul_rn(mask_at(g, y, x), sub_rn(sum, X[i]));
  }
}

// Epilogue of the analysis contraction: row = kernel, column = position
// (image, p, q) -- lanes run along q, the contiguous axis of the code maps.
struct EpiPatchProx {
  float* Y;
  float* C;
  int64_t s, map;
  ProxParams pp;
  double local;
  // two-phase protocol of gemm_f32.h: Y and the previous codes are read
  // before the K loop, nothing waits on memory afterwards
  static constexpr bool kElemFetch = true;
  struct Fetched {
    float yv, cv;
  };
  __device__ __forceinline__ int64_t index_of(int64_t row, int64_t col) const {
    // (positions fit 32 bits -- patch_geometry limits the route to small
    // maps -- and a 64-bit division per output element cost more than the
    // proximal step itself)
    const unsigned img32 = (unsigned)col / (unsigned)map;
    const int64_t img = img32, pq = col - img * map;
    return (img * s + row) * map + pq;
  }
  __device__ __forceinline__ Fetched fetch(int64_t row, int64_t col) const {
    const int64_t idx = index_of(row, col);
    Fetched f;
    f.yv = Y[idx];
    f.cv = pp.fista ? C[idx] : 0.f;
    return f;
  }
  __device__ __forceinline__ void apply(int64_t row, int64_t col, float v, int,
                                        const Fetched& f) {
    const int64_t idx = index_of(row, col);
    const float c = shrink(sub_rn(f.yv, mul_rn(pp.eta, v)), pp.cutoff, pp.mode);
    float d;
    if (pp.fista) {
      d = sub_rn(c, f.cv);
      pp.y_out[idx] = add_rn(c, mul_rn(pp.beta, d));
    } else {
      d = sub_rn(c, f.yv);
    }
    pp.c_out[idx] = c;
    if (pp.delta_sum) local += (double)(fabsf(d) / pp.eta);
  }
  __device__ __forceinline__ void block_end() const {
    if (pp.delta_sum) {
      const double w = wave_sum(local);
      if ((threadIdx.x & 63) == 0) atomicAdd(pp.delta_sum, w);
    }
  }
};

// The im2col view as a B operand source of gemm_f32_small_kernel: element
// (pos, t) = R[img][chan][p*sv + dy][q*sh + dx], read in place.
struct BIm2col {
  const float* R;
  int c, H, W, kh, kw, sv, sh, ch, cw;
  int vec;   // 8 consecutive t of a lane are 8 consecutive, 16-byte aligned
             // pixels (kw % 8 == 0, strides and rows multiples of 4 floats)
  __device__ __forceinline__ const float* row(int64_t pos) const {
    const unsigned p32 = (unsigned)pos;
    const unsigned q = p32 % (unsigned)cw, pc = p32 / (unsigned)cw;
    const unsigned p = pc % (unsigned)ch, img = pc / (unsigned)ch;
    return R + ((int64_t)img * c * H + (int64_t)p * sv) * W + q * sh;
  }
  __device__ __forceinline__ const float* at(const float* base,
                                             int64_t k) const {
    const unsigned t = (unsigned)k;
    const unsigned dx = t % (unsigned)kw, rest = t / (unsigned)kw;
    const unsigned dy = rest % (unsigned)kh, chan = rest / (unsigned)kh;
    return base + ((int64_t)chan * H + dy) * W + dx;
  }
};

// The transposed view for the dictionary gradient of ONE image: element
// (tap t, position k) = R_img[chan][p*sv + dy][q*sh + dx], k = p*cw + q.
struct BIm2colT {
  const float* Rimg;
  int H, W, kh, kw, sv, sh, cw;
  int vec;   // 0: consecutive positions are `sh` pixels apart
  __device__ __forceinline__ const float* row(int64_t tap) const {
    const unsigned t = (unsigned)tap;
    const unsigned dx = t % (unsigned)kw, rest = t / (unsigned)kw;
    const unsigned dy = rest % (unsigned)kh, chan = rest / (unsigned)kh;
    return Rimg + ((int64_t)chan * H + dy) * W + dx;
  }
  __device__ __forceinline__ const float* at(const float* base,
                                             int64_t k) const {
    const unsigned k32 = (unsigned)k;
    const unsigned p = k32 / (unsigned)cw, q = k32 - p * (unsigned)cw;
    return base + (int64_t)(p * sv) * W + q * sh;
  }
};

// few images, small dictionary: the gradient per image on the 32x32-tile kernel
static bool patch_gradient_small(const ConvGeo& g) {
  return patch_geometry(g) && g.b <= 16 &&
         gemm_prefers_small(g.s, (int64_t)g.c * g.kh * g.kw);
}

// slabs[img][s][t] = sum_pos C[img][s][pos] * P[img][pos][t], the patches read
// in place from the residual
static int patch_gradient_slabs(const float* residual, const float* codes,
                                float* slabs, const ConvGeo& g,
                                hipStream_t st) {
  const int64_t map = (int64_t)g.ch * g.cw;
  const int64_t ctaps = (int64_t)g.c * g.kh * g.kw;
  for (int64_t img = 0; img < g.b; ++img) {
    BIm2colT src{residual + img * g.c * g.H * (int64_t)g.W, g.H, g.W, g.kh,
                 g.kw, g.sv, g.sh, g.cw, 0};
    EpiStore e{slabs + img * g.s * ctaps, ctaps};
    const int rc = launch_gemm_f32_small_mapped(codes + img * g.s * map, map,
                                                (int64_t)g.s, ctaps, map, e,
                                                src, st);
    if (rc != VTC_OK) return rc;
  }
  return VTC_OK;
}

static unsigned patch_grid(int64_t total) {
  int64_t blocks = ceil_div(total, 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// residual = mask * (conv_transpose2d(Y, D) - X)
static int patch_synthesis(const float* Y, const float* D, const float* X,
                           float* residual, float* Q, const ConvGeo& g,
                           hipStream_t st) {
  const int64_t map = (int64_t)g.ch * g.cw;
  const int64_t ctaps = (int64_t)g.c * g.kh * g.kw;
  // per image: Q[pos, t] = sum_s Y[s, pos] D[s, t];  A = Y_img stored [K][M]
  EpiStore e{Q, ctaps};
  int rc = launch_gemm_f32<false, false>(Y, map, D, ctaps, map, ctaps, g.s, 1,
                                         e, st, g.b, (int64_t)g.s * map, 0);
  if (rc != VTC_OK) return rc;
  hipLaunchKernelGGL(conv_col2im_residual_kernel,
                     dim3(patch_grid(g.b * g.c * (int64_t)g.H * g.W)),
                     dim3(256), 0, st, Q, X, residual, g);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

// gradient step + threshold + extrapolation from the residual
static int patch_analysis(const float* residual, const float* D, float* Y,
                          float* C, float* P, const ConvGeo& g,
                          const ProxParams& pp, hipStream_t st) {
  const int64_t map = (int64_t)g.ch * g.cw;
  const int64_t ctaps = (int64_t)g.c * g.kh * g.kw;
  EpiPatchProx e{Y, C, g.s, map, pp, 0.0};
  // Few tiles (the reference's example: 64 kernels, 5 images): the 32x32-tile
  // kernel reads the patches straight from the residual image -- no im2col
  // pass, no patch matrix in memory (9.5 of 43 us per iteration there).
  {
    BIm2col src{residual, g.c, g.H, g.W, g.kh, g.kw, g.sv, g.sh, g.ch, g.cw, 0};
    src.vec = (g.kw % 8 == 0 && g.sh % 4 == 0 && g.W % 4 == 0 &&
               (reinterpret_cast<uintptr_t>(residual) & 15) == 0)
                  ? 1 : 0;
    const int rc = launch_gemm_f32_small_mapped(D, ctaps, (int64_t)g.s,
                                                g.b * map, ctaps, e, src, st);
    if (rc != VTC_ERR_UNSUPPORTED) return rc;
  }
  hipLaunchKernelGGL(conv_im2col_kernel, dim3(patch_grid(g.b * map * ctaps)),
                     dim3(256), 0, st, residual, P, g);
  VTC_LAUNCH_CHECK();
  // G[s, pos] = sum_t D[s, t] P[pos, t]: both operands k-contiguous
  return launch_gemm_f32<true, true>(D, ctaps, P, ctaps, g.s, g.b * map, ctaps,
                                     1, e, st);
}

}  // namespace vtc
