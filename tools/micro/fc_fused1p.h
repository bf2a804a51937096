// One-packing variant of the fused persistent FISTA kernel (included by
// fc_fused.hip, three-product modes only).
//
// fused_fista_kernel streams every dictionary tile twice per iteration from L2
// (packA for step 1, packT -- the same numbers transposed -- for step 3) and is
// bound by that stream.  Here the step-1 fragments of NT of a phase's four
// 32-atom tiles are also written, as they arrive, into an LDS image
// [atom][pixel]; step 3 reads them back transposed with ds_read_b64_tr_b16
// (gfx950), so only (4 - NT) / 4 of packT still comes from L2:
//
//   bytes from L2 per 128-atom phase and workgroup: 256 KiB -> 160 KiB (NT = 3)
//
// LDS image of one tile part (32 atoms x 256 pixels, 16-bit, 16 KiB): subtiles
// of 8 atoms x 32 pixels (512 B); inside a subtile a row of 64 B per atom whose
// four 16-byte chunks are XOR-permuted by (atom >> 1) & 3:
//   off(atom, px) = 512 ((atom >> 3) 8 + (px >> 5)) + 64 (atom & 7)
//                   + 16 (((px >> 3) & 3) ^ ((atom >> 1) & 3)) + 2 (px & 7)
// ds_write_b128 of a step-1 fragment (8 consecutive lanes = 8 atoms, one chunk)
// touches 8 different 16-byte slots of 128 B; a 32-lane half of the transposed
// read takes 4 atoms x 64 B = 256 contiguous bytes: both conflict-free.
//
// The image is single-buffered, so the order inside an iteration is
//   A(p) + image writes | epilogue(p) | barrier | T(p) | barrier | A(p+1) ...
// (fused_fista_kernel runs step 1 one phase ahead of step 3, which would need
// two images).  What pays for the image in LDS: the previous codes C live in
// registers for all phases, the patch X is not kept at all -- the residual
// accumulators start every iteration at -X (re-read from memory, latency
// hidden under step 1 of phase 0) -- and the Y' exchange is single-buffered.
#pragma once

namespace vtc {

constexpr int kTilePart = 16384;          // one tile part in LDS
constexpr int kTileSlot = 2 * kTilePart;  // hi + lo

template <int NT>
struct Fused1pLds {
  static constexpr int tile_bytes = NT * kTileSlot;
  static constexpr int yx_bytes = 2 * kYxPart;
  static constexpr int rx_bytes = 2 * kRxPart;
  static constexpr int stat_bytes = 2 * 4 * 32 * 4;
  static constexpr int total = tile_bytes + yx_bytes + rx_bytes + stat_bytes;
};

typedef short s16x4v __attribute__((ext_vector_type(4)));

// two transposed 8-byte reads = one 32x32x16 A fragment (8 k values per lane)
__device__ __forceinline__ uint4 lds_read_tr_frag(const char* lo4,
                                                  const char* hi4) {
  typedef __attribute__((address_space(3))) s16x4v* lds_ptr;
  const s16x4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)lo4);
  const s16x4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)hi4);
  const uint2 ua = __builtin_bit_cast(uint2, a);
  const uint2 ub = __builtin_bit_cast(uint2, b);
  return make_uint4(ua.x, ua.y, ub.x, ub.y);
}

template <int NPH, int MODE, bool F16, int NT, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void fused1p_kernel(FusedParams P) {
  static_assert(NT >= 1 && NT <= 3, "tiles of a phase kept in LDS");
  constexpr int NP = 2;
  using L = Fused1pLds<NT>;
  constexpr int RING = 8;
  constexpr int KS_LDS = 2 * NT;               // step-3 k-steps served by LDS
  constexpr int SP = 16 + 2 * (8 - KS_LDS);    // stream positions per phase
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Tile = smem;
  char* Yx = smem + L::tile_bytes;
  char* Rx = Yx + L::yx_bytes;
  float* Stat = reinterpret_cast<float*>(Rx + L::rx_bytes);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int64_t patch = (int64_t)blockIdx.x * kFP + r;
  const bool live = patch < P.b;
  const int s = P.s;

  const unsigned pack_bytes_total = (unsigned)s * kFN * 2u;
  const unsigned a_wave_off = (unsigned)w * (16u * 64u * 16u);
  const unsigned t_wave_off = (unsigned)(2 * w) * (8u * 64u * 16u);
  __amdgpu_buffer_rsrc_t rsA[NP], rsT[NP];
#pragma unroll
  for (int part = 0; part < NP; ++part) {
    rsA[part] = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)P.packA[part] + a_wave_off), 0,
        (int)(pack_bytes_total - a_wave_off), 0x00020000);
    rsT[part] = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)P.packT[part] + t_wave_off), 0,
        (int)(pack_bytes_total - t_wave_off), 0x00020000);
  }
  const unsigned frag_voff = (unsigned)lane * 16u;
#define VTC1_LOAD_A(part, p, i) \
  buffer_load16(rsA[part], frag_voff, (unsigned)(((4 * (p)) * 16 + (i)) * 1024))
#define VTC1_LOAD_T(part, p, nb, ks) \
  buffer_load16(rsT[part], frag_voff,  \
                (unsigned)((((p) * 8 + (nb)) * 8 + (ks)) * 1024))
  // stream position j of phase p: 0..15 the step-1 fragments, then the step-3
  // fragments of the k-steps the image does not hold, in consumption order
#define VTC1_LOAD_POS(part, p, j)                                        \
  ((j) < 16 ? VTC1_LOAD_A(part, p, j)                                    \
            : VTC1_LOAD_T(part, p, ((j) - 16) & 1, KS_LDS + (((j) - 16) >> 1)))
#define VTC1_SLOT(p, j) ((SP * (p) + (j)) % RING)
#define VTC1_REFILL(p, j)                                                 \
  {                                                                       \
    const int j_ = (j) + RING;                                            \
    const int p_ = (j_ < SP) ? (p) : (((p) + 1) % NPH);                   \
    const int jj_ = (j_ < SP) ? j_ : j_ - SP;                             \
    _Pragma("unroll") for (int part = 0; part < NP; ++part)               \
        ring[part][VTC1_SLOT(p, j)] = VTC1_LOAD_POS(part, p_, jj_);       \
  }

  // LDS lane bases
  const int yx_rd = r * kYxRow + 16 * h;            // + 32 ks
  const int yx_wr = r * kYxRow + 64 * w + 8 * h;    // + 16 g
  const int rx_rd = r * kRxRow + 16 * h;            // + 32 ks
  const int rx_wr = r * kRxRow + 128 * w + 8 * h;   // + 64 nb + 16 g
  // image write of step-1 fragment i: lane = atom (lane & 31), chunk 2 (i & 1) + h
  const int sw = (lane >> 1) & 3;
  const int tw_base = w * kTileSlot + 4096 * (r >> 3) + 64 * (lane & 7);
  const int tw0 = tw_base + 16 * ((0 + h) ^ sw);    // even k-steps
  const int tw1 = tw_base + 16 * ((2 + h) ^ sw);    // odd k-steps
  // transposed read: lane = 32 kg + 16 g1 + 4 q + p supplies atom 4 r2 + q of
  // the 8-atom group, pixels 16 g1 + 4 p .. + 3 of the 32-pixel block
  const int tq = (lane >> 2) & 3, tp = lane & 3, tg1 = (lane >> 4) & 1;
  const int tr_common = 4096 * h + 64 * tq + 8 * (tp & 1);
  const int tr0 = tr_common + 16 * ((2 * tg1 + (tp >> 1)) ^ (tq >> 1));
  const int tr1 = tr_common + 256 + 16 * ((2 * tg1 + (tp >> 1)) ^ (2 + (tq >> 1)));

  // ---- per-wave state --------------------------------------------------
  f32x16v Y[NPH];    // gradient evaluation point, this wave's tile per phase
  f32x16v C[NPH];    // previous codes
  f32x16v Racc[2];   // residual accumulators, this wave's two 32-pixel blocks
  uint4 ring[NP][RING];

  const bool warm = (P.init != nullptr);
#pragma unroll
  for (int p = 0; p < NPH; ++p) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (warm && live)
        v = *reinterpret_cast<const float4*>(
            P.init + patch * s + kPhaseAtoms * p + 32 * w + 8 * g + 4 * h);
      Y[p][4 * g + 0] = v.x;
      Y[p][4 * g + 1] = v.y;
      Y[p][4 * g + 2] = v.z;
      Y[p][4 * g + 3] = v.w;
    }
  }
  // Racc <- x_scale * (-X) for this wave's pixels (exact: x_scale is a power
  // of two); loads only -- `finish_x` applies sign and scale when the values
  // are needed, so that the wait sits there and not behind the loads
  const float* xsrc = P.images + patch * kFN + 64 * w + 4 * h;
  auto load_x = [&]() {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) v = *reinterpret_cast<const float4*>(xsrc + 32 * nb + 8 * g);
        Racc[nb][4 * g + 0] = v.x;
        Racc[nb][4 * g + 1] = v.y;
        Racc[nb][4 * g + 2] = v.z;
        Racc[nb][4 * g + 3] = v.w;
      }
  };
  float x_scale = 1.f;
  auto finish_x = [&]() {
    const float ns = -x_scale;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) Racc[nb][e] *= ns;
  };
  load_x();

  // F16: the patch's power-of-two scale (fc_fused.hip header comment)
  float sigma_y = 1.f, inv_sigma_y = 1.f;   // per lane (= per patch)
  float sigma_d = 1.f, inv_sigma_d = 1.f;
  if (F16) {
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) sx += Racc[nb][e] * Racc[nb][e];
    if (warm) {
#pragma unroll
      for (int p = 0; p < NPH; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) sy += Y[p][e] * Y[p][e];
    }
    sx += __shfl_xor(sx, 32, 64);
    sy += __shfl_xor(sy, 32, 64);
    float* red = reinterpret_cast<float*>(Rx);
    if (h == 0) {
      red[w * 64 + r] = sx;
      red[w * 64 + 32 + r] = sy;
    }
    __syncthreads();
    float tx = 0.f, ty = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      tx += red[v * 64 + r];
      ty += red[v * 64 + 32 + r];
    }
    __syncthreads();
    const float m2 = fmaxf(tx, ty);
    int e2 = 0;
    if (m2 > 0.f && m2 < __builtin_inff()) e2 = ilogbf(m2) >> 1;
    e2 = e2 < -60 ? -60 : (e2 > 60 ? 60 : e2);
    sigma_y = ldexpf(1.f, 8 - e2);
    inv_sigma_y = ldexpf(1.f, e2 - 8);
    sigma_d = P.dscale[0];
    inv_sigma_d = P.dscale[1];
    x_scale = sigma_d * sigma_y;
#pragma unroll
    for (int p = 0; p < NPH; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) Y[p][e] *= sigma_y;
  }
  finish_x();
  float eta = P.eta, cutoff_l = P.cutoff;
  if (P.eta_dev) {
    eta = *P.eta_dev;
    cutoff_l = mul_rn(P.lam, eta);
  }
  if (F16) {
    eta = eta * (0.5f * inv_sigma_d);
    cutoff_l = cutoff_l * sigma_y;
  }
  const float r_scale = F16 ? 2.f * inv_sigma_d : 1.f;
  int xr_calls = 0;
#pragma unroll
  for (int p = 0; p < NPH; ++p) C[p] = Y[p];

  auto publish_y = [&](const f32x16v& y) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float v4[4] = {y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3]};
      uint2 hi, lo;
      split4<F16, NP>(v4, &hi, &lo);
      char* dst = Yx + yx_wr + 16 * g;
      *reinterpret_cast<uint2*>(dst) = hi;
      *reinterpret_cast<uint2*>(dst + kYxPart) = lo;
    }
  };

  // step 3 of phase p.  `direct`: every fragment from packT on the spot (warm
  // start, no image yet); otherwise k-steps < KS_LDS from the image, the rest
  // from the ring.
  auto step3 = [&](int p, bool direct) {
    uint4 yb_next[NP];
#pragma unroll
    for (int part = 0; part < NP; ++part)
      yb_next[part] =
          *reinterpret_cast<const uint4*>(Yx + part * kYxPart + yx_rd);
    // image fragments are read one item ahead of their products
    auto image_frag = [&](int ks, int nb, int part) {
      const char* base = Tile + (ks >> 1) * kTileSlot + part * kTilePart +
                         8192 * (ks & 1) + 512 * (2 * w + nb);
      return lds_read_tr_frag(base + tr0, base + tr1);
    };
    uint4 a_next[NP];
    if (!direct) {
#pragma unroll
      for (int part = 0; part < NP; ++part) a_next[part] = image_frag(0, 0, part);
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      uint4 yb[NP];
#pragma unroll
      for (int part = 0; part < NP; ++part) {
        yb[part] = yb_next[part];
        if (ks + 1 < 8)
          yb_next[part] = *reinterpret_cast<const uint4*>(
              Yx + part * kYxPart + yx_rd + 32 * (ks + 1));
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        uint4 a[NP];
        const bool from_lds = !direct && ks < KS_LDS;
        const int j = 16 + 2 * (ks - KS_LDS) + nb;   // stream position
        const int item = 2 * ks + nb;
#pragma unroll
        for (int part = 0; part < NP; ++part) {
          if (direct) {
            a[part] = VTC1_LOAD_T(part, p, nb, ks);
          } else if (from_lds) {
            a[part] = a_next[part];
            if (item + 1 < 2 * KS_LDS)
              a_next[part] = image_frag((item + 1) >> 1, (item + 1) & 1, part);
          } else {
            a[part] = ring[part][VTC1_SLOT(p, j)];
          }
        }
        Racc[nb] = VTC_MFMA(a[0], yb[0], Racc[nb]);
        Racc[nb] = VTC_MFMA(a[0], yb[1], Racc[nb]);
        Racc[nb] = VTC_MFMA(a[1], yb[0], Racc[nb]);
        if (!direct && !from_lds) VTC1_REFILL(p, j)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // R_{k+1} = (Racc already holds D^T y - X) -> 16-bit parts -> LDS
  auto exchange_r = [&]() {
    float v[2][16];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        v[nb][e] = Racc[nb][e];
        if (F16) v[nb][e] *= r_scale;
      }
    if (F16) {
      // f16 range guard (fc_fused.hip, exchange_r)
      float m = 0.f;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) m = fmaxf(m, fabsf(v[nb][e]));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float f = 1.f;
      if (xr_calls > 0) {
        const float* prev = Stat + ((xr_calls - 1) & 1) * 128;
        const float Mx = fmaxf(fmaxf(prev[r], prev[32 + r]),
                               fmaxf(prev[64 + r], prev[96 + r]));
        if (Mx > 2048.f && Mx < __builtin_inff())
          f = ldexpf(1.f, 9 - ilogbf(Mx));
      }
      if (h == 0) Stat[(xr_calls & 1) * 128 + w * 32 + r] = m * f;
      ++xr_calls;
      if (__any(f != 1.f)) {
#pragma unroll
        for (int p = 0; p < NPH; ++p)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            Y[p][e] *= f;
            C[p][e] *= f;
          }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int e = 0; e < 16; ++e) v[nb][e] *= f;
        x_scale *= f;
        cutoff_l *= f;
        inv_sigma_y *= 1.f / f;
      }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v4[4] = {v[nb][4 * g], v[nb][4 * g + 1], v[nb][4 * g + 2],
                             v[nb][4 * g + 3]};
        uint2 hi, lo;
        split4<F16, NP>(v4, &hi, &lo);
        char* dst = Rx + rx_wr + 64 * nb + 16 * g;
        *reinterpret_cast<uint2*>(dst) = hi;
        *reinterpret_cast<uint2*>(dst + kRxPart) = lo;
      }
    }
    load_x();   // next iteration's accumulators start at -X
    __syncthreads();
  };

  // ---- R_0 = Y_0 D - X ---------------------------------------------------
  if (warm) {
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
      publish_y(Y[p]);
      __syncthreads();
      step3(p, true);
      __syncthreads();
    }
  }
  exchange_r();

#pragma unroll
  for (int i = 0; i < RING; ++i)
#pragma unroll
    for (int part = 0; part < NP; ++part)
      ring[part][i] = VTC1_LOAD_POS(part, 0, i);

  const bool fista = P.fista != 0;
  unsigned long long acc_t[5] = {0, 0, 0, 0, 0};
  unsigned long long t0 = 0, t1 = 0;
#define VTC1_STAMP(slot)                   \
  if (STAMP) {                             \
    t1 = stamp_now();                      \
    acc_t[slot] += t1 - t0;                \
    t0 = t1;                               \
  }
  if (STAMP) t0 = stamp_now();

  f32x16v G;

  // step 1 of phase p: G = D[tile] R_k; the fragments of the first NT waves
  // also go to the image
  auto step1 = [&](int p) {
#pragma unroll
    for (int e = 0; e < 16; ++e) G[e] = 0.f;
    uint4 rb_next[NP];
#pragma unroll
    for (int part = 0; part < NP; ++part)
      rb_next[part] =
          *reinterpret_cast<const uint4*>(Rx + part * kRxPart + rx_rd);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      uint4 rb[NP];
#pragma unroll
      for (int part = 0; part < NP; ++part) {
        rb[part] = rb_next[part];
        if (i + 1 < 16)
          rb_next[part] = *reinterpret_cast<const uint4*>(
              Rx + part * kRxPart + rx_rd + 32 * (i + 1));
      }
      const int sl = VTC1_SLOT(p, i);
      G = VTC_MFMA(ring[0][sl], rb[0], G);
      G = VTC_MFMA(ring[0][sl], rb[1], G);
      G = VTC_MFMA(ring[1][sl], rb[0], G);
      if (w < NT) {
        char* dst = Tile + ((i & 1) ? tw1 : tw0) + 512 * (i >> 1);
        *reinterpret_cast<uint4*>(dst) = ring[0][sl];
        *reinterpret_cast<uint4*>(dst + kTilePart) = ring[1][sl];
      }
      VTC1_REFILL(p, i)
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // proximal step + extrapolation of phase p (ista_fista.py:105-131)
  auto epilogue = [&](int p, float beta) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float yn[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = 4 * g + k;
        const float c = sub_rn(Y[p][e], mul_rn(eta, G[e]));
        const float cn = shrink_fast<MODE>(c, cutoff_l);
        // ISTA (ista_fista.py:133): y = codes, not codes + 0 * (codes - old)
        yn[k] = fista ? add_rn(cn, mul_rn(beta, sub_rn(cn, C[p][e]))) : cn;
        Y[p][e] = yn[k];
        C[p][e] = cn;
      }
      uint2 hi, lo;
      split4<F16, NP>(yn, &hi, &lo);
      char* dst = Yx + yx_wr + 16 * g;
      *reinterpret_cast<uint2*>(dst) = hi;
      *reinterpret_cast<uint2*>(dst + kYxPart) = lo;
    }
  };

  for (int it = 0; it < P.num_iters; ++it) {
    const float beta = fista ? P.betas[it] : 0.f;
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
      step1(p);
      VTC1_STAMP(0)
      epilogue(p, beta);
      VTC1_STAMP(1)
      __syncthreads();
      VTC1_STAMP(2)
      if (p == 0) {
        __builtin_amdgcn_sched_barrier(0);
        finish_x();
      }
      step3(p, false);
      VTC1_STAMP(3)
      if (p + 1 < NPH) __syncthreads();   // image and Y' free again
      VTC1_STAMP(4)
    }
    exchange_r();
    VTC1_STAMP(4)
  }
  if (STAMP && lane == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) atomicAdd(P.stamps + k, acc_t[k]);
    atomicAdd(P.stamps + 7, 1ull);
  }
#undef VTC1_STAMP

  // ---- codes out: the last C -----------------------------------------
#pragma unroll
  for (int p = 0; p < NPH; ++p) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = make_float4(C[p][4 * g + 0], C[p][4 * g + 1], C[p][4 * g + 2],
                             C[p][4 * g + 3]);
      if (F16) {
        v.x *= inv_sigma_y;
        v.y *= inv_sigma_y;
        v.z *= inv_sigma_y;
        v.w *= inv_sigma_y;
      }
      if (live)
        *reinterpret_cast<float4*>(P.codes + patch * s + kPhaseAtoms * p +
                                   32 * w + 8 * g + 4 * h) = v;
    }
  }
#undef VTC1_LOAD_A
#undef VTC1_LOAD_T
#undef VTC1_LOAD_POS
#undef VTC1_SLOT
#undef VTC1_REFILL
}

}  // namespace vtc
