// Variant 3 of the fused FISTA kernel (bf16): every dictionary byte reaches the
// CU once per iteration, without any shared staging buffer.
//
// Variant 1 loads each dictionary tile twice per iteration (row fragments for
// the gradient, transposed fragments for the residual) and both MFMA phases
// run at the per-CU vector-memory rate.  Here each wave keeps the residual
// update to ITS OWN atoms:
//   step 1   G = D[tile w] R_k                          (as before)
//   epilogue Y' for the tile, kept in registers: the accumulator layout of
//            Y' (atoms on the register index, patches on the lanes) is
//            already the B-operand layout of the next product
//   step 3   Racc[all 256 pixels] += D[tile w]^T Y'[tile w]
// so the A operand of step 3 is the transpose of the very fragments the wave
// used in step 1.  They are still in its register ring: the wave writes them,
// 2 KiB at a time, to a private LDS scratch and reads them back transposed
// with ds_read_b64_tr_b16 (no other wave involved, no barrier), then refills
// the ring slots with the next phase's fragments.  No Y' exchange, no barrier
// inside an iteration.  The price: every wave accumulates a partial residual
// over all 256 pixels (8 accumulator tiles instead of 2) and the four partials
// are summed through LDS once per iteration (3 ring-exchange rounds, fixed
// order, bitwise reproducible).
//
// LDS: previous codes of 6 phases (96 KiB) | R exchange (16.5 KiB) | partial-R
// exchange (32 KiB) | transposition scratch (4 waves x 2 KiB).  The patch X is
// re-read from global memory once per iteration instead of living in VGPRs.
#pragma once

namespace vtc {

constexpr int kPrivCreg = 2;                 // phases of C kept in VGPRs
constexpr int kPrivSxBytes = 4 * 8192;       // partial-R exchange
constexpr int kPrivTxBytes = 4 * 2048;       // transposition scratch

template <int NPH>
struct PrivLds {
  static constexpr int CREG = kPrivCreg < NPH ? kPrivCreg : NPH;
  static constexpr int cst_bytes = (NPH - CREG) * 16384;
  static constexpr int total =
      cst_bytes + kRxPart + kPrivSxBytes + kPrivTxBytes;
};

template <int NPH, int MODE, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void fused_fista_priv_kernel(
    FusedParams P) {
  using L = PrivLds<NPH>;
  constexpr int CREG = L::CREG;
  constexpr int CR = CREG > 0 ? CREG : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Cst = smem;
  char* Rx = smem + L::cst_bytes;
  char* Sx = Rx + kRxPart;
  char* Tx = Sx + kPrivSxBytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int64_t patch = (int64_t)blockIdx.x * kFP + r;
  const bool live = patch < P.b;
  const int s = P.s;

  // dictionary row fragments (packA of variant 1), this wave's tiles only
  const unsigned pack_bytes_total = (unsigned)s * kFN * 2u;
  const unsigned a_wave_off = (unsigned)w * (16u * 64u * 16u);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)P.packA[0] + a_wave_off), 0,
      (int)(pack_bytes_total - a_wave_off), 0x00020000);
  const unsigned frag_voff = (unsigned)lane * 16u;
#define VTC_LOAD_A3(p, i) \
  buffer_load16(rsA, frag_voff, (unsigned)(((4 * (p)) * 16 + (i)) * 1024))

  // LDS lane bases
  const int rx_rd = r * kRxRow + 16 * h;            // + 32 ks
  const int rx_wr = r * kRxRow + 128 * w + 8 * h;   // + 64 nb + 16 g
  const int cst_ln = w * 4096 + lane * 16;          // + pl*16384 + g*1024
  // transposition scratch: image [32 atoms][64 B], 16-byte chunk c of row a
  // stored at chunk c ^ ((a >> 1) & 3)
  char* tx_base = Tx + w * 2048;
  const int tx_wr0 = r * 64 + (((0 + h) ^ ((r >> 1) & 3)) << 4);  // ks even
  const int tx_wr1 = r * 64 + (((2 + h) ^ ((r >> 1) & 3)) << 4);  // ks odd
  const int rho = (lane & 15) >> 2, pi = lane & 3, g16 = (lane >> 4) & 1;
  const int tx_rd = (4 * h + rho) * 64 +
                    (((2 * g16 + (pi >> 1)) ^ ((2 * h + (rho >> 1)) & 3)) << 4) +
                    8 * (pi & 1);                   // + (16 s + 8 t) * 64
  // partial-R exchange: region of the sending wave, [block][g][lane] f32x4
  const int sx_ln = lane * 16;                      // + region*8192 + ...

  f32x16v Y[NPH], Cr[CR], Racc[8];
  uint4 ring[16];
  const float* x_row = P.images + (live ? patch : 0) * kFN + 64 * w + 4 * h;

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
      if (p < CREG) {
        Cr[p < CREG ? p : 0][4 * g + 0] = v.x;
        Cr[p < CREG ? p : 0][4 * g + 1] = v.y;
        Cr[p < CREG ? p : 0][4 * g + 2] = v.z;
        Cr[p < CREG ? p : 0][4 * g + 3] = v.w;
      } else {
        *reinterpret_cast<float4*>(Cst + cst_ln + (p - CREG) * 16384 +
                                   g * 1024) = v;
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) Racc[nb][e] = 0.f;

  // Y' tile (accumulator layout) -> the two B fragments of step 3: k-step ss
  // takes accumulator registers 8ss .. 8ss+7 in order (element j of lane half
  // h is atom 16ss + 8(j>>2) + 4h + (j&3) of the tile; the transposed reads
  // below deliver the A operand in the same k order)
  auto y_frag = [&](const f32x16v& y, int ss) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (__bf16)y[8 * ss + j];
    return __builtin_bit_cast(uint4, f);
  };

  // step 3 for this wave's tile of phase p: for each 32-pixel block, transpose
  // the two row fragments that cover it through the private scratch and
  // multiply with Y'.  refill: load the next phase's fragments into the two
  // ring slots just written out.
  auto step3 = [&](const uint4& yb0, const uint4& yb1, bool refill,
                   int next_p) {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      char* img = tx_base;
      *reinterpret_cast<uint4*>(img + tx_wr0) = ring[2 * nb];
      *reinterpret_cast<uint4*>(img + tx_wr1) = ring[2 * nb + 1];
      if (refill) {
        ring[2 * nb] = VTC_LOAD_A3(next_p, 2 * nb);
        ring[2 * nb + 1] = VTC_LOAD_A3(next_p, 2 * nb + 1);
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const char* rd = img + tx_rd + (16 * ss) * 64;
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(rd));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(rd + 8 * 64));
        const uint2 l2 = __builtin_bit_cast(uint2, lo4);
        const uint2 h2 = __builtin_bit_cast(uint2, hi4);
        const uint4 a = make_uint4(l2.x, l2.y, h2.x, h2.y);
        Racc[nb] = VTC_MFMA(a, ss == 0 ? yb0 : yb1, Racc[nb]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // sum of the four waves' partial residuals; wave v ends up owning pixels
  // 64v .. 64v+63 (accumulators 2v, 2v+1), as a ring exchange: in round rr
  // wave w sends its partial of slice (w + rr) % 4 and adds the partial it
  // receives for its own slice from wave (w - rr) % 4.
  auto send_slice = [&](const f32x16v& a0, const f32x16v& a1) {
    char* dst = Sx + w * 8192 + sx_ln;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<float4*>(dst + g * 1024) = make_float4(
          a0[4 * g], a0[4 * g + 1], a0[4 * g + 2], a0[4 * g + 3]);
      *reinterpret_cast<float4*>(dst + 4096 + g * 1024) = make_float4(
          a1[4 * g], a1[4 * g + 1], a1[4 * g + 2], a1[4 * g + 3]);
    }
  };
  auto recv_slice = [&](f32x16v& a0, f32x16v& a1, int from) {
    const char* src = Sx + from * 8192 + sx_ln;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 u = *reinterpret_cast<const float4*>(src + g * 1024);
      const float4 v =
          *reinterpret_cast<const float4*>(src + 4096 + g * 1024);
      a0[4 * g] += u.x; a0[4 * g + 1] += u.y;
      a0[4 * g + 2] += u.z; a0[4 * g + 3] += u.w;
      a1[4 * g] += v.x; a1[4 * g + 1] += v.y;
      a1[4 * g + 2] += v.z; a1[4 * g + 3] += v.w;
    }
  };
  // R_{k+1}[own slice] = sum - X -> bf16 -> Rx (layout of variant 1)
  auto publish_r = [&](const f32x16v& a0, const f32x16v& a1) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const f32x16v& a = nb == 0 ? a0 : a1;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // dead lanes (batch tail) read row 0 and are never stored
        const float4 x =
            *reinterpret_cast<const float4*>(x_row + 32 * nb + 8 * g);
        const float xs[4] = {x.x, x.y, x.z, x.w};
        bf16x4 hi;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          hi[k] = (__bf16)sub_rn(a[4 * g + k], xs[k]);
        *reinterpret_cast<uint2*>(Rx + rx_wr + 64 * nb + 16 * g) =
            __builtin_bit_cast(uint2, hi);
      }
    }
  };
  // w is wave-uniform but a run-time value, and accumulator tiles must be
  // indexed with compile-time constants (a run-time index sends the whole
  // array to scratch memory): every slice gets its own predicated copy of
  // the code, exactly one of which executes in a given wave
  auto reduce_and_publish = [&]() {
#pragma unroll
    for (int rr = 1; rr < 4; ++rr) {
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
        if (((w + rr) & 3) == sl) send_slice(Racc[2 * sl], Racc[2 * sl + 1]);
      __syncthreads();
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
        if (w == sl)
          recv_slice(Racc[2 * sl], Racc[2 * sl + 1], (sl + 4 - rr) & 3);
      __syncthreads();
    }
#pragma unroll
    for (int sl = 0; sl < 4; ++sl)
      if (w == sl) publish_r(Racc[2 * sl], Racc[2 * sl + 1]);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) Racc[nb][e] = 0.f;
    __syncthreads();
  };

  // ---- R_0 = Y_0 D - X ---------------------------------------------------
  if (warm) {
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
#pragma unroll
      for (int i = 0; i < 16; ++i) ring[i] = VTC_LOAD_A3(p, i);
      step3(y_frag(Y[p], 0), y_frag(Y[p], 1), false, 0);
    }
  }
  reduce_and_publish();

#pragma unroll
  for (int i = 0; i < 16; ++i) ring[i] = VTC_LOAD_A3(0, i);

  const float eta = P.eta, cutoff = P.cutoff;
  unsigned long long acc_t[5] = {0, 0, 0, 0, 0};
  unsigned long long t0 = 0, t1 = 0;
#define VTC_STAMP(slot)                    \
  if (STAMP) {                             \
    t1 = stamp_now();                      \
    acc_t[slot] += t1 - t0;                \
    t0 = t1;                               \
  }
  if (STAMP) t0 = stamp_now();
  for (int it = 0; it < P.num_iters; ++it) {
    const float beta = P.betas[it];
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
      // ---- step 1: G = D[tile] R_k
      f32x16v G;
#pragma unroll
      for (int e = 0; e < 16; ++e) G[e] = 0.f;
      uint4 rb_next = *reinterpret_cast<const uint4*>(Rx + rx_rd);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint4 rb = rb_next;
        if (i + 1 < 16)
          rb_next =
              *reinterpret_cast<const uint4*>(Rx + rx_rd + 32 * (i + 1));
        G = VTC_MFMA(ring[i], rb, G);
        __builtin_amdgcn_sched_barrier(0);
      }
      VTC_STAMP(0)
      // ---- proximal step + extrapolation (ista_fista.py:105-131)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 cold;
        if (p < CREG) {
          cold = make_float4(Cr[p < CREG ? p : 0][4 * g + 0],
                             Cr[p < CREG ? p : 0][4 * g + 1],
                             Cr[p < CREG ? p : 0][4 * g + 2],
                             Cr[p < CREG ? p : 0][4 * g + 3]);
        } else {
          cold = *reinterpret_cast<const float4*>(
              Cst + cst_ln + (p - CREG) * 16384 + g * 1024);
        }
        const float co[4] = {cold.x, cold.y, cold.z, cold.w};
        float cn[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = 4 * g + k;
          const float c = sub_rn(Y[p][e], mul_rn(eta, G[e]));
          cn[k] = shrink_fast<MODE>(c, cutoff);
          const float d = sub_rn(cn[k], co[k]);
          Y[p][e] = add_rn(cn[k], mul_rn(beta, d));
        }
        if (p < CREG) {
#pragma unroll
          for (int k = 0; k < 4; ++k) Cr[p < CREG ? p : 0][4 * g + k] = cn[k];
        } else {
          *reinterpret_cast<float4*>(Cst + cst_ln + (p - CREG) * 16384 +
                                     g * 1024) =
              make_float4(cn[0], cn[1], cn[2], cn[3]);
        }
      }
      VTC_STAMP(1)
      // ---- step 3 on the same tile, fragments transposed through LDS
      step3(y_frag(Y[p], 0), y_frag(Y[p], 1), true, (p + 1) % NPH);
      VTC_STAMP(3)
    }
    reduce_and_publish();
    VTC_STAMP(4)
  }
  if (STAMP && lane == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) atomicAdd(P.stamps + k, acc_t[k]);
    atomicAdd(P.stamps + 7, 1ull);
  }
#undef VTC_STAMP

#pragma unroll
  for (int p = 0; p < NPH; ++p) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v;
      if (p < CREG) {
        v = make_float4(Cr[p < CREG ? p : 0][4 * g + 0],
                        Cr[p < CREG ? p : 0][4 * g + 1],
                        Cr[p < CREG ? p : 0][4 * g + 2],
                        Cr[p < CREG ? p : 0][4 * g + 3]);
      } else {
        v = *reinterpret_cast<const float4*>(Cst + cst_ln +
                                             (p - CREG) * 16384 + g * 1024);
      }
      if (live)
        *reinterpret_cast<float4*>(P.codes + patch * s + kPhaseAtoms * p +
                                   32 * w + 8 * g + 4 * h) = v;
    }
  }
#undef VTC_LOAD_A3
}

}  // namespace vtc
