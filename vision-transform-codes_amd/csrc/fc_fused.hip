// Fused persistent FISTA kernel for 16x16 patches (n = 256) on gfx950.
//
// One workgroup (4 waves, one per SIMD) owns 32 patches and runs ALL the
// iterations of analysis_transforms/fully_connected/ista_fista.py:100-146 for
// them without touching HBM in between: the per-patch state (gradient
// evaluation point Y, previous codes C, residual R, the patch X) lives in
// VGPRs / LDS, only the dictionary is streamed (from L2) every iteration.
//
// Orientation.  Everything is computed transposed, patches on the MFMA column
// (= lane) index:
//     G^T[atoms x 32]  = D[atoms x 256]     . R^T[256 x 32]        ("step 1")
//     R^T[256 x 32]   += D^T[256 x atoms]   . Ynew^T[atoms x 32]   ("step 3")
// so that both products consume their right-hand operand with the reduction
// index on the accumulator ROW axis: the result of one product is, after the
// epilogue, the B operand of the next in the same lanes.
//
// Work split.  The atoms are walked in phases of 128 = 4 tiles of 32, one tile
// per wave.  In phase p wave w
//   step 1   G = D[tile 4p+w] R_k                     16 MFMA 32x32x16 (x NP)
//   epilogue C' = shrink(Y - eta G); Y' = C' + beta (C' - C)   (f32, exact op
//            order of the reference), Y' -> bf16 -> LDS exchange buffer
//   barrier
//   step 3   Racc[n-slice w] += D[phase p]^T[n-slice w] Y'[phase p]   16 MFMA
// and after the last phase the waves exchange R_{k+1} = Racc - X through LDS.
// The reference's two GEMMs per iteration become one sweep over D in which
// each dictionary tile is used for the gradient (step 1) and immediately for
// the next residual (step 3).
//
// Dictionary operands are pre-packed once per call into MFMA A-fragment order
// (pack_dictionary_kernel), so every fragment load is one fully coalesced
// 1 KiB global_load_dwordx4 per wave, straight to registers, prefetched one
// step (16 fragments) ahead through a register ring.
//
// Precision (NP): 1 = single bf16 product (fast mode); 2 = bf16 hi/lo split of
// both operands, three products hi*hi + hi*lo + lo*hi accumulated in f32
// (bf16x3, ~2^-16 relative per product: float32-level results on the bf16
// matrix pipe).  State, epilogue and accumulation are f32 in both.
#include "fc_fused.h"

#include <stdlib.h>
#include <vector>

namespace vtc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));

constexpr int kFP = 32;    // patches per workgroup
constexpr int kFN = 256;   // pixels per patch
constexpr int kPhaseAtoms = 128;

// ---------------------------------------------------------------- packing
// packA fragment (tile t of 32 atoms, k-step ks over pixels), lane l:
//   D[32t + (l&31)][16ks + 8(l>>5) + j],  j = 0..7
// packT fragment (phase p, pixel block nb of 32, k-step ks over the phase's
// atoms), lane l:
//   D[128p + 16ks + 8(l>>5) + j][32nb + (l&31)]
// LO = 0 stores bf16(x), LO = 1 stores bf16(x - float(bf16(x))).
__device__ __forceinline__ __bf16 split_part(float x, int lo) {
  const __bf16 hi = (__bf16)x;
  if (!lo) return hi;
  return (__bf16)(x - (float)hi);
}

__global__ void pack_dictionary_kernel(const float* __restrict__ D, int s,
                                       __bf16* __restrict__ packA,
                                       __bf16* __restrict__ packT, int lo) {
  const int64_t frags = (int64_t)s * kFN / 8;  // 16-byte units per packing
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < frags;
       u += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(u & 63);
    const int r = l & 31, h = l >> 5;
    {
      const int64_t f = u >> 6;  // = t*16 + ks
      const int t = (int)(f >> 4), ks = (int)(f & 15);
      const float* src = D + (int64_t)(32 * t + r) * kFN + 16 * ks + 8 * h;
#pragma unroll
      for (int j = 0; j < 8; ++j) packA[u * 8 + j] = split_part(src[j], lo);
    }
    {
      const int64_t f = u >> 6;  // = (p*8 + nb)*8 + ks
      const int ks = (int)(f & 7), nb = (int)((f >> 3) & 7), p = (int)(f >> 6);
      const float* src =
          D + (int64_t)(128 * p + 16 * ks + 8 * h) * kFN + 32 * nb + r;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        packT[u * 8 + j] = split_part(src[(int64_t)j * kFN], lo);
    }
  }
}

struct FusedParams {
  const float* images;
  const float* init;   // may be null
  float* codes;
  const uint4* packA[2];  // [hi, lo]
  const uint4* packT[2];
  const float* betas;
  int64_t b;
  int s;
  int num_iters;
  float eta, cutoff;
  unsigned long long* stamps;  // diagnostic build only: 8 cycle sums
};

template <int MODE>
__device__ __forceinline__ float shrink_fast(float c, float cutoff) {
  if (MODE == VTC_SOFT) {
    // sign(c) * max(|c| - t, 0) == c - clamp(c, -t, t) bit for bit (up to the
    // sign of a zero result): one v_med3 + one v_sub.
    return sub_rn(c, __builtin_amdgcn_fmed3f(c, -cutoff, cutoff));
  }
  return shrink(c, cutoff, MODE);
}

__device__ __forceinline__ bf16x8 as_frag(const uint4& u) {
  return __builtin_bit_cast(bf16x8, u);
}

#define VTC_MFMA(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a), as_frag(b), c, 0, 0, 0)

// LDS plan (bytes), NPH phases, NP precision parts, CREG phases of C in VGPRs:
//   Cst : (NPH-CREG) x 16 KiB   previous codes, [phase][wave][group][lane] f32x4
//   Yx  : 2 x NP x 8704         Y' exchange, double buffered,
//                               [buf][part][patch][256 B + 16 B pad]
//   Rx  : NP x 16896            R exchange, [part][patch][512 B + 16 B pad]
// The 16-byte row pad makes the ds_read_b128 fragment reads of a 16-lane
// group (16 different patches, same column chunk) hit 16 different 4-bank
// slots, and keeps every address of the form lane_base + immediate.
constexpr int kYxRow = 272;
constexpr int kRxRow = 528;
constexpr int kYxPart = 32 * kYxRow;   // 8704
constexpr int kRxPart = 32 * kRxRow;   // 16896

template <int NPH, int NP>
struct FusedLds {
  static constexpr int CREG_WANT = (NP == 1) ? 1 : 3;
  static constexpr int CREG = CREG_WANT < NPH ? CREG_WANT : NPH;
  static constexpr int CL = NPH - CREG;
  static constexpr int cst_bytes = CL * 16384;
  static constexpr int yx_bytes = 2 * NP * kYxPart;
  static constexpr int rx_bytes = NP * kRxPart;
  static constexpr int total = cst_bytes + yx_bytes + rx_bytes;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 buffer_load16(__amdgpu_buffer_rsrc_t rsrc,
                                               unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// In-kernel stamps (diagnostic instantiation only, STAMP = true): where a
// phase spends its cycles.  s_memtime + its wait in one statement, fenced.
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

template <int NPH, int NP, int MODE, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void fused_fista_kernel(FusedParams P) {
  using L = FusedLds<NPH, NP>;
  constexpr int CREG = L::CREG;
  constexpr int CR = CREG > 0 ? CREG : 1;
  // fragments (k-steps) in flight; must divide 16.  bf16x3 with 16 spills and
  // measures slower (77.7 vs 74.8 ms): the phases are bandwidth-, not
  // latency-bound
  constexpr int RING = (NP == 1) ? 16 : 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Cst = smem;
  char* Yx = smem + L::cst_bytes;
  char* Rx = Yx + L::yx_bytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int64_t patch = (int64_t)blockIdx.x * kFP + r;
  const bool live = patch < P.b;
  const int s = P.s;

  // Dictionary fragments through buffer loads: wave-uniform descriptor (base
  // already offset to this wave's share), one VGPR byte offset (lane * 16),
  // the fragment index as a scalar offset.
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
  //   packA fragment (phase p, k-step i):      ((4p) * 16 + i) * 1024 bytes
  //   packT fragment (phase p, block nb, ks):  ((p * 8 + nb) * 8 + ks) * 1024
#define VTC_LOAD_A(part, p, i) \
  buffer_load16(rsA[part], frag_voff, (unsigned)(((4 * (p)) * 16 + (i)) * 1024))
#define VTC_LOAD_T(part, p, nb, ks) \
  buffer_load16(rsT[part], frag_voff,  \
                (unsigned)((((p) * 8 + (nb)) * 8 + (ks)) * 1024))

  // LDS lane bases
  const int yx_rd = r * kYxRow + 16 * h;            // + 32 ks
  const int yx_wr = r * kYxRow + 64 * w + 8 * h;    // + 16 g
  const int rx_rd = r * kRxRow + 16 * h;            // + 32 ks
  const int rx_wr = r * kRxRow + 128 * w + 8 * h;   // + 64 nb + 16 g
  const int cst_ln = w * 4096 + lane * 16;          // + pl*16384 + g*1024

  // ---- per-wave state --------------------------------------------------
  f32x16v Y[NPH];    // gradient evaluation point, this wave's tile per phase
  f32x16v Cr[CR];    // previous codes of the first CREG phases
  f32x16v Xr[2];     // the patches, this wave's two 32-pixel blocks
  f32x16v Racc[2];
  uint4 ring[NP][RING];

  // element e of a 32x32 accumulator: row (e&3) + 8(e>>2) + 4h, column r
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live)
        v = *reinterpret_cast<const float4*>(
            P.images + patch * kFN + 64 * w + 32 * nb + 8 * g + 4 * h);
      Xr[nb][4 * g + 0] = v.x;
      Xr[nb][4 * g + 1] = v.y;
      Xr[nb][4 * g + 2] = v.z;
      Xr[nb][4 * g + 3] = v.w;
    }
  }
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
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) Racc[nb][e] = 0.f;

  // write the wave's tile of Y (as bf16 parts) into exchange buffer `buf`
  auto publish_y = [&](const f32x16v& y, int buf) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = y[4 * g + k];
        hi[k] = (__bf16)v;
        if (NP == 2) lo[k] = (__bf16)(v - (float)hi[k]);
      }
      char* dst = Yx + buf * NP * kYxPart + yx_wr + 16 * g;
      *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, hi);
      if (NP == 2)
        *reinterpret_cast<uint2*>(dst + kYxPart) =
            __builtin_bit_cast(uint2, lo);
    }
  };

  // Per iteration the dictionary is consumed as a stream of 2*NPH segments of
  // 16 fragments; the epilogue of phase p is software-pipelined under step 1
  // of phase p+1, so the order is
  //   A(0) | A(1) T(0) | A(2) T(1) | ... | A(NPH-1) T(NPH-2) | T(NPH-1)
  // (A(p) = step-1 fragments of phase p, T(p) = its step-3 fragments).
  // Segment sigma: is it a T segment, and of which phase?
#define VTC_SEG_IS_T(sg) (((sg) >= 2 && ((sg) % 2) == 0) || (sg) == 2 * NPH - 1)
#define VTC_SEG_PHASE(sg)                                        \
  ((sg) == 0 ? 0                                                 \
             : (sg) == 2 * NPH - 1 ? NPH - 1                     \
                                   : ((sg) % 2 ? ((sg) + 1) / 2 : (sg) / 2 - 1))
#define VTC_LOAD_SEG(part, sg, i)                                          \
  (VTC_SEG_IS_T(sg) ? VTC_LOAD_T(part, VTC_SEG_PHASE(sg), (i) & 1, (i) >> 1) \
                    : VTC_LOAD_A(part, VTC_SEG_PHASE(sg), (i)))
  // refill of the ring slot consumed at (segment sg, position i): the
  // fragment RING positions further down the stream (wrapping to the next
  // iteration's first segment)
#define VTC_REFILL(sg, i)                                                   \
  {                                                                         \
    const int j_ = (i) + RING;                                              \
    const int sg_ = (j_ < 16) ? (sg) : (((sg) + 1) % (2 * NPH));            \
    const int i_ = (j_ < 16) ? j_ : j_ - 16;                                \
    _Pragma("unroll") for (int part = 0; part < NP; ++part)                 \
        ring[part][(i) % RING] = VTC_LOAD_SEG(part, sg_, i_);               \
  }

  // step 3 of phase p: Racc[nb] += D^T fragments x Y' fragments.
  // pipe: fragments come from the ring (stream segment sg); otherwise they
  // are loaded on the spot (warm-start prologue).
  // hipcc would sink every dictionary load down to its consumer (to shorten
  // live ranges), which serialises load -> wait -> MFMA; the sched_barrier
  // after each k-step pins the issue order written here.
  auto step3 = [&](int p, int buf, bool pipe, int sg) {
    uint4 yb_next[NP];
#pragma unroll
    for (int part = 0; part < NP; ++part)
      yb_next[part] = *reinterpret_cast<const uint4*>(
          Yx + (buf * NP + part) * kYxPart + yx_rd);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      uint4 yb[NP];
#pragma unroll
      for (int part = 0; part < NP; ++part) {
        yb[part] = yb_next[part];
        if (ks + 1 < 8)
          yb_next[part] = *reinterpret_cast<const uint4*>(
              Yx + (buf * NP + part) * kYxPart + yx_rd + 32 * (ks + 1));
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int i = 2 * ks + nb;          // position in the segment
        uint4 a[NP];
#pragma unroll
        for (int part = 0; part < NP; ++part)
          a[part] = pipe ? ring[part][i % RING] : VTC_LOAD_T(part, p, nb, ks);
        Racc[nb] = VTC_MFMA(a[0], yb[0], Racc[nb]);
        if (NP == 2) {
          Racc[nb] = VTC_MFMA(a[0], yb[1], Racc[nb]);
          Racc[nb] = VTC_MFMA(a[1], yb[0], Racc[nb]);
        }
        if (pipe) VTC_REFILL(sg, i)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // R_{k+1} = Racc - X  ->  bf16 parts -> LDS (read back as B fragments by
  // every wave during step 1 of the next iteration)
  auto exchange_r = [&]() {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 hi, lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v = sub_rn(Racc[nb][4 * g + k], Xr[nb][4 * g + k]);
          hi[k] = (__bf16)v;
          if (NP == 2) lo[k] = (__bf16)(v - (float)hi[k]);
        }
        char* dst = Rx + rx_wr + 64 * nb + 16 * g;
        *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, hi);
        if (NP == 2)
          *reinterpret_cast<uint2*>(dst + kRxPart) =
              __builtin_bit_cast(uint2, lo);
      }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) Racc[nb][e] = 0.f;
    __syncthreads();
  };

  // ---- R_0 = Y_0 D - X ---------------------------------------------------
  if (warm) {
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
      publish_y(Y[p], p & 1);
      __syncthreads();
      step3(p, p & 1, false, 0);
    }
  }
  exchange_r();

  // prime the ring with the first RING fragments of segment 0
#pragma unroll
  for (int i = 0; i < RING; ++i)
#pragma unroll
    for (int part = 0; part < NP; ++part)
      ring[part][i] = VTC_LOAD_SEG(part, 0, i);

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

  f32x16v Gb[2];     // gradient tiles of two consecutive phases
  float4 cold4;      // previous codes of the 4 elements being processed
  float cn4[4];

  // Proximal step + extrapolation for element e of phase p
  // (ista_fista.py:105-131); elements are visited in order 0..15, group
  // loads/stores of C and the bf16 publication of Y' happen at group edges.
  auto epilogue_elem = [&](int p, int e, const f32x16v& Gp, float beta) {
    const int g = e >> 2, k = e & 3;
    if (k == 0) {
      if (p < CREG) {
        cold4 = make_float4(Cr[p < CREG ? p : 0][4 * g + 0],
                            Cr[p < CREG ? p : 0][4 * g + 1],
                            Cr[p < CREG ? p : 0][4 * g + 2],
                            Cr[p < CREG ? p : 0][4 * g + 3]);
      } else {
        cold4 = *reinterpret_cast<const float4*>(
            Cst + cst_ln + (p - CREG) * 16384 + g * 1024);
      }
    }
    const float co = (k == 0) ? cold4.x : (k == 1) ? cold4.y
                   : (k == 2) ? cold4.z : cold4.w;
    const float c = sub_rn(Y[p][e], mul_rn(eta, Gp[e]));
    const float cn = shrink_fast<MODE>(c, cutoff);
    const float d = sub_rn(cn, co);
    Y[p][e] = add_rn(cn, mul_rn(beta, d));
    cn4[k] = cn;
    if (k == 3) {
      if (p < CREG) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Cr[p < CREG ? p : 0][4 * g + q] = cn4[q];
      } else {
        *reinterpret_cast<float4*>(Cst + cst_ln + (p - CREG) * 16384 +
                                   g * 1024) =
            make_float4(cn4[0], cn4[1], cn4[2], cn4[3]);
      }
      bf16x4 hi, lo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = Y[p][4 * g + q];
        hi[q] = (__bf16)v;
        if (NP == 2) lo[q] = (__bf16)(v - (float)hi[q]);
      }
      char* dst = Yx + (p & 1) * NP * kYxPart + yx_wr + 16 * g;
      *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, hi);
      if (NP == 2)
        *reinterpret_cast<uint2*>(dst + kYxPart) =
            __builtin_bit_cast(uint2, lo);
    }
  };

  // step 1 of phase p (stream segment sg): Gb[p&1] = D[tile] R_k, with the
  // epilogue of phase p-1 interleaved element by element when `overlap`
  auto step1 = [&](int p, int sg, bool overlap, float beta) {
    f32x16v& G = Gb[p & 1];
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
      G = VTC_MFMA(ring[0][i % RING], rb[0], G);
      if (NP == 2) {
        G = VTC_MFMA(ring[0][i % RING], rb[1], G);
        G = VTC_MFMA(ring[1][i % RING], rb[0], G);
      }
      // epilogue arithmetic first, then the refill: a wave blocks at a
      // buffer load while the vector-memory path is busy, and in that order
      // its VALU work would wait behind the load instead of overlapping it
      if (overlap) {
        epilogue_elem(p - 1, i, Gb[(p - 1) & 1], beta);
        __builtin_amdgcn_sched_barrier(0);
      }
      VTC_REFILL(sg, i)
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  for (int it = 0; it < P.num_iters; ++it) {
    const float beta = P.betas[it];
    step1(0, 0, false, beta);
    VTC_STAMP(0)
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
      if (p + 1 < NPH) {
        step1(p + 1, 2 * p + 1, true, beta);   // + epilogue of phase p
        VTC_STAMP(0)
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) epilogue_elem(p, e, Gb[p & 1], beta);
        VTC_STAMP(1)
      }
      __syncthreads();
      VTC_STAMP(2)
      // ---- step 3: next residual, this wave's 64 pixels
      step3(p, p & 1, true, (p + 1 < NPH) ? 2 * p + 2 : 2 * NPH - 1);
      VTC_STAMP(3)
    }
    exchange_r();
    VTC_STAMP(4)
  }
  if (STAMP && lane == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) atomicAdd(P.stamps + k, acc_t[k]);
    atomicAdd(P.stamps + 7, 1ull);
  }
#undef VTC_STAMP

  // ---- codes out: the last C -----------------------------------------
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
#undef VTC_LOAD_A
#undef VTC_LOAD_T
#undef VTC_LOAD_SEG
#undef VTC_REFILL
#undef VTC_SEG_IS_T
#undef VTC_SEG_PHASE
}

// ===========================================================================
// Variant 2 (bf16): the dictionary goes through LDS once per iteration.
//
// In the kernel above every dictionary byte reaches the CU twice per
// iteration (row-wise fragments for step 1, transposed fragments for step 3),
// and both MFMA phases run at the per-CU vector-memory rate (64 B/clk) rather
// than at the MFMA rate (in-kernel stamps, profiles/r01_fused_stamps.txt).
// Here a phase's 128 x 256 bf16 block (64 KiB) is copied ONCE by LDS-DMA
// (global_load_lds_dwordx4, no VGPRs) into one of two LDS buffers; step 1 reads
// its A fragments from it by rows (ds_read_b128) and step 3 reads the SAME image
// transposed (ds_read_b64_tr_b16).  The global copy is pre-packed as the exact
// LDS image, XOR-swizzled so that both read patterns are bank-conflict free:
//   element (atom row r of the phase, pixel x) -> byte
//   r*512 + (((x>>3) ^ f(r)) << 4) + (x&7)*2,   f(r) = 4*(r&3) + ((r>>3)&3)
// With the LDS holding 2 x 64 KiB of dictionary the previous codes C move from
// LDS to VGPRs (Y and C: 256 VGPRs per lane); the register ring disappears.
// Two barriers per phase: after the epilogue (Y' published) and after step 3
// (buffer and exchange area free; the barrier also drains the DMA issued one
// phase ahead).
// ===========================================================================
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__global__ void pack_lds_image_kernel(const float* __restrict__ D, int s,
                                      __bf16* __restrict__ image) {
  const int64_t chunks = (int64_t)s * kFN / 8;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < chunks;
       u += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(u & 31);          // 8-pixel chunk of the row
    const int64_t atom = u >> 5;
    const int rl = (int)(atom & 127);     // row inside the phase
    const int f = 4 * (rl & 3) + ((rl >> 3) & 3);
    const float* src = D + atom * kFN + 8 * c;
    __bf16* dst = image + (atom >> 7) * 32768 + (int64_t)rl * 256 +
                  ((c ^ f) << 3);
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] = (__bf16)src[j];
  }
}

constexpr int kDbufBytes = 65536;
constexpr int kLdsV2Total = 2 * kDbufBytes + kYxPart + kRxPart;  // 156672

template <int NPH, int MODE, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void fused_fista_lds_kernel(
    FusedParams P) {
  static_assert(NPH % 2 == 0, "buffer parity must continue across iterations");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Dbuf = smem;
  char* Yx = smem + 2 * kDbufBytes;
  char* Rx = Yx + kYxPart;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int64_t patch = (int64_t)blockIdx.x * kFP + r;
  const bool live = patch < P.b;
  const int s = P.s;

  // LDS lane bases (exchange areas as in the kernel above)
  const int yx_rd = r * kYxRow + 16 * h;
  const int yx_wr = r * kYxRow + 64 * w + 8 * h;
  const int rx_rd = r * kRxRow + 16 * h;
  const int rx_wr = r * kRxRow + 128 * w + 8 * h;
  // step-1 row reads: row 32w + r of the phase image, chunk (2ks + h) ^ fR
  const int fR = 4 * (r & 3) + ((r >> 3) & 3);
  const int a_row = (32 * w + r) * 512;
  // step-3 transposed reads: lane (rho, pi) of 16-lane group g16
  const int rho = (lane & 15) >> 2, pi = lane & 3, g16 = (lane >> 4) & 1;
  const int t_lane_chunk = ((pi >> 1) ^ h) | (g16 << 1) | ((rho & 1) << 2) |
                           (((w & 1) ^ (rho >> 1)) << 3) | ((w >> 1) << 4);
  const int t_row = (8 * h + rho) * 512 + 8 * (pi & 1);

  // LDS-DMA through a buffer descriptor: one VGPR (lane * 16) for every copy,
  // the piece address as a scalar offset.  (With flat pointers hipcc
  // pre-computes a 64-bit VGPR address per piece and keeps all 16 * NPH of
  // them live across the iteration loop.)
  const __amdgpu_buffer_rsrc_t image_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)P.packA[0], 0, (int)((unsigned)s * kFN * 2u), 0x00020000);
  const int dma_voff = lane * 16;
  const int dma_wave = w * 16384;
  // one phase = 64 pieces of 1 KiB; wave w copies pieces 16w .. 16w+15
  auto dma_piece = [&](int p, int buf, int i) {
    char* dst = Dbuf + buf * kDbufBytes + dma_wave;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        image_rsrc, (lds_void*)(dst + i * 1024), 16, dma_voff,
        p * kDbufBytes + dma_wave + i * 1024, 0, 0);
  };
  auto dma_phase = [&](int p, int buf) {
#pragma unroll
    for (int i = 0; i < 16; ++i) dma_piece(p, buf, i);
  };

  f32x16v Y[NPH], C[NPH], Xr[2], Racc[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live)
        v = *reinterpret_cast<const float4*>(
            P.images + patch * kFN + 64 * w + 32 * nb + 8 * g + 4 * h);
      Xr[nb][4 * g + 0] = v.x;
      Xr[nb][4 * g + 1] = v.y;
      Xr[nb][4 * g + 2] = v.z;
      Xr[nb][4 * g + 3] = v.w;
    }
  }
  const bool warm = (P.init != nullptr);
#pragma unroll
  for (int p = 0; p < NPH; ++p) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (warm && live)
        v = *reinterpret_cast<const float4*>(
            P.init + patch * s + kPhaseAtoms * p + 32 * w + 8 * g + 4 * h);
      Y[p][4 * g + 0] = C[p][4 * g + 0] = v.x;
      Y[p][4 * g + 1] = C[p][4 * g + 1] = v.y;
      Y[p][4 * g + 2] = C[p][4 * g + 2] = v.z;
      Y[p][4 * g + 3] = C[p][4 * g + 3] = v.w;
    }
  }
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) Racc[nb][e] = 0.f;

  auto publish_y = [&](const f32x16v& y) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 hi;
#pragma unroll
      for (int k = 0; k < 4; ++k) hi[k] = (__bf16)y[4 * g + k];
      *reinterpret_cast<uint2*>(Yx + yx_wr + 16 * g) =
          __builtin_bit_cast(uint2, hi);
    }
  };

  // step 3 from LDS buffer `buf`: Racc[nb] += D^T (transposed reads) x Y'.
  // Operands are read one MFMA ahead; the sched_barrier per MFMA keeps hipcc
  // from hoisting all 48 LDS reads of the step to its top (register blow-up).
  auto tr_frag = [&](const char* base, int ks, int nb) {
    const int kconst = ((ks & 1) << 1) | (nb << 2);
    const int off = ((t_lane_chunk ^ kconst) << 4);
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(base + off +
                                                   (16 * ks + 0) * 512));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(base + off +
                                                   (16 * ks + 4) * 512));
    const uint2 l2 = __builtin_bit_cast(uint2, lo4);
    const uint2 h2 = __builtin_bit_cast(uint2, hi4);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
  };
  auto step3 = [&](int buf, int next_p, bool prefetch) {
    const char* base = Dbuf + buf * kDbufBytes + t_row;
    uint4 a_next = tr_frag(base, 0, 0);
    uint4 yb_next = *reinterpret_cast<const uint4*>(Yx + yx_rd);
    uint4 yb = yb_next;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ks = i >> 1, nb = i & 1;
      const uint4 a = a_next;
      if (nb == 0) {
        yb = yb_next;
        if (ks + 1 < 8)
          yb_next =
              *reinterpret_cast<const uint4*>(Yx + yx_rd + 32 * (ks + 1));
      }
      if (i + 1 < 16) a_next = tr_frag(base, (i + 1) >> 1, (i + 1) & 1);
      Racc[nb] = VTC_MFMA(a, yb, Racc[nb]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  auto exchange_r = [&]() {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 hi;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          hi[k] = (__bf16)sub_rn(Racc[nb][4 * g + k], Xr[nb][4 * g + k]);
        *reinterpret_cast<uint2*>(Rx + rx_wr + 64 * nb + 16 * g) =
            __builtin_bit_cast(uint2, hi);
      }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) Racc[nb][e] = 0.f;
    __syncthreads();
  };

  // ---- R_0 = Y_0 D - X ---------------------------------------------------
  if (warm) {
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
      dma_phase(p, p & 1);
      publish_y(Y[p]);
      __syncthreads();          // drains the DMA, publishes Y'
      step3(p & 1, 0, false);
      __syncthreads();
    }
  }
  exchange_r();
  dma_phase(0, 0);
  __syncthreads();

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
      // prefetch the next phase into the other buffer (free since the barrier
      // that closed the previous phase).  Measured on MI355X: a 1 KiB LDS-DMA
      // piece costs the issuing wave ~200 cycles wherever it is placed (as one
      // burst here: step 1 = 2186 cycles; sprinkled between the MFMAs of
      // step 1 / step 3: 2255 / 1481), i.e. the LDS-DMA path delivers less
      // than the plain buffer loads of the register-ring variant, which is why
      // this variant is not the default.
      dma_phase((p + 1) % NPH, (p + 1) & 1);
      // ---- step 1: G = D[tile] R_k, A fragments by rows from LDS
      f32x16v G;
#pragma unroll
      for (int e = 0; e < 16; ++e) G[e] = 0.f;
      const char* abase = Dbuf + (p & 1) * kDbufBytes + a_row;
      uint4 a_next =
          *reinterpret_cast<const uint4*>(abase + (((0 + h) ^ fR) << 4));
      uint4 rb_next = *reinterpret_cast<const uint4*>(Rx + rx_rd);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint4 a = a_next, rb = rb_next;
        if (i + 1 < 16) {
          a_next = *reinterpret_cast<const uint4*>(
              abase + (((2 * (i + 1) + h) ^ fR) << 4));
          rb_next =
              *reinterpret_cast<const uint4*>(Rx + rx_rd + 32 * (i + 1));
        }
        G = VTC_MFMA(a, rb, G);
        __builtin_amdgcn_sched_barrier(0);
      }
      VTC_STAMP(0)
      // ---- proximal step + extrapolation (ista_fista.py:105-131)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float c = sub_rn(Y[p][e], mul_rn(eta, G[e]));
        const float cn = shrink_fast<MODE>(c, cutoff);
        const float d = sub_rn(cn, C[p][e]);
        Y[p][e] = add_rn(cn, mul_rn(beta, d));
        C[p][e] = cn;
      }
      publish_y(Y[p]);
      VTC_STAMP(1)
      // Y' visible to the other waves; LDS writes only, the DMA stays in flight
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      VTC_STAMP(2)
      step3(p & 1, (p + 1) % NPH, true);
      VTC_STAMP(3)
      __syncthreads();   // buffer p&1 and Yx free; next phase's DMA landed
      VTC_STAMP(2)
    }
    exchange_r();
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
      if (live)
        *reinterpret_cast<float4*>(P.codes + patch * s + kPhaseAtoms * p +
                                   32 * w + 8 * g + 4 * h) =
            make_float4(C[p][4 * g + 0], C[p][4 * g + 1], C[p][4 * g + 2],
                        C[p][4 * g + 3]);
    }
  }
}

}  // namespace vtc
#include "fc_fused_priv.h"
namespace vtc {

// -------------------------------------------------------------------- host
static int phases_for(int64_t s) { return (int)(s / kPhaseAtoms); }

bool fused_shape_supported(int64_t b, int64_t n, int64_t s, int precision) {
  if (n != kFN || b <= 0) return false;
  if (s % kPhaseAtoms != 0) return false;
  const int nph = phases_for(s);
  if (!(nph == 2 || nph == 4 || nph == 8)) return false;
  return precision == VTC_BF16 || precision == VTC_BF16X3;
}

static size_t pack_bytes(int64_t s) { return (size_t)s * kFN * sizeof(__bf16); }

size_t fused_workspace_bytes(int64_t b, int64_t n, int64_t s, int precision) {
  if (!fused_shape_supported(b, n, s, precision)) return 256;
  const int parts = (precision == VTC_BF16X3) ? 2 : 1;
  return (size_t)parts * 2 * align_up(pack_bytes(s), 256) + 4096 * sizeof(float);
}

template <int NPH, int NP, int MODE>
static int launch_fused(const FusedParams& P, hipStream_t st) {
  using L = FusedLds<NPH, NP>;
  auto kernel = fused_fista_kernel<NPH, NP, MODE>;
  static unsigned long long configured = 0;
  if (first_use_on_this_device(&configured)) {
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, L::total));
  }
  const unsigned grid = (unsigned)ceil_div(P.b, kFP);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), L::total, st, P);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

// Diagnostic: VTC_FUSED_STAMPS=1 runs the stamped instantiation (soft
// threshold only) and prints per-segment cycle shares.  Its run time is not
// representative (the stamps fence the schedule); read the shares only.
template <int NPH, int NP>
static int launch_stamped(FusedParams P, hipStream_t st) {
  using L = FusedLds<NPH, NP>;
  auto kernel = fused_fista_kernel<NPH, NP, VTC_SOFT, true>;
  unsigned long long* dev = nullptr;
  VTC_HIP_CHECK(hipMalloc(&dev, 8 * sizeof(unsigned long long)));
  VTC_HIP_CHECK(hipMemsetAsync(dev, 0, 8 * sizeof(unsigned long long), st));
  VTC_HIP_CHECK(hipFuncSetAttribute(
      reinterpret_cast<const void*>(kernel),
      hipFuncAttributeMaxDynamicSharedMemorySize, L::total));
  P.stamps = dev;
  hipLaunchKernelGGL(kernel, dim3((unsigned)ceil_div(P.b, kFP)), dim3(256),
                     L::total, st, P);
  VTC_LAUNCH_CHECK();
  unsigned long long host[8];
  VTC_HIP_CHECK(hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost,
                               st));
  VTC_HIP_CHECK(hipStreamSynchronize(st));
  VTC_HIP_CHECK(hipFree(dev));
  const char* names[5] = {"step1+epi", "epi-alone", "barrier", "step3",
                          "exchange"};
  double total = 0;
  for (int k = 0; k < 5; ++k) total += (double)host[k];
  const double per = (double)host[7] * P.num_iters * NPH;
  for (int k = 0; k < 5; ++k)
    fprintf(stderr, "[vtc stamps] %-9s %5.1f%%  %8.0f cycles/phase/wave\n",
            names[k], 100.0 * host[k] / total, host[k] / per);
  return VTC_OK;
}

template <int NPH, int NP>
static int dispatch_mode(const FusedParams& P, int threshold, hipStream_t st) {
  if (threshold == VTC_SOFT && NPH == 8 && getenv("VTC_FUSED_STAMPS"))
    return launch_stamped<NPH, NP>(P, st);
  switch (threshold) {
    case VTC_SOFT: return launch_fused<NPH, NP, VTC_SOFT>(P, st);
    case VTC_SOFT_NONNEG: return launch_fused<NPH, NP, VTC_SOFT_NONNEG>(P, st);
    case VTC_HARD: return launch_fused<NPH, NP, VTC_HARD>(P, st);
    default: return launch_fused<NPH, NP, VTC_HARD_NONNEG>(P, st);
  }
}

// ---- variant 2 (LDS-staged dictionary, bf16) ------------------------------
static void print_stamps(const unsigned long long* host, int num_iters,
                         int nph, const char* const* names) {
  double total = 0;
  for (int k = 0; k < 5; ++k) total += (double)host[k];
  const double per = (double)host[7] * num_iters * nph;
  for (int k = 0; k < 5; ++k)
    fprintf(stderr, "[vtc stamps] %-9s %5.1f%%  %8.0f cycles/phase/wave\n",
            names[k], 100.0 * host[k] / total, host[k] / per);
}

template <int NPH, int MODE, bool STAMP>
static int launch_lds(FusedParams P, hipStream_t st) {
  auto kernel = fused_fista_lds_kernel<NPH, MODE, STAMP>;
  static unsigned long long configured = 0;
  if (first_use_on_this_device(&configured)) {
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsV2Total));
  }
  unsigned long long* dev = nullptr;
  if (STAMP) {
    VTC_HIP_CHECK(hipMalloc(&dev, 8 * sizeof(unsigned long long)));
    VTC_HIP_CHECK(hipMemsetAsync(dev, 0, 8 * sizeof(unsigned long long), st));
    P.stamps = dev;
  }
  hipLaunchKernelGGL(kernel, dim3((unsigned)ceil_div(P.b, kFP)), dim3(256),
                     kLdsV2Total, st, P);
  VTC_LAUNCH_CHECK();
  if (STAMP) {
    unsigned long long host[8];
    VTC_HIP_CHECK(hipMemcpyAsync(host, dev, sizeof(host),
                                 hipMemcpyDeviceToHost, st));
    VTC_HIP_CHECK(hipStreamSynchronize(st));
    VTC_HIP_CHECK(hipFree(dev));
    const char* names[5] = {"step1", "epilogue", "barriers", "step3",
                            "exchange"};
    print_stamps(host, P.num_iters, NPH, names);
  }
  return VTC_OK;
}

template <int NPH>
static int dispatch_lds_mode(const FusedParams& P, int threshold,
                             hipStream_t st) {
  if (threshold == VTC_SOFT && getenv("VTC_FUSED_STAMPS"))
    return launch_lds<NPH, VTC_SOFT, true>(P, st);
  switch (threshold) {
    case VTC_SOFT: return launch_lds<NPH, VTC_SOFT, false>(P, st);
    case VTC_SOFT_NONNEG:
      return launch_lds<NPH, VTC_SOFT_NONNEG, false>(P, st);
    case VTC_HARD: return launch_lds<NPH, VTC_HARD, false>(P, st);
    default: return launch_lds<NPH, VTC_HARD_NONNEG, false>(P, st);
  }
}

static int dispatch_lds(const FusedParams& P, int threshold, hipStream_t st) {
  switch (phases_for(P.s)) {
    case 2: return dispatch_lds_mode<2>(P, threshold, st);
    case 4: return dispatch_lds_mode<4>(P, threshold, st);
    case 8: return dispatch_lds_mode<8>(P, threshold, st);
  }
  set_error("fused FISTA: unsupported atom count %d", P.s);
  return VTC_ERR_UNSUPPORTED;
}

// ---- variant 3 (private transposition, bf16) ------------------------------
template <int NPH, int MODE, bool STAMP>
static int launch_priv(FusedParams P, hipStream_t st) {
  auto kernel = fused_fista_priv_kernel<NPH, MODE, STAMP>;
  static unsigned long long configured = 0;
  if (first_use_on_this_device(&configured)) {
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, PrivLds<NPH>::total));
  }
  unsigned long long* dev = nullptr;
  if (STAMP) {
    VTC_HIP_CHECK(hipMalloc(&dev, 8 * sizeof(unsigned long long)));
    VTC_HIP_CHECK(hipMemsetAsync(dev, 0, 8 * sizeof(unsigned long long), st));
    P.stamps = dev;
  }
  hipLaunchKernelGGL(kernel, dim3((unsigned)ceil_div(P.b, kFP)), dim3(256),
                     PrivLds<NPH>::total, st, P);
  VTC_LAUNCH_CHECK();
  if (STAMP) {
    unsigned long long host[8];
    VTC_HIP_CHECK(hipMemcpyAsync(host, dev, sizeof(host),
                                 hipMemcpyDeviceToHost, st));
    VTC_HIP_CHECK(hipStreamSynchronize(st));
    VTC_HIP_CHECK(hipFree(dev));
    const char* names[5] = {"step1", "epilogue", "-", "step3", "reduce-R"};
    print_stamps(host, P.num_iters, NPH, names);
  }
  return VTC_OK;
}

template <int NPH>
static int dispatch_priv_mode(const FusedParams& P, int threshold,
                              hipStream_t st) {
  if (threshold == VTC_SOFT && getenv("VTC_FUSED_STAMPS"))
    return launch_priv<NPH, VTC_SOFT, true>(P, st);
  switch (threshold) {
    case VTC_SOFT: return launch_priv<NPH, VTC_SOFT, false>(P, st);
    case VTC_SOFT_NONNEG:
      return launch_priv<NPH, VTC_SOFT_NONNEG, false>(P, st);
    case VTC_HARD: return launch_priv<NPH, VTC_HARD, false>(P, st);
    default: return launch_priv<NPH, VTC_HARD_NONNEG, false>(P, st);
  }
}

static int dispatch_priv(const FusedParams& P, int threshold, hipStream_t st) {
  switch (phases_for(P.s)) {
    case 2: return dispatch_priv_mode<2>(P, threshold, st);
    case 4: return dispatch_priv_mode<4>(P, threshold, st);
    case 8: return dispatch_priv_mode<8>(P, threshold, st);
  }
  set_error("fused FISTA: unsupported atom count %d", P.s);
  return VTC_ERR_UNSUPPORTED;
}

template <int NP>
static int dispatch_phases(const FusedParams& P, int threshold,
                           hipStream_t st) {
  switch (phases_for(P.s)) {
    case 2: return dispatch_mode<2, NP>(P, threshold, st);
    case 4: return dispatch_mode<4, NP>(P, threshold, st);
    case 8: return dispatch_mode<8, NP>(P, threshold, st);
  }
  set_error("fused FISTA: unsupported atom count %d", P.s);
  return VTC_ERR_UNSUPPORTED;
}

int run_fused(const float* images, const float* dictionary,
              const float* initial_codes, float* codes, int64_t b, int64_t n,
              int64_t s, float eta, float cutoff, int num_iters, int variant,
              int threshold, int precision, void* workspace,
              size_t workspace_bytes, int* iters_run, hipStream_t st) {
  if (!fused_shape_supported(b, n, s, precision)) {
    set_error("fused FISTA: unsupported shape");
    return VTC_ERR_UNSUPPORTED;
  }
  if (num_iters > 4096) {
    set_error("fused FISTA: at most 4096 iterations per call");
    return VTC_ERR_UNSUPPORTED;
  }
  if (!workspace ||
      workspace_bytes < fused_workspace_bytes(b, n, s, precision)) {
    set_error("fused FISTA: workspace too small");
    return VTC_ERR_WORKSPACE;
  }
  const int parts = (precision == VTC_BF16X3) ? 2 : 1;
  Carver ws(workspace);
  FusedParams P;
  __bf16* packs[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int part = 0; part < parts; ++part) {
    packs[2 * part] = ws.take<__bf16>((size_t)s * kFN);
    packs[2 * part + 1] = ws.take<__bf16>((size_t)s * kFN);
  }
  float* betas_dev = ws.take<float>(4096);
  // bf16: VTC_FUSED_VARIANT=2 selects the experimental LDS-staged variant
  // (same results bit for bit; slower on MI355X, see its header); the
  // register-ring variant is the default and the only one for bf16x3
  const char* variant_env = getenv("VTC_FUSED_VARIANT");
  const bool use_lds_variant =
      (parts == 1) && variant_env && variant_env[0] == '2';
  if (use_lds_variant) {
    hipLaunchKernelGGL(pack_lds_image_kernel, dim3(256), dim3(256), 0, st,
                       dictionary, (int)s, packs[0]);
    VTC_LAUNCH_CHECK();
  } else {
    for (int part = 0; part < parts; ++part) {
      hipLaunchKernelGGL(pack_dictionary_kernel, dim3(256), dim3(256), 0, st,
                         dictionary, (int)s, packs[2 * part],
                         packs[2 * part + 1], part);
      VTC_LAUNCH_CHECK();
    }
  }
  // ISTA is FISTA with beta = 0: y = c + 0 * (c - c_prev) = c exactly
  std::vector<float> betas;
  fista_betas(num_iters, &betas);
  if (variant == VTC_ISTA)
    for (auto& v : betas) v = 0.f;
  // the staging buffer must outlive the async copy: keep it per thread
  static thread_local std::vector<float> staged;
  staged = betas;
  VTC_HIP_CHECK(hipMemcpyAsync(betas_dev, staged.data(),
                               sizeof(float) * num_iters,
                               hipMemcpyHostToDevice, st));
  P.images = images;
  P.init = initial_codes;
  P.codes = codes;
  for (int part = 0; part < 2; ++part) {
    P.packA[part] = reinterpret_cast<const uint4*>(packs[2 * (part % parts)]);
    P.packT[part] =
        reinterpret_cast<const uint4*>(packs[2 * (part % parts) + 1]);
  }
  P.betas = betas_dev;
  P.b = b;
  P.s = (int)s;
  P.num_iters = num_iters;
  P.eta = eta;
  P.cutoff = cutoff;
  P.stamps = nullptr;
  const bool use_priv_variant =
      (parts == 1) && variant_env && variant_env[0] == '3';
  int rc = use_lds_variant    ? dispatch_lds(P, threshold, st)
           : use_priv_variant ? dispatch_priv(P, threshold, st)
           : (parts == 2)     ? dispatch_phases<2>(P, threshold, st)
                              : dispatch_phases<1>(P, threshold, st);
  if (rc == VTC_OK && iters_run) *iters_run = num_iters;
  return rc;
}

}  // namespace vtc
