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
//
// F16 (with NP = 2): the same three products on an f16 hi/lo split -- 11 + 11
// significand bits instead of 8 + 8, i.e. ~2^-21 per product at the same MFMA
// rate and the same bytes.  f16 has 5 exponent bits, so the kernel works in
// power-of-two scaled units: the dictionary is packed as sigma_D * D (one
// global sigma_D, max |D| in [256, 512)), and each patch carries its own
// sigma_Y = 2^(8 - e), e = exponent of max(||x||, ||y0||), so that the
// residual and the codes sit around 2^9 whatever the data's magnitude; the
// f32 state (Y, C, X) is kept in those units for the whole launch.  Scaling by
// a power of two commutes with every IEEE operation of the epilogue (no
// under/overflow at these magnitudes), so the iteration is the reference's
// own, bit for bit, given the products; codes are scaled back on the way out.
#include "fc_fused.h"
#include "fused_common.h"

#include <stdlib.h>
#include <mutex>
#include <vector>

namespace vtc {

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
  int fista;              // 0: ISTA, the gradient is evaluated at the codes
  float eta, cutoff;      // used when eta_dev is null
  const float* eta_dev;   // device scalar 1/L (vtc_lambda_max out[1]) or null
  float lam;              // sparsity weight, cutoff = lam * *eta_dev
  const float* dscale;    // F16: {sigma_D, 1 / sigma_D} (dictionary_scale_kernel)
  unsigned long long* stamps;  // diagnostic build only: 8 cycle sums
};

#define VTC_MFMA(a, b, c) mfma_frag<F16>(a, b, c)

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
  static constexpr int stat_bytes = 2 * 4 * 32 * 4;   // F16: residual maxima, [call parity][wave][patch]
  static constexpr int total = cst_bytes + yx_bytes + rx_bytes + stat_bytes;
};

template <int NPH, int NP, int MODE, bool F16 = false, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void fused_fista_kernel(FusedParams P) {
  static_assert(!F16 || NP == 2, "the f16 split exists as three products only");
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
  float* Stat = reinterpret_cast<float*>(Rx + L::rx_bytes);

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
    }
  }
  // F16: the patch's power-of-two scale (header comment).  e = exponent of
  // max(||x||, ||y0||) over the whole patch: partial sums of squares per lane,
  // the two lane halves by a lane exchange, the four waves through LDS (the R
  // exchange area is not in use yet).
  float sigma_y = 1.f, inv_sigma_y = 1.f;   // per lane (= per patch)
  float sigma_d = 1.f, inv_sigma_d = 1.f;
  if (F16) {
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) sx += Xr[nb][e] * Xr[nb][e];
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
    const float sx_scale = sigma_d * sigma_y;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) Xr[nb][e] *= sx_scale;
#pragma unroll
    for (int p = 0; p < NPH; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) Y[p][e] *= sigma_y;
  }
  // eta either came by value or sits in device memory (sync-free callers);
  // lambda * eta is then the same single f32 multiply the host would do
  float eta = P.eta, cutoff_l = P.cutoff;
  if (P.eta_dev) {
    eta = *P.eta_dev;
    cutoff_l = mul_rn(P.lam, eta);
  }
  if (F16) {
    // scaled units: eta_s * Gacc = sigma_Y * (eta * G) with Gacc = sigma_D
    // sigma_R G, sigma_R = 2 sigma_Y; the threshold scales with the codes
    // (per lane, i.e. per patch)
    eta = eta * (0.5f * inv_sigma_d);
    cutoff_l = cutoff_l * sigma_y;
  }
  // residual operand: Rs = (Racc - Xs) * r_scale = sigma_R * R with
  // sigma_R = 2 sigma_Y (||Rs|| starts in [2^12, 2^13))
  const float r_scale = F16 ? 2.f * inv_sigma_d : 1.f;
  int xr_calls = 0;
#pragma unroll
  for (int p = 0; p < NPH; ++p) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (p < CREG) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          Cr[p < CREG ? p : 0][4 * g + q] = Y[p][4 * g + q];
      } else {
        *reinterpret_cast<float4*>(Cst + cst_ln + (p - CREG) * 16384 +
                                   g * 1024) =
            make_float4(Y[p][4 * g + 0], Y[p][4 * g + 1], Y[p][4 * g + 2],
                        Y[p][4 * g + 3]);
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
      const float v4[4] = {y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3]};
      uint2 hi, lo;
      split4<F16, NP>(v4, &hi, &lo);
      char* dst = Yx + buf * NP * kYxPart + yx_wr + 16 * g;
      *reinterpret_cast<uint2*>(dst) = hi;
      if (NP == 2) *reinterpret_cast<uint2*>(dst + kYxPart) = lo;
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
  auto step3 = [&](int p, int buf, bool pipe, int sg, auto&& between) {
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
        // (epilogue arithmetic of another phase, if any, before the refill:
        // see step 1)
        between(i);
        if (pipe) VTC_REFILL(sg, i)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto nothing = [](int) {};

  // R_{k+1} = Racc - X  ->  bf16 parts -> LDS (read back as B fragments by
  // every wave during step 1 of the next iteration)
  auto exchange_r = [&]() {
    float v[2][16];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        v[nb][e] = sub_rn(Racc[nb][e], Xr[nb][e]);
        if (F16) v[nb][e] *= r_scale;
      }
    if (F16) {
      // f16 range guard.  With eta = 1/L the residual never grows past a small
      // multiple of ||x|| (2^9 .. 2^10 in these units), but a caller's own
      // stepsize may make the iteration diverge -- as it does in f32 in the
      // reference.  When a patch's residual has passed 2^11 all of its state
      // drops by a power of two (exact, so the iteration goes on bit for bit as
      // before, in smaller units; growth of up to 32x per iteration is
      // absorbed before f16 overflows): max over the patch through LDS, then a
      // wave-uniform branch that is not taken in a convergent run.
      float m = 0.f;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) m = fmaxf(m, fabsf(v[nb][e]));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      // the decision uses the maxima all four waves left at the PREVIOUS call
      // (visible since that call's closing barrier): no barrier of its own
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
          for (int e = 0; e < 16; ++e) Y[p][e] *= f;
#pragma unroll
        for (int p = 0; p < CR; ++p)
#pragma unroll
          for (int e = 0; e < 16; ++e) Cr[p][e] *= f;
#pragma unroll
        for (int pl = 0; pl < L::CL; ++pl)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float4* c4 = reinterpret_cast<float4*>(Cst + cst_ln + pl * 16384 +
                                                   g * 1024);
            float4 c = *c4;
            c.x *= f; c.y *= f; c.z *= f; c.w *= f;
            *c4 = c;
          }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            Xr[nb][e] *= f;
            v[nb][e] *= f;
          }
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
        if (NP == 2) *reinterpret_cast<uint2*>(dst + kRxPart) = lo;
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
      step3(p, p & 1, false, 0, nothing);
    }
  }
  exchange_r();

  // prime the ring with the first RING fragments of segment 0
#pragma unroll
  for (int i = 0; i < RING; ++i)
#pragma unroll
    for (int part = 0; part < NP; ++part)
      ring[part][i] = VTC_LOAD_SEG(part, 0, i);

  const bool fista = P.fista != 0;   // ISTA: every beta is 0
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
    if (MODE == 7) {         // diagnostic: step 1 with (almost) no epilogue work
      const float cn = Y[p][e] + Gp[e];
      Y[p][e] = cn;
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
      }
      return;
    }
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
    const float cn = shrink_fast<MODE>(c, cutoff_l);
    // ISTA: y = codes (ista_fista.py:133), not codes + 0 * (codes - old):
    // a code that has overflowed to inf stays inf as it does there
    Y[p][e] = fista ? add_rn(cn, mul_rn(beta, sub_rn(cn, co))) : cn;
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
      const float v4[4] = {Y[p][4 * g], Y[p][4 * g + 1], Y[p][4 * g + 2],
                           Y[p][4 * g + 3]};
      uint2 hi, lo;
      split4<F16, NP>(v4, &hi, &lo);
      char* dst = Yx + (p & 1) * NP * kYxPart + yx_wr + 16 * g;
      *reinterpret_cast<uint2*>(dst) = hi;
      if (NP == 2) *reinterpret_cast<uint2*>(dst + kYxPart) = lo;
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
    const float beta = fista ? P.betas[it] : 0.f;
    step1(0, 0, false, beta);
    VTC_STAMP(0)
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
      if (p + 1 < NPH) {
        step1(p + 1, 2 * p + 1, true, beta);   // + epilogue of phase p
        VTC_STAMP(0)
      }
      // (the epilogue of the LAST phase has no step 1 to hide under: it rides
      // on step 3 of the phase before it -- its gradient tile is complete by
      // then, and it writes the exchange buffer step 3 is not reading)
      __syncthreads();
      VTC_STAMP(2)
      // ---- step 3: next residual, this wave's 64 pixels
      if (p + 2 == NPH) {
        step3(p, p & 1, true, 2 * p + 2, [&](int i) {
          epilogue_elem(NPH - 1, i, Gb[(NPH - 1) & 1], beta);
          __builtin_amdgcn_sched_barrier(0);
        });
      } else {
        step3(p, p & 1, true, (p + 1 < NPH) ? 2 * p + 2 : 2 * NPH - 1, nothing);
      }
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
#undef VTC_LOAD_A
#undef VTC_LOAD_T
#undef VTC_LOAD_SEG
#undef VTC_REFILL
#undef VTC_SEG_IS_T
#undef VTC_SEG_PHASE
}

}  // namespace vtc

// Experiments kept out of the library (profiles/r03_fused_onepacking.txt).
// Eight waves per workgroup (two per SIMD, 16-atom step-1 tiles): build with
// -DVTC_EXPERIMENT_EIGHT_WAVES, run with VTC_FUSED_8W=1.
#ifdef VTC_EXPERIMENT_EIGHT_WAVES
#include "../../tools/micro/fc_fused8.h"
#endif

// Experiment kept out of the library (profiles/r03_fused_onepacking.txt): a
// variant that keeps part of each phase's dictionary tile in LDS and reads it
// back transposed instead of streaming packT.  Build with
// -DVTC_EXPERIMENT_ONE_PACKING, run with VTC_FUSED_1P=2|3.
#ifdef VTC_EXPERIMENT_ONE_PACKING
#include "../../tools/micro/fc_fused1p.h"
#endif

namespace vtc {

// -------------------------------------------------------------------- host
static int phases_for(int64_t s) { return (int)(s / kPhaseAtoms); }

bool fused_shape_supported(int64_t b, int64_t n, int64_t s, int precision) {
  if (n != kFN || b <= 0) return false;
  if (s % kPhaseAtoms != 0) return false;
  const int nph = phases_for(s);
  if (!(nph == 2 || nph == 4 || nph == 8)) return false;
  return precision == VTC_BF16 || precision == VTC_BF16X3 ||
         precision == VTC_F16X3;
}

static size_t pack_bytes(int64_t s) { return (size_t)s * kFN * 2; }

size_t fused_workspace_bytes(int64_t b, int64_t n, int64_t s, int precision) {
  if (!fused_shape_supported(b, n, s, precision)) return 256;
  const int parts = (precision == VTC_BF16) ? 1 : 2;
  return (size_t)parts * 2 * align_up(pack_bytes(s), 256) + 256;
}

template <int NPH, int NP, int MODE, bool F16>
static int launch_fused(const FusedParams& P, hipStream_t st) {
  using L = FusedLds<NPH, NP>;
  auto kernel = fused_fista_kernel<NPH, NP, MODE, F16>;
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
template <int NPH, int NP, bool F16>
static int launch_stamped(FusedParams P, hipStream_t st) {
  using L = FusedLds<NPH, NP>;
  static const bool no_epilogue = getenv("VTC_FUSED_NOEPI") != nullptr;
  auto kernel = no_epilogue ? fused_fista_kernel<NPH, NP, 7, F16, true>
                            : fused_fista_kernel<NPH, NP, VTC_SOFT, F16, true>;
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

#ifdef VTC_EXPERIMENT_ONE_PACKING
// One-packing variant (fc_fused1p.h), three-product modes.
template <int NPH, int MODE, bool F16, int NT>
static int launch_fused1p(const FusedParams& P, hipStream_t st) {
  using L = Fused1pLds<NT>;
  auto kernel = fused1p_kernel<NPH, MODE, F16, NT>;
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

template <int NPH, bool F16, int NT>
static int launch_stamped1p(FusedParams P, hipStream_t st) {
  using L = Fused1pLds<NT>;
  auto kernel = fused1p_kernel<NPH, VTC_SOFT, F16, NT, true>;
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
  const char* names[5] = {"step1", "epilogue", "barrier1", "step3",
                          "barrier2+x"};
  double total = 0;
  for (int k = 0; k < 5; ++k) total += (double)host[k];
  const double per = (double)host[7] * P.num_iters * NPH;
  for (int k = 0; k < 5; ++k)
    fprintf(stderr, "[vtc stamps 1p] %-10s %5.1f%%  %8.0f cycles/phase/wave\n",
            names[k], 100.0 * host[k] / total, host[k] / per);
  return VTC_OK;
}

template <int NPH, bool F16>
static int dispatch_mode1p(const FusedParams& P, int threshold, int nt,
                           hipStream_t st) {
  static const bool stamps = getenv("VTC_FUSED_STAMPS") != nullptr;
  if (stamps && threshold == VTC_SOFT && NPH == 8)
    return nt == 2 ? launch_stamped1p<NPH, F16, 2>(P, st)
                   : launch_stamped1p<NPH, F16, 3>(P, st);
#define VTC1_CASE(M)                                              \
  case M:                                                         \
    return nt == 2 ? launch_fused1p<NPH, M, F16, 2>(P, st)        \
                   : launch_fused1p<NPH, M, F16, 3>(P, st);
  switch (threshold) {
    VTC1_CASE(VTC_SOFT)
    VTC1_CASE(VTC_SOFT_NONNEG)
    VTC1_CASE(VTC_HARD)
    default:
      return nt == 2 ? launch_fused1p<NPH, VTC_HARD_NONNEG, F16, 2>(P, st)
                     : launch_fused1p<NPH, VTC_HARD_NONNEG, F16, 3>(P, st);
  }
#undef VTC1_CASE
}

#endif

#ifdef VTC_EXPERIMENT_EIGHT_WAVES
// Eight-wave form (tools/micro/fc_fused8.h), three-product modes.
static bool fused_eight_waves() {
  static const bool on = [] {
    const char* v = getenv("VTC_FUSED_8W");
    return v != nullptr && atoi(v) != 0;
  }();
  return on;
}

template <int NPH, int MODE, bool F16>
static int launch_fused8(const FusedParams& P, hipStream_t st) {
  using L = Fused8Lds<NPH>;
  auto kernel = fused8_kernel<NPH, MODE, F16>;
  static unsigned long long configured = 0;
  if (first_use_on_this_device(&configured)) {
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, L::total));
  }
  const unsigned grid = (unsigned)ceil_div(P.b, kFP);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), L::total, st, P);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

template <int NPH, bool F16>
static int launch_stamped8(FusedParams P, hipStream_t st) {
  using L = Fused8Lds<NPH>;
  auto kernel = fused8_kernel<NPH, VTC_SOFT, F16, true>;
  unsigned long long* dev = nullptr;
  VTC_HIP_CHECK(hipMalloc(&dev, 8 * sizeof(unsigned long long)));
  VTC_HIP_CHECK(hipMemsetAsync(dev, 0, 8 * sizeof(unsigned long long), st));
  VTC_HIP_CHECK(hipFuncSetAttribute(
      reinterpret_cast<const void*>(kernel),
      hipFuncAttributeMaxDynamicSharedMemorySize, L::total));
  P.stamps = dev;
  hipLaunchKernelGGL(kernel, dim3((unsigned)ceil_div(P.b, kFP)), dim3(512),
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
    fprintf(stderr, "[vtc stamps 8w] %-9s %5.1f%%  %8.0f cycles/phase/wave\n",
            names[k], 100.0 * host[k] / total, host[k] / per);
  fprintf(stderr, "[vtc stamps 8w] barrier wait of waves 0-3: %.0f, of waves "
          "4-7: %.0f cycles/phase/wave\n", 2.0 * host[5] / per,
          2.0 * host[6] / per);
  return VTC_OK;
}

template <int NPH, bool F16>
static int dispatch_mode8(const FusedParams& P, int threshold, hipStream_t st) {
  static const bool stamps = getenv("VTC_FUSED_STAMPS") != nullptr;
  if (stamps && threshold == VTC_SOFT && NPH == 8)
    return launch_stamped8<NPH, F16>(P, st);
  switch (threshold) {
    case VTC_SOFT: return launch_fused8<NPH, VTC_SOFT, F16>(P, st);
    case VTC_SOFT_NONNEG: return launch_fused8<NPH, VTC_SOFT_NONNEG, F16>(P, st);
    case VTC_HARD: return launch_fused8<NPH, VTC_HARD, F16>(P, st);
    default: return launch_fused8<NPH, VTC_HARD_NONNEG, F16>(P, st);
  }
}
#endif

template <int NPH, int NP, bool F16>
static int dispatch_mode(const FusedParams& P, int threshold, hipStream_t st) {
  static const bool stamps = getenv("VTC_FUSED_STAMPS") != nullptr;
#ifdef VTC_EXPERIMENT_EIGHT_WAVES
  if (NP == 2 && fused_eight_waves())
    return dispatch_mode8<NPH, F16>(P, threshold, st);
#endif
#ifdef VTC_EXPERIMENT_ONE_PACKING
  static const int one_packing =
      getenv("VTC_FUSED_1P") ? atoi(getenv("VTC_FUSED_1P")) : 0;
  if (NP == 2 && one_packing >= 2)
    return dispatch_mode1p<NPH, F16>(P, threshold, one_packing, st);
#endif
  if (stamps && threshold == VTC_SOFT && NPH == 8 && NP == 2)
    return launch_stamped<NPH, NP, F16>(P, st);
  switch (threshold) {
    case VTC_SOFT: return launch_fused<NPH, NP, VTC_SOFT, F16>(P, st);
    case VTC_SOFT_NONNEG:
      return launch_fused<NPH, NP, VTC_SOFT_NONNEG, F16>(P, st);
    case VTC_HARD: return launch_fused<NPH, NP, VTC_HARD, F16>(P, st);
    default: return launch_fused<NPH, NP, VTC_HARD_NONNEG, F16>(P, st);
  }
}

template <int NP, bool F16>
static int dispatch_phases(const FusedParams& P, int threshold,
                           hipStream_t st) {
  switch (phases_for(P.s)) {
    case 2: return dispatch_mode<2, NP, F16>(P, threshold, st);
    case 4: return dispatch_mode<4, NP, F16>(P, threshold, st);
    case 8: return dispatch_mode<8, NP, F16>(P, threshold, st);
  }
  set_error("fused FISTA: unsupported atom count %d", P.s);
  return VTC_ERR_UNSUPPORTED;
}

// The FISTA momentum table beta_k = (t_k - 1) / t_{k+1} (ista_fista.py:123-127,
// float64 recurrence rounded to f32) does not depend on the call: one copy per
// device, placed by vtc_init() -- or by the first inference call on a device
// nobody prepared (one hipMalloc and a blocking 64 KiB copy, under a lock);
// afterwards a call only enqueues.
constexpr int kBetaTable = 16384;
const float* fista_beta_table_on_this_device() {
  static float* tables[64] = {nullptr};
  static std::mutex guard;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return nullptr;
  std::lock_guard<std::mutex> lock(guard);
  if (tables[dev]) return tables[dev];
  std::vector<float> host;
  fista_betas(kBetaTable, &host);
  float* table = nullptr;
  if (hipMalloc(&table, sizeof(float) * kBetaTable) != hipSuccess)
    return nullptr;
  if (hipMemcpy(table, host.data(), sizeof(float) * kBetaTable,
                hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(table);
    return nullptr;
  }
  tables[dev] = table;
  return table;
}

int fused_max_iters() { return kBetaTable; }

int run_fused(const float* images, const float* dictionary,
              const float* initial_codes, float* codes, int64_t b, int64_t n,
              int64_t s, float eta, const float* eta_dev,
              float sparsity_weight, int num_iters, int variant,
              int threshold, int precision, void* workspace,
              size_t workspace_bytes, int* iters_run, hipStream_t st) {
  if (!fused_shape_supported(b, n, s, precision)) {
    set_error("fused FISTA: unsupported shape");
    return VTC_ERR_UNSUPPORTED;
  }
  if (num_iters > kBetaTable) {
    set_error("fused FISTA: at most %d iterations per call", kBetaTable);
    return VTC_ERR_UNSUPPORTED;
  }
  if (!workspace ||
      workspace_bytes < fused_workspace_bytes(b, n, s, precision)) {
    set_error("fused FISTA: workspace too small");
    return VTC_ERR_WORKSPACE;
  }
  const float* betas_dev = fista_beta_table_on_this_device();
  if (!betas_dev) {
    set_error("fused FISTA: could not place the momentum table on the device");
    return VTC_ERR_HIP;
  }
  const int parts = (precision == VTC_BF16) ? 1 : 2;
  const bool f16 = (precision == VTC_F16X3);
  Carver ws(workspace);
  FusedParams P;
  unsigned short* packs[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int part = 0; part < parts; ++part) {
    packs[2 * part] = ws.take<unsigned short>((size_t)s * kFN);
    packs[2 * part + 1] = ws.take<unsigned short>((size_t)s * kFN);
  }
  float* dscale = ws.take<float>(2);
  if (f16) {
    hipLaunchKernelGGL(dictionary_scale_kernel, dim3(1), dim3(1024), 0, st,
                       dictionary, (int64_t)s * kFN, dscale);
    VTC_LAUNCH_CHECK();
  }
  if (f16)
    hipLaunchKernelGGL(pack_dictionary_kernel<true>, dim3(256), dim3(256), 0,
                       st, dictionary, (int)s, packs[0], packs[1], packs[2],
                       packs[3], dscale);
  else
    hipLaunchKernelGGL(pack_dictionary_kernel<false>, dim3(256), dim3(256), 0,
                       st, dictionary, (int)s, packs[0], packs[1], packs[2],
                       packs[3], dscale);
  VTC_LAUNCH_CHECK();
#ifdef VTC_EXPERIMENT_EIGHT_WAVES
  if (parts == 2 && fused_eight_waves()) {
    // the eight-wave kernel reads its step-1 operand as 16-atom tiles
    if (f16)
      hipLaunchKernelGGL(pack_dictionary8_kernel<true>, dim3(256), dim3(256), 0,
                         st, dictionary, (int)s, packs[0], packs[2], dscale);
    else
      hipLaunchKernelGGL(pack_dictionary8_kernel<false>, dim3(256), dim3(256),
                         0, st, dictionary, (int)s, packs[0], packs[2], dscale);
    VTC_LAUNCH_CHECK();
  }
#endif
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
  P.fista = (variant == VTC_FISTA) ? 1 : 0;
  P.eta = eta;
  P.eta_dev = eta_dev;
  P.lam = sparsity_weight;
  // lambda * eta: the Python float rounded to f32, then one f32 multiply
  P.cutoff = sparsity_weight * eta;
  P.dscale = dscale;
  P.stamps = nullptr;
  int rc = f16          ? dispatch_phases<2, true>(P, threshold, st)
           : parts == 2 ? dispatch_phases<2, false>(P, threshold, st)
                        : dispatch_phases<1, false>(P, threshold, st);
  if (rc == VTC_OK && iters_run) *iters_run = num_iters;
  return rc;
}

}  // namespace vtc

extern "C" int vtc_init(void) {
  if (!vtc::fista_beta_table_on_this_device()) {
    vtc::set_error("vtc_init: could not place the momentum table on the device");
    return VTC_ERR_HIP;
  }
  return VTC_OK;
}
