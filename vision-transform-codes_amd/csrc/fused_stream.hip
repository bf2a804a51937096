// Fused persistent ISTA/FISTA kernel with STREAMED state, 16x16 patches.
//
// fc_fused.hip keeps the whole per-patch state (Y and the previous codes of 32
// patches x s atoms) on chip, which caps s at 1024.  Here s is unbounded: the
// subspace plugin's padded dictionary of configs[3] has 4096 slots, i.e.
// 32 KiB of f32 state per patch and iteration.  What stays resident for all
// the iterations of a launch is what couples the slots: the residual R
// (32 patches x 256 pixels, exchanged through LDS) and the patches X.  The
// codes stream through once per iteration:
//
//   for every phase of 128 slots (one 32-slot tile per wave):
//     step 1    G  = D[tile] R_k                       (MFMA, as fc_fused.hip)
//     epilogue  y  = c_k + beta_{k-1} (c_k - c_{k-1})   (recomputed, see below)
//               p  = y - eta G
//               c' = prox(p)                element-wise, or the group soft
//                                           threshold over m adjacent slots
//                                           (subspace_ista_fista.py:144-156)
//               y' = c' + beta_k (c' - c_k)
//               c' -> memory (over c_{k-1});  y' -> bf16/f16 parts -> LDS
//     barrier
//     step 3    Racc[this wave's pixels] += D[phase]^T y'     (MFMA)
//   R_{k+1} = Racc - X  -> LDS
//
// State traffic.  The reference's loop keeps y and the codes (two (b, slots)
// arrays read and written per iteration: 16 B per slot); here only the last
// two code iterates c_k, c_{k-1} are kept and y is recomputed from them with
// the very operations that produced it (bit-identical), so an iteration reads
// 8 B and writes 4 B per slot: 12 B, the floor for a FISTA state that does not
// fit on chip.  Both iterates live in the caller's workspace in FRAGMENT ORDER
// [workgroup][phase][wave][group of 4 slots][lane] so that every access is a
// full 1 KiB wave instruction; the (b, slots) layout of the caller is only
// touched by the two small conversion kernels around the launch.
//
// The dictionary (bf16 / f16 hi and lo parts, both fragment packings: 8 MiB at
// 4096 slots) streams from L2 / Infinity Cache twice per iteration exactly as
// in fc_fused.hip, through the same register ring; segment order per iteration
//   A(0) | A(1) T(0) | A(2) T(1) | ... | A(N-1) T(N-2) | T(N-1)
// with the epilogue of phase q issued under step 1 of phase q+1.
#include "fused_stream.h"

#include <stdlib.h>

#include "fc_fused.h"
#include "fused_common.h"

namespace vtc {

constexpr int kSYxRow = 272;            // Y' exchange row: 256 B + 16 B pad
constexpr int kSRxRow = 528;            // R exchange row: 512 B + 16 B pad
constexpr int kSYxPart = 32 * kSYxRow;  // 8704
constexpr int kSRxPart = 32 * kSRxRow;  // 16896
constexpr int kSLds = 2 * 2 * kSYxPart + 2 * kSRxPart + 1024;
constexpr int kSRing = 8;

struct StreamParams {
  const float* images;     // (b, 256)
  float4* state[2];        // fragment-order code iterates: [0] = c_0, [1] = c_-1
  const void* pack[2];     // [hi, lo]; each: packA bytes, then packT bytes
  unsigned pack_half;      // bytes of one packing = slots * 256 * 2
  const float* betas;
  float* patch_scale;      // (b) out: factor that brings the stored codes back
  int64_t b;
  int nph;                 // phases of 128 slots (even)
  int num_iters;
  int fista;
  float eta, cutoff;
  const float* eta_dev;
  float lam;
  const float* dscale;
  unsigned long long* stamps;   // diagnostic: 8 cycle sums
  int debug;               // diagnostic: 32 = accumulate phase stamps (VTC_STREAM_DEBUG)
};

// ---- conversions between the caller's (b, slots) rows and fragment order ---
// element (patch 32 wg + r, slot 128 q + 32 w + 8 g + 4 h + k) lives in float4
// number (((wg nph + q) 4 + w) 4 + g) 64 + (32 h + r), component k
__global__ void stream_state_pack_kernel(const float* __restrict__ rows,
                                         float4* __restrict__ frag0,
                                         float4* __restrict__ frag1,
                                         int64_t b, int nph) {
  const int64_t total = (ceil_div_dev(b, 32)) * nph * 1024;
  const int64_t slots = (int64_t)nph * 128;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total;
       u += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(u & 63), g = (int)((u >> 6) & 3);
    const int w = (int)((u >> 8) & 3);
    const int64_t wq = u >> 10;
    const int q = (int)(wq % nph);
    const int64_t wg = wq / nph;
    const int r = lane & 31, h = lane >> 5;
    const int64_t patch = wg * 32 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rows && patch < b)
      v = *reinterpret_cast<const float4*>(rows + patch * slots + 128 * q +
                                           32 * w + 8 * g + 4 * h);
    frag0[u] = v;
    frag1[u] = v;
  }
}

__global__ void stream_state_unpack_kernel(const float4* __restrict__ frag,
                                           const float* __restrict__ scale,
                                           float* __restrict__ rows, int64_t b,
                                           int nph) {
  const int64_t total = (ceil_div_dev(b, 32)) * nph * 1024;
  const int64_t slots = (int64_t)nph * 128;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total;
       u += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(u & 63), g = (int)((u >> 6) & 3);
    const int w = (int)((u >> 8) & 3);
    const int64_t wq = u >> 10;
    const int q = (int)(wq % nph);
    const int64_t wg = wq / nph;
    const int r = lane & 31, h = lane >> 5;
    const int64_t patch = wg * 32 + r;
    if (patch >= b) continue;
    float4 v = frag[u];
    const float f = scale[patch];
    v.x *= f; v.y *= f; v.z *= f; v.w *= f;
    *reinterpret_cast<float4*>(rows + patch * slots + 128 * q + 32 * w + 8 * g +
                               4 * h) = v;
  }
}

// x[l] + x[l ^ 32] on every lane, through v_permlane32_swap (no LDS)
__device__ __forceinline__ float add_across_halves(float x) {
  const unsigned u = __float_as_uint(x);
  const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return add_rn(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}

#define VTC_MFMA(a, b, c) mfma_frag<F16>(a, b, c)

typedef __attribute__((address_space(3))) void lds_void_t;

// Workgroup = 8 waves, two per SIMD: waves 0-3 compute (MFMA + epilogue, one
// 32-slot tile each, as in fc_fused.hip), waves 4-7 MOVE STATE: wave 4 + w
// copies the c_k / c_{k-1} tiles wave w will need from memory into an LDS slot
// by LDS-DMA, one to two phases ahead.  Why separate waves: a wave's vector
// memory operations retire in order (one vmcnt counter), so a compute wave that
// issued the HBM-latency state loads itself had every younger -- L2-resident --
// dictionary fragment wait behind them: 12 000 cycles per phase instead of
// 5 000 (measured, first version of this kernel).  The same holds for the c'
// stores, so they leave through the mover waves as well; the compute waves'
// vector-memory stream is the dictionary ring and nothing else.
//
// M: group size (1 = element-wise threshold MODE; 2, 4, 8 = group soft
// threshold over M adjacent slots).  Both operands split hi/lo, 3 products.
constexpr int kSSlotBytes = 2 * 4 * 4096;     // [c_k | c_{k-1}][wave][g][lane]
constexpr int kSSlots = 2;
constexpr int kSLdsTotal = kSLds + kSSlots * kSSlotBytes + 4096;  // + prefetch sink

template <int M, int MODE, bool F16>
__global__ __launch_bounds__(512) void fused_stream_kernel(StreamParams P) {
  constexpr int NP = 2;
  constexpr int RING = kSRing;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Yx = smem;
  char* Rx = Yx + 2 * NP * kSYxPart;
  float* Stat = reinterpret_cast<float*>(Rx + NP * kSRxPart);
  char* Slots = reinterpret_cast<char*>(Stat) + 1024;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = wave & 3;
  const bool mover = wave >= 4;
  const int r = lane & 31, h = lane >> 5;
  const int64_t wg = blockIdx.x;
  const int64_t patch = wg * kFP + r;
  const bool live = patch < P.b;
  const int nph = P.nph;
  // this wave's state tiles: float4 index of (phase q, group g)
  const int64_t st_wave = (wg * nph * 4 + w) * 256 + lane;   // + q*1024 + g*64

  // =========================================================== mover waves
  if (mover) {
    // barrier-for-barrier mirror of the compute path below
    if (F16) {
      __syncthreads();
      __syncthreads();
    }
    for (int q = 0; q < nph; ++q) {          // R_0 prologue
      __syncthreads();
      __syncthreads();
      const bool phase_nz = (Stat[0] + Stat[1] + Stat[2] + Stat[3]) != 0.f;
      if (phase_nz) __syncthreads();
      __syncthreads();
    }
    __syncthreads();                         // exchange_r
    // state buffers as buffer resources: lane * 16 in the vector offset, the
    // tile in the scalar offset (bytes; one workgroup's share is < 4 GiB)
    const size_t wg_bytes = (size_t)nph * 4 * 4 * 1024;
    int which_cur = 0;
    // L2 prefetch of the dictionary.  At 4096 slots the packed operands are
    // 8 MiB, twice an XCD's L2, and all CUs of an XCD ask for the same line at
    // about the same time: every request then waits on the fill from the
    // Infinity Cache (measured: the dictionary stream alone ran at half the
    // rate of the L2-resident 1024-atom case).  The mover waves touch, one to
    // three phases ahead, a 1/nshare share each of the lines all CUs of the
    // XCD will want (copies into a 1 KiB LDS sink, nobody reads it); the share
    // comes from blockIdx under the round-robin XCD placement the dispatcher
    // uses in practice -- a wrong guess costs speed, never correctness.
    // (Measured gain: 17 %.  What remains, from in-kernel stamps: with the
    // dictionary twice the size of L2 its fragments arrive at ~42 B/clk per CU
    // instead of the 61 B/clk of the L2-resident 1024-atom case, and the
    // HBM-latency state copies on the same CU slow them further -- 10 500
    // cycles per phase against 5 000 in fc_fused.hip; profiles/
    // r02_stream_stamps.txt.)
    __amdgpu_buffer_rsrc_t rs_pack[2];
#pragma unroll
    for (int part = 0; part < 2; ++part)
      rs_pack[part] = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<void*>(P.pack[part]), 0, (int)(2u * P.pack_half),
          0x00020000);
    const int per_xcd = (int)(gridDim.x / 8u);
    const int nshare = per_xcd >= 32 ? 32 : (per_xcd >= 16 ? 16 : 0);
    const int rank = (int)((blockIdx.x / 8u) % 32u);
    char* sink = Slots + kSSlots * kSSlotBytes + w * 1024;
    auto prefetch_dict = [&](int q) {
      if (nshare == 0) return;
      const int pT = (q + 1) % nph, pA = (q + 3) % nph;
      // 128 KiB per kind and phase (hi + lo), 4 KiB (nshare 32) or 8 KiB per CU
      const int pieces = nshare == 32 ? 1 : 2;
      for (int piece = 0; piece < pieces; ++piece) {
        const int slice = (rank % nshare) * pieces + piece;     // 0..31
        const int part = slice >> 4;
        const int off = (slice & 15) * 4096 + w * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rs_pack[part], (lds_void_t*)sink, 16, lane * 16,
            pA * 65536 + off, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rs_pack[part], (lds_void_t*)sink, 16, lane * 16,
            (int)P.pack_half + pT * 65536 + off, 0, 0);
      }
    };
    for (int it = 0; it < P.num_iters; ++it) {
      const char* base_cur = reinterpret_cast<const char*>(P.state[which_cur]) +
                             (size_t)wg * wg_bytes;
      const char* base_old =
          reinterpret_cast<const char*>(P.state[which_cur ^ 1]) +
          (size_t)wg * wg_bytes;
      const __amdgpu_buffer_rsrc_t rs_cur = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(base_cur), 0, (int)wg_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_old = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(base_old), 0, (int)wg_bytes, 0x00020000);
      auto dma_phase = [&](int q) {
        char* slot = Slots + (q & 1) * kSSlotBytes + w * 4096;
        const int soff = (q * 4 + w) * 4096;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // nt (aux = 2): read-once lines, kept out of the dictionary's way
          // in L2 (measured: 5 % per iteration)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              rs_cur, (lds_void_t*)(slot + g * 1024), 16, lane * 16,
              soff + g * 1024, 0, 2);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              rs_old, (lds_void_t*)(slot + 16384 + g * 1024), 16, lane * 16,
              soff + g * 1024, 0, 2);
        }
      };
      dma_phase(0);
      if (nph > 1) dma_phase(1);
      // phase 0 landed (all but the youngest 8 operations) before the compute
      // waves start its epilogue
      if (nph > 1)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                       // barrier E
      float4* out = P.state[which_cur ^ 1] + st_wave;
      for (int q = 0; q < nph; ++q) {
        // phase q+1 landed (and the stores of phase q-1 left) before the
        // barrier that lets its epilogue start; the 8 copies of phase q+2,
        // issued last, stay in flight
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                     // barrier B(q)
        // the compute wave left c' of phase q where c_{k-1} was: to memory,
        // then the slot is free for phase q+2
        const char* src = Slots + (q & 1) * kSSlotBytes + w * 4096 + 16384 +
                          lane * 16;
        // (named values, not an array: the array stayed in scratch memory
        // and every copy paid a second, dead store through the same path)
        const float4 c0 = *reinterpret_cast<const float4*>(src);
        const float4 c1 = *reinterpret_cast<const float4*>(src + 1024);
        const float4 c2 = *reinterpret_cast<const float4*>(src + 2048);
        const float4 c3 = *reinterpret_cast<const float4*>(src + 3072);
        float4* dst = out + (int64_t)q * 1024;
        dst[0] = c0;
        dst[64] = c1;
        dst[128] = c2;
        dst[192] = c3;
        if (q + 2 < nph) dma_phase(q + 2);
        prefetch_dict(q);
      }
      __syncthreads();                       // exchange_r
      which_cur ^= 1;
    }
    return;
  }

  // ========================================================= compute waves
  // dictionary fragments: one resource per part, [packA | packT]
  __amdgpu_buffer_rsrc_t rs[NP];
#pragma unroll
  for (int part = 0; part < NP; ++part)
    rs[part] = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(P.pack[part]), 0, (int)(2u * P.pack_half),
        0x00020000);
  const unsigned frag_voff = (unsigned)lane * 16u;
  const unsigned a_base = (unsigned)w * 16384u;                 // wave's tile
  const unsigned t_base = P.pack_half + (unsigned)(2 * w) * 8192u;
  // fragment i of segment (kind, q): kind 0 = A (step 1), 1 = T (step 3)
#define VTC_OFF_A(i) ((unsigned)(i) * 1024u)
#define VTC_OFF_T(i) ((unsigned)((((i) & 1) * 8 + ((i) >> 1)) * 1024))
  auto seg_base = [&](int kind, int q) -> unsigned {
    return (kind ? t_base : a_base) + (unsigned)q * 65536u;
  };

  // LDS lane bases (as in fc_fused.hip)
  const int yx_rd = r * kSYxRow + 16 * h;
  const int yx_wr = r * kSYxRow + 64 * w + 8 * h;
  const int rx_rd = r * kSRxRow + 16 * h;
  const int rx_wr = r * kSRxRow + 128 * w + 8 * h;
  const int slot_ln = w * 4096 + lane * 16;      // + slot, + 16384 (c_{k-1}), + g*1024

  f32x16v Racc[2], Gb[2];
  uint4 ring[NP][RING];
  const float* x_row =
      P.images + (live ? patch : 0) * kFN + 64 * w + 4 * h;   // + 32 nb + 8 g

  // ---- F16: per-patch power-of-two units (see fc_fused.hip) ---------------
  float sigma_y = 1.f, inv_sigma_y = 1.f, sigma_d = 1.f, inv_sigma_d = 1.f;
  if (F16) {
    float sx = 0.f, sy = 0.f;
    if (live) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 x =
              *reinterpret_cast<const float4*>(x_row + 32 * nb + 8 * g);
          sx += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        }
    }
    for (int q = 0; q < nph; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 c = P.state[0][st_wave + (int64_t)q * 1024 + g * 64];
        sy += c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
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
    // the stored iterates move to the scaled units as well
    for (int q = 0; q < nph; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t at = st_wave + (int64_t)q * 1024 + g * 64;
        float4 c = P.state[0][at];
        c.x *= sigma_y; c.y *= sigma_y; c.z *= sigma_y; c.w *= sigma_y;
        P.state[0][at] = c;
        P.state[1][at] = c;
      }
  }
  float x_scale = F16 ? sigma_d * sigma_y : 1.f;   // X in residual units
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
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) Racc[nb][e] = 0.f;

  // Workgroup barrier that orders LDS traffic only.  __syncthreads() also
  // waits for vmcnt(0), i.e. drains the dictionary ring at every phase: the
  // youngest fragment load's full latency, exposed once per barrier.
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // publish four values of y' for step 3
  auto publish4 = [&](const float (&v4)[4], int buf, int g) {
    uint2 hi, lo;
    split4<F16, NP>(v4, &hi, &lo);
    char* dst = Yx + buf * NP * kSYxPart + yx_wr + 16 * g;
    *reinterpret_cast<uint2*>(dst) = hi;
    *reinterpret_cast<uint2*>(dst + kSYxPart) = lo;
  };

  // step 3 of phase q from exchange buffer `buf`; fragments from the ring
  // (stream segment T(q), next segment (nk, nq)) or loaded on the spot
  auto step3 = [&](int q, int buf, bool pipe, int nk, int nq) {
    const unsigned cur = seg_base(1, q);
    const unsigned nxt = seg_base(nk, nq);
    uint4 yb_next[NP];
#pragma unroll
    for (int part = 0; part < NP; ++part)
      yb_next[part] = *reinterpret_cast<const uint4*>(
          Yx + (buf * NP + part) * kSYxPart + yx_rd);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      uint4 yb[NP];
#pragma unroll
      for (int part = 0; part < NP; ++part) {
        yb[part] = yb_next[part];
        if (ks + 1 < 8)
          yb_next[part] = *reinterpret_cast<const uint4*>(
              Yx + (buf * NP + part) * kSYxPart + yx_rd + 32 * (ks + 1));
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int i = 2 * ks + nb;
        uint4 a[NP];
#pragma unroll
        for (int part = 0; part < NP; ++part)
          a[part] = pipe ? ring[part][i % RING]
                         : buffer_load16(rs[part], frag_voff,
                                         cur + VTC_OFF_T(i));
        Racc[nb] = VTC_MFMA(a[0], yb[0], Racc[nb]);
        Racc[nb] = VTC_MFMA(a[0], yb[1], Racc[nb]);
        Racc[nb] = VTC_MFMA(a[1], yb[0], Racc[nb]);
        if (pipe) {
          const int j = i + RING;
          const unsigned off =
              (j < 16) ? cur + VTC_OFF_T(j < 16 ? j : 0)
                       : nxt + (nk ? VTC_OFF_T(j >= 16 ? j - 16 : 0)
                                   : VTC_OFF_A(j >= 16 ? j - 16 : 0));
#pragma unroll
          for (int part = 0; part < NP; ++part)
            ring[part][i % RING] = buffer_load16(rs[part], frag_voff, off);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // R_{k+1} = Racc - X -> parts -> LDS; F16: range guard (fc_fused.hip), the
  // rare rescale also walks this wave's tiles of both stored iterates
  auto exchange_r = [&]() {
    float v[2][16];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live)
          x = *reinterpret_cast<const float4*>(x_row + 32 * nb + 8 * g);
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xk = F16 ? xs[k] * x_scale : xs[k];
          float d = sub_rn(Racc[nb][4 * g + k], xk);
          if (F16) d *= r_scale;
          v[nb][4 * g + k] = d;
        }
      }
    if (F16) {
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
        for (int q = 0; q < nph; ++q)
          for (int g = 0; g < 4; ++g)
            for (int which = 0; which < 2; ++which) {
              const int64_t at = st_wave + (int64_t)q * 1024 + g * 64;
              float4 c = P.state[which][at];
              c.x *= f; c.y *= f; c.z *= f; c.w *= f;
              P.state[which][at] = c;
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
        *reinterpret_cast<uint2*>(dst + kSRxPart) = lo;
      }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) Racc[nb][e] = 0.f;
    // every store to the iterates (the c' tiles of this iteration, a rescale)
    // has completed before the mover waves fetch them again
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  // ---- R_0 = Y_0 D - X (Y_0 = c_0; all zero for a cold start) -------------
  {
    // a cold start has an all-zero state: skip the products
    for (int q = 0; q < nph; ++q) {
      float4 c[4];
      bool nz = false;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        c[g] = P.state[0][st_wave + (int64_t)q * 1024 + g * 64];
        nz = nz || c[g].x != 0.f || c[g].y != 0.f || c[g].z != 0.f ||
             c[g].w != 0.f;
      }
      // the decision must be the same in all waves: through LDS
      if (lane == 0) Stat[w] = 0.f;
      __syncthreads();
      if (__any(nz) && lane == 0) Stat[w] = 1.f;
      __syncthreads();
      const bool phase_nz = (Stat[0] + Stat[1] + Stat[2] + Stat[3]) != 0.f;
      if (phase_nz) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float v4[4] = {c[g].x, c[g].y, c[g].z, c[g].w};
          publish4(v4, q & 1, g);
        }
        __syncthreads();
        step3(q, q & 1, false, 0, 0);
      }
      __syncthreads();
    }
  }
  exchange_r();

  // prime the ring with the first RING fragments of segment A(0)
#pragma unroll
  for (int i = 0; i < RING; ++i)
#pragma unroll
    for (int part = 0; part < NP; ++part)
      ring[part][i] =
          buffer_load16(rs[part], frag_voff, seg_base(0, 0) + VTC_OFF_A(i));

  const bool fista = P.fista != 0;
  float4* st_old = P.state[1];    // c_{k-1}, overwritten by c_{k+1}
  float4* st_cur = P.state[0];

  float pvals[4];   // p = y - eta G of the group being processed

  // Epilogue of phase q (gradient tile Gb[set]), element e.  Elements come in
  // order 0..15; a group of four (one float4 of the state) is fetched from the
  // phase's LDS slot when its first element arrives and finished with its last.
  float4 cur4, old4;
  auto epilogue_elem = [&](int q, int set, int e, float beta_prev,
                           float beta) {
    const int g = e >> 2, k = e & 3;
    if (k == 0) {
      const char* slot = Slots + (q & 1) * kSSlotBytes + slot_ln + g * 1024;
      cur4 = *reinterpret_cast<const float4*>(slot);
      old4 = *reinterpret_cast<const float4*>(slot + 16384);
    }
    const float ck = (k == 0) ? cur4.x : (k == 1) ? cur4.y
                   : (k == 2) ? cur4.z : cur4.w;
    const float co = (k == 0) ? old4.x : (k == 1) ? old4.y
                   : (k == 2) ? old4.z : old4.w;
    // y_k exactly as the previous iteration formed it
    const float y = add_rn(ck, mul_rn(beta_prev, sub_rn(ck, co)));
    pvals[k] = sub_rn(y, mul_rn(eta, Gb[set][e]));
    if (k != 3) return;
    float cn[4];
    if (M == 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t) cn[t] = shrink_fast<MODE>(pvals[t], cutoff_l);
    } else {
      float sq[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) sq[t] = mul_rn(pvals[t], pvals[t]);
      if (M == 2) {
        sq[0] = sq[1] = add_rn(sq[0], sq[1]);
        sq[2] = sq[3] = add_rn(sq[2], sq[3]);
      } else {
        float total = add_rn(add_rn(sq[0], sq[1]), add_rn(sq[2], sq[3]));
        // slots 8g .. 8g+7 of the tile: this lane's four and the four of the
        // lane 32 further on
        if (M == 8) total = add_across_halves(total);
        sq[0] = sq[1] = sq[2] = sq[3] = total;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float norm = sqrtf(sq[t]);
        if (norm == 0.f) norm = 1.f;          // subspace_ista_fista.py:150
        const float scale = clamp_min0(sub_rn(1.f, cutoff_l / norm));
        cn[t] = mul_rn(pvals[t], scale);
      }
    }
    const float cks[4] = {cur4.x, cur4.y, cur4.z, cur4.w};
    float yn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
      yn[t] = add_rn(cn[t], mul_rn(beta, sub_rn(cn[t], cks[t])));
    // c' goes back into the slot, over the c_{k-1} it replaces; the mover wave
    // stores it to memory after the phase's barrier (a store issued here would
    // sit in this wave's in-order vmcnt queue ahead of the dictionary ring)
    *reinterpret_cast<float4*>(Slots + (q & 1) * kSSlotBytes + slot_ln +
                               16384 + g * 1024) =
        make_float4(cn[0], cn[1], cn[2], cn[3]);
    publish4(yn, q & 1, g);
  };

  // step 1 of phase q (segment A(q), next segment (nk, nq)) into Gb[set], with
  // the epilogue of phase q-1 (the other tile) interleaved when `overlap`
  auto step1 = [&](int q, int set, bool overlap, int nk, int nq,
                   float beta_prev, float beta) {
    const unsigned cur = seg_base(0, q);
    const unsigned nxt = seg_base(nk, nq);
    f32x16v& G = Gb[set];
#pragma unroll
    for (int e = 0; e < 16; ++e) G[e] = 0.f;
    uint4 rb_next[NP];
#pragma unroll
    for (int part = 0; part < NP; ++part)
      rb_next[part] =
          *reinterpret_cast<const uint4*>(Rx + part * kSRxPart + rx_rd);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      uint4 rb[NP];
#pragma unroll
      for (int part = 0; part < NP; ++part) {
        rb[part] = rb_next[part];
        if (i + 1 < 16)
          rb_next[part] = *reinterpret_cast<const uint4*>(
              Rx + part * kSRxPart + rx_rd + 32 * (i + 1));
      }
      G = VTC_MFMA(ring[0][i % RING], rb[0], G);
      G = VTC_MFMA(ring[0][i % RING], rb[1], G);
      G = VTC_MFMA(ring[1][i % RING], rb[0], G);
      if (overlap) {
        epilogue_elem(q - 1, set ^ 1, i, beta_prev, beta);
        __builtin_amdgcn_sched_barrier(0);
      }
      const int j = i + RING;
      const unsigned off =
          (j < 16) ? cur + VTC_OFF_A(j < 16 ? j : 0)
                   : nxt + (nk ? VTC_OFF_T(j >= 16 ? j - 16 : 0)
                               : VTC_OFF_A(j >= 16 ? j - 16 : 0));
#pragma unroll
      for (int part = 0; part < NP; ++part)
        ring[part][i % RING] = buffer_load16(rs[part], frag_voff, off);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const bool stamping = (P.debug & 32) != 0;
  unsigned long long acc_t[4] = {0, 0, 0, 0}, t_prev = 0;
  auto stamp = [&](int slot) {
    if (!stamping) return;
    unsigned long long now;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
    acc_t[slot] += now - t_prev;
    t_prev = now;
  };
  if (stamping)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
  for (int it = 0; it < P.num_iters; ++it) {
    const float beta = fista ? P.betas[it] : 0.f;
    const float beta_prev = (fista && it > 0) ? P.betas[it - 1] : 0.f;
    step1(0, 0, false, 0, 1, beta_prev, beta);          // A(0), then A(1)
    lds_barrier();                                      // barrier E: slot 0 in
    stamp(3);
    for (int q = 0; q < nph; q += 2) {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int qq = q + par;        // this phase: gradient tile `par`
        if (qq + 1 < nph) {
          // A(qq+1) is followed by T(qq)
          step1(qq + 1, par ^ 1, true, 1, qq, beta_prev, beta);
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            epilogue_elem(qq, par, e, beta_prev, beta);
        }
        stamp(0);
        lds_barrier();                 // barrier B(qq)
        stamp(1);
        // T(qq) is followed by A(qq+2), by T(nph-1) after T(nph-2), and by the
        // next iteration's A(0) after T(nph-1)
        const int nk = (qq + 2 <= nph - 1) ? 0 : (qq == nph - 2 ? 1 : 0);
        const int nq = (qq + 2 <= nph - 1) ? qq + 2
                                           : (qq == nph - 2 ? nph - 1 : 0);
        step3(qq, qq & 1, true, nk, nq);
        stamp(2);
      }
    }
    exchange_r();
    stamp(3);
    float4* swap = st_cur;
    st_cur = st_old;
    st_old = swap;
  }
  // the newest iterate is in st_cur (the host knows which buffer from the
  // parity of num_iters); tell the unpack kernel how to scale it back
  if (live && w == 0 && h == 0) P.patch_scale[patch] = inv_sigma_y;
  if (stamping && lane == 0) {
    for (int k = 0; k < 4; ++k) atomicAdd(P.stamps + k, acc_t[k]);
    atomicAdd(P.stamps + 7, 1ull);
  }
#undef VTC_OFF_A
#undef VTC_OFF_T
}

// -------------------------------------------------------------------- host
bool stream_shape_supported(int64_t b, int64_t n, int64_t slots, int64_t m,
                            int precision) {
  if (n != kFN || b <= 0 || slots <= 0) return false;
  if (slots % 256 != 0 || slots > 16384) return false;
  if (!(m == 1 || m == 2 || m == 4 || m == 8)) return false;
  return precision == VTC_BF16X3 || precision == VTC_F16X3;
}

static size_t stream_state_bytes(int64_t b, int64_t slots) {
  return (size_t)ceil_div(b, 32) * 32 * slots * sizeof(float);
}

size_t stream_workspace_bytes(int64_t b, int64_t n, int64_t slots,
                              int precision) {
  if (!stream_shape_supported(b, n, slots, 1, precision)) return 256;
  return 2 * align_up(stream_state_bytes(b, slots), 256) +
         2 * align_up((size_t)2 * slots * kFN * 2, 256) +
         align_up((size_t)b * sizeof(float), 256) + 256;
}

template <int M, int MODE, bool F16>
static int launch_stream(const StreamParams& P, hipStream_t st) {
  auto kernel = fused_stream_kernel<M, MODE, F16>;
  static unsigned long long configured = 0;
  if (first_use_on_this_device(&configured)) {
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, kSLdsTotal));
  }
  hipLaunchKernelGGL(kernel, dim3((unsigned)ceil_div(P.b, kFP)), dim3(512),
                     kSLdsTotal, st, P);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

template <bool F16>
static int dispatch_stream(const StreamParams& P, int m, int threshold,
                           hipStream_t st) {
  switch (m) {
    case 2: return launch_stream<2, VTC_SOFT, F16>(P, st);
    case 4: return launch_stream<4, VTC_SOFT, F16>(P, st);
    case 8: return launch_stream<8, VTC_SOFT, F16>(P, st);
    default: break;
  }
  switch (threshold) {
    case VTC_SOFT: return launch_stream<1, VTC_SOFT, F16>(P, st);
    case VTC_SOFT_NONNEG: return launch_stream<1, VTC_SOFT_NONNEG, F16>(P, st);
    case VTC_HARD: return launch_stream<1, VTC_HARD, F16>(P, st);
    default: return launch_stream<1, VTC_HARD_NONNEG, F16>(P, st);
  }
}

int fused_max_iters_for_stream() { return fused_max_iters(); }

int run_stream(const float* images, const float* dictionary,
               const float* initial, float* codes, int64_t b, int64_t n,
               int64_t slots, int64_t m, float eta, const float* eta_dev,
               float sparsity_weight, int num_iters, int variant,
               int threshold, int precision, void* workspace,
               size_t workspace_bytes, int* iters_run, hipStream_t st) {
  if (!stream_shape_supported(b, n, slots, m, precision)) {
    set_error("streamed fused FISTA: unsupported shape");
    return VTC_ERR_UNSUPPORTED;
  }
  if (num_iters > fused_max_iters()) {
    set_error("streamed fused FISTA: at most %d iterations per call",
              fused_max_iters());
    return VTC_ERR_UNSUPPORTED;
  }
  if (!workspace ||
      workspace_bytes < stream_workspace_bytes(b, n, slots, precision)) {
    set_error("streamed fused FISTA: workspace too small");
    return VTC_ERR_WORKSPACE;
  }
  const float* betas_dev = fista_beta_table_on_this_device();
  if (!betas_dev) {
    set_error("streamed fused FISTA: could not place the momentum table");
    return VTC_ERR_HIP;
  }
  const bool f16 = (precision == VTC_F16X3);
  const size_t pack_half = (size_t)slots * kFN * 2;
  Carver ws(workspace);
  StreamParams P;
  P.state[0] = reinterpret_cast<float4*>(
      ws.take<char>(stream_state_bytes(b, slots)));
  P.state[1] = reinterpret_cast<float4*>(
      ws.take<char>(stream_state_bytes(b, slots)));
  unsigned short* packs[2];
  for (int part = 0; part < 2; ++part)
    packs[part] = ws.take<unsigned short>(2 * (size_t)slots * kFN);
  float* patch_scale = ws.take<float>((size_t)b);
  float* dscale = ws.take<float>(2);
  if (f16) {
    hipLaunchKernelGGL(dictionary_scale_kernel, dim3(1), dim3(1024), 0, st,
                       dictionary, (int64_t)slots * kFN, dscale);
    VTC_LAUNCH_CHECK();
  }
  {
    unsigned short* hiT = packs[0] + (size_t)slots * kFN;
    unsigned short* loT = packs[1] + (size_t)slots * kFN;
    if (f16)
      hipLaunchKernelGGL(pack_dictionary_kernel<true>, dim3(1024), dim3(256),
                         0, st, dictionary, (int)slots, packs[0], hiT,
                         packs[1], loT, dscale);
    else
      hipLaunchKernelGGL(pack_dictionary_kernel<false>, dim3(1024), dim3(256),
                         0, st, dictionary, (int)slots, packs[0], hiT,
                         packs[1], loT, dscale);
    VTC_LAUNCH_CHECK();
  }
  const int nph = (int)(slots / kPhaseAtoms);
  hipLaunchKernelGGL(stream_state_pack_kernel, dim3(2048), dim3(256), 0, st,
                     initial, P.state[0], P.state[1], b, nph);
  VTC_LAUNCH_CHECK();
  P.images = images;
  P.pack[0] = packs[0];
  P.pack[1] = packs[1];
  P.pack_half = (unsigned)pack_half;
  P.betas = betas_dev;
  P.patch_scale = patch_scale;
  P.b = b;
  P.nph = nph;
  P.num_iters = num_iters;
  P.fista = (variant == VTC_FISTA) ? 1 : 0;
  P.eta = eta;
  P.eta_dev = eta_dev;
  P.lam = sparsity_weight;
  P.cutoff = sparsity_weight * eta;
  P.dscale = dscale;
  {
    static const int debug_flags = [] {
      const char* dbg = getenv("VTC_STREAM_DEBUG");
      return dbg ? atoi(dbg) : 0;
    }();
    P.debug = debug_flags;
  }
  unsigned long long* stamps_dev = nullptr;
  P.stamps = nullptr;
  if (P.debug & 32) {
    VTC_HIP_CHECK(hipMalloc(&stamps_dev, 64));
    VTC_HIP_CHECK(hipMemsetAsync(stamps_dev, 0, 64, st));
    P.stamps = stamps_dev;
  }
  int rc = f16 ? dispatch_stream<true>(P, (int)m, threshold, st)
               : dispatch_stream<false>(P, (int)m, threshold, st);
  if (rc != VTC_OK) return rc;
  if (stamps_dev) {
    unsigned long long host[8];
    VTC_HIP_CHECK(hipMemcpyAsync(host, stamps_dev, 64, hipMemcpyDeviceToHost, st));
    VTC_HIP_CHECK(hipStreamSynchronize(st));
    VTC_HIP_CHECK(hipFree(stamps_dev));
    const double per = (double)host[7] * num_iters * nph;
    const char* names[4] = {"step1+epi", "barrier", "step3", "iter-edge"};
    for (int k = 0; k < 4; ++k)
      fprintf(stderr, "[vtc stream stamps] %-9s %8.0f cycles/phase/wave\n",
              names[k], host[k] / per);
  }
  // iteration k writes c_{k+1} over the older iterate: after T iterations the
  // newest one is in state[T & 1]
  hipLaunchKernelGGL(stream_state_unpack_kernel, dim3(2048), dim3(256), 0, st,
                     P.state[num_iters & 1], patch_scale, codes, b, nph);
  VTC_LAUNCH_CHECK();
  if (iters_run) *iters_run = num_iters;
  return VTC_OK;
}

}  // namespace vtc
