// Proximal epilogue of the gradient product on the bf16x3 tiles, shared by the
// subspace plugin (group shrinkage) and the fully-connected plugin outside the
// fused kernel (element-wise thresholds).
#pragma once

#include "gemm_x3.h"

namespace vtc {

// Gradient step and group proximal step in the epilogue of the R Dg^T product
// (bf16x3 path, group size a power of two <= 32): the m slots of a group are m
// adjacent columns, i.e. m adjacent lanes of the accumulator tile, so the
// group norm is a lane-shuffle tree and the (b, slots) state is read and
// written once per iteration instead of three times.
// f16 split of the next product (x3_scale.h): TRACK adds the slot that
// receives the maximum of the new iterate |Y'| (a separate instantiation: the
// functors of the other paths stay as they were)
template <bool TRACK>
struct EpiMaxTrack {
  __device__ __forceinline__ void note(float) {}
  __device__ __forceinline__ void publish() const {}
};
template <>
struct EpiMaxTrack<true> {
  unsigned* max_out = nullptr;
  float mx = 0.f;
  __device__ __forceinline__ void note(float v) { mx = fmaxf(mx, fabsf(v)); }
  __device__ __forceinline__ void publish() const {
    if (max_out) cx_publish_max_wave(mx, max_out);
  }
};

template <int M, bool ELEMENTWISE = false, bool TRACK = false>
struct EpiGroupProx {
  static constexpr bool kWholeTile = true;
  static constexpr int kPrefetch = 32;     // 16 of Y + 16 of the codes
  float* Y;
  float* C;
  int64_t ld;
  float eta, cutoff, beta;
  int fista;
  double* delta_sum;
  double local;
  int mode;   // ELEMENTWISE: vtc_threshold of the fully-connected plugin
  // sync-free callers: eta lives in device memory (vtc_lambda_max out[1]) and
  // the threshold is lam * eta, one f32 multiply as on the host
  const float* eta_dev = nullptr;
  float lam = 0.f;
  // Where the new Y and the new codes go: other buffers than the ones read
  // (the caller swaps them per iteration).  The state is in the caller's
  // row-major layout; when a row is not a multiple of 128 bytes, tiles of
  // blocks on different XCDs share cache lines, and a line that two XCDs both
  // read and partly rewrite within one launch can lose one of the updates
  // (their L2s are not coherent within a launch).
  float* Yo = nullptr;
  float* Co = nullptr;
  EpiMaxTrack<TRACK> track;
  __device__ __forceinline__ void resolve() {
    if (eta_dev) {
      eta = *eta_dev;
      cutoff = mul_rn(lam, eta);
    }
  }
  // rows of the block as buffer resources: a row past the batch is past the
  // end of the resource (reads give 0, writes are dropped)
  struct Ctx {
    __amdgpu_buffer_rsrc_t yrs, crs, yws, cws;
  };
  __device__ __forceinline__ Ctx begin(int64_t m0, int64_t rows) const {
    const int64_t left = rows - m0 < kX3BM ? rows - m0 : kX3BM;
    const int bytes = (int)(left * ld * 4);
    Ctx ctx;
    ctx.yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(Y + m0 * ld), 0, bytes,
                                                0x00020000);
    ctx.crs = __builtin_amdgcn_make_buffer_rsrc((void*)(C + m0 * ld), 0, bytes,
                                                0x00020000);
    ctx.yws = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((Yo ? Yo : Y) + m0 * ld), 0, bytes, 0x00020000);
    ctx.cws = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((Co ? Co : C) + m0 * ld), 0, bytes, 0x00020000);
    return ctx;
  }
  // Global accesses are 16 bytes per lane: lane l owns row (l >> 3) + 8 q
  // (q = 0..3) and columns 4 (l & 7) .. +3 of the 32 x 32 tile, so one wave
  // instruction moves 8 full 128-byte row segments.  (The vector-memory
  // address unit takes 16 cycles per wave instruction whatever the width: a
  // dword per lane is a quarter of the 64 B/clk the CU can move.)  The
  // accumulator tile (lane = column, register = row) is brought into that
  // shape through the wave's slice of the block's LDS, free after the K loop.
  static constexpr int kPitch = 36;        // floats per staged row
  __device__ __forceinline__ unsigned lane_offset(int row0, int64_t col0,
                                                  int lane,
                                                  int64_t cols) const {
    const int64_t col = col0 + 4 * (lane & 7);
    return col < cols ? (unsigned)(((int64_t)(row0 + (lane >> 3)) * ld + col) *
                                   4)
                      : 0x80000000u;
  }
  __device__ __forceinline__ void load(const Ctx& ctx, int row0, int64_t col0,
                                       int lane, int64_t cols,
                                       float (&buf)[32]) const {
    const unsigned off = lane_offset(row0, col0, lane, cols);
    const unsigned ld4 = (unsigned)(ld * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const x3_u32x4 y4 = __builtin_amdgcn_raw_buffer_load_b128(
          ctx.yrs, off, (unsigned)(8 * q) * ld4, 0);
      const x3_u32x4 c4 = __builtin_amdgcn_raw_buffer_load_b128(
          ctx.crs, off, (unsigned)(8 * q) * ld4, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        buf[4 * q + i] = __uint_as_float(y4[i]);
        buf[16 + 4 * q + i] = __uint_as_float(c4[i]);
      }
    }
  }
  __device__ __forceinline__ void finish(const Ctx& ctx, int row0,
                                         int64_t col0, int lane, int64_t cols,
                                         const f32x16& acc,
                                         const float (&buf)[32],
                                         float* scratch) {
    const unsigned off = lane_offset(row0, col0, lane, cols);
    const unsigned ld4 = (unsigned)(ld * 4);
    // accumulator (column l & 31, rows (r&3) + 8 (r>>2) + 4 (l>>5)) -> LDS
#pragma unroll
    for (int r = 0; r < 16; ++r)
      scratch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * kPitch +
              (lane & 31)] = acc[r];
    // the reads below take what OTHER lanes just wrote: keep the compiler
    // from moving them above the writes, and the next tile's writes above
    // them (the LDS itself serves a wave's accesses in order)
    asm volatile("" ::: "memory");
    float g[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(
          scratch + ((lane >> 3) + 8 * q) * kPitch + 4 * (lane & 7));
      g[4 * q + 0] = t.x; g[4 * q + 1] = t.y;
      g[4 * q + 2] = t.z; g[4 * q + 3] = t.w;
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // rows / columns outside the problem read as zero: p = 0, written
      // nowhere
      float p[4], sq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        p[i] = sub_rn(buf[4 * q + i], mul_rn(eta, g[4 * q + i]));
        sq[i] = mul_rn(p[i], p[i]);
      }
      // group norm: M consecutive columns = M/4 lanes x 4, or parts of a lane
      if (ELEMENTWISE) {
        // fully-connected plugin: one of the four thresholds per element
      } else if (M == 2) {
        sq[0] = sq[1] = add_rn(sq[0], sq[1]);
        sq[2] = sq[3] = add_rn(sq[2], sq[3]);
      } else if (M >= 4) {
        float total = add_rn(add_rn(sq[0], sq[1]), add_rn(sq[2], sq[3]));
#pragma unroll
        for (int o = 1; o < M / 4; o <<= 1)
          total = add_rn(total, __shfl_xor(total, o, 64));
        sq[0] = sq[1] = sq[2] = sq[3] = total;
      }
      x3_u32x4 y4, c4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float cn;
        if (ELEMENTWISE) {
          cn = shrink(p[i], cutoff, mode);     // ista_fista.py:107-120
        } else {
          float norm = sqrtf(sq[i]);
          if (norm == 0.f) norm = 1.f;  // subspace_ista_fista.py:150
          const float scale = clamp_min0(sub_rn(1.f, cutoff / norm));
          cn = mul_rn(p[i], scale);
        }
        const float d = sub_rn(cn, buf[16 + 4 * q + i]);
        const float yn = fista ? add_rn(cn, mul_rn(beta, d)) : cn;
        y4[i] = __float_as_uint(yn);
        c4[i] = __float_as_uint(cn);
        track.note(yn);
        if (delta_sum) local += (double)(fabsf(d) / eta);
      }
      // The row offset goes into the VECTOR offset, the scalar offset stays
      // the constant 0.  With a register in the scalar-offset field hipcc
      // (ROCm 7.2) treats a 16-byte buffer store as free of the "VALU
      // overwrites store data" hazard and puts no wait state behind it; on
      // gfx950 the hazard is there: `buffer_store_dwordx4 v[26:29], ..., s77
      // offen` followed at once by `v_fma_f32 v26, ...` stored the NEW v26 in
      // a few lanes, when the memory pipeline was busy (found by the bitwise
      // soak: ~0.03 % of the codes, first element of a 16-byte piece, run to
      // run different).  An out-of-range lane offset (0x80000000) stays out
      // of range with the row offset added.
      const unsigned row_off = off + (unsigned)(8 * q) * ld4;
      __builtin_amdgcn_raw_buffer_store_b128(y4, ctx.yws, row_off, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(c4, ctx.cws, row_off, 0, 0);
    }
  }
  __device__ __forceinline__ void operator()(int64_t, int64_t, float,
                                             int) const {}
  __device__ __forceinline__ void block_end() const {
    if (delta_sum) {
      const double w = wave_sum(local);
      if ((threadIdx.x & 63) == 0) atomicAdd(delta_sum, w);
    }
    track.publish();
  }
};

}  // namespace vtc
