// Dictionary gradient / apply / normalise for the fully-connected plugins, the
// Gram product of the Lipschitz step, and the Hessian-diagonal EMA.
//
// Reference lines restated here:
//   dict_update_rules/fully_connected/sc_steepest_descent.py:37-41
//   dict_update_rules/fully_connected/sc_cheap_quadratic_descent.py:42-48
//   dict_update_rules/fully_connected/subspace_sc_cheap_quadratic_descent.py:59-127
//   training/sparse_coding.py:154,160-161
//
// The update is split in two calls so that a data-parallel caller can
// all-reduce the un-normalised gradient sum between them:
//   gradient : G = C^T (C D - X)             two exact-f32 MFMA contractions,
//                                            the second split over the batch
//                                            (K = b) into slabs summed in a
//                                            fixed order -> reproducible
//   apply    : D -= eta_d * (G / b) [/ (h + 1e-3)] ; D /= ||row||
#include "common.h"
#include "gemm_f32.h"

namespace vtc {

// ---------------------------------------------------------------- slab sum
// Fixed summation order: slab z goes to partial sum z mod 8, the eight partial
// sums are combined pairwise.  (Eight independent chains keep eight loads in
// flight per thread: with up to 1024 slabs of a small gradient a single chain
// is latency bound -- 290 us for the convolutional gradient of configs[4],
// more than the contraction that produced the slabs.)
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, int slices,
                                   int64_t count, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += stride) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 8 <= slices; z += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        a[u] = add_rn(a[u], slabs[(int64_t)(z + u) * count + i]);
    }
    for (int u = 0; z < slices; ++z, ++u)
      a[u] = add_rn(a[u], slabs[(int64_t)z * count + i]);
    out[i] = add_rn(add_rn(add_rn(a[0], a[1]), add_rn(a[2], a[3])),
                    add_rn(add_rn(a[4], a[5]), add_rn(a[6], a[7])));
  }
}

__global__ void slab_reduce_minus_kernel(const float* __restrict__ slabs,
                                         int slices, int64_t count,
                                         const float* __restrict__ X,
                                         float* __restrict__ out,
                                         unsigned* max_out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float mx = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += stride) {
    float acc = slabs[i];
    for (int z = 1; z < slices; ++z) acc += slabs[(int64_t)z * count + i];
    const float r = sub_rn(acc, X[i]);
    out[i] = r;
    mx = fmaxf(mx, fabsf(r));
  }
  if (max_out) cx_publish_max_wave(mx, max_out);   // f16 split: x3_scale.h
}

int launch_slab_reduce_minus(const float* slabs, int slices, int64_t count,
                             const float* X, float* out, hipStream_t st,
                             unsigned* max_out) {
  if (count <= 0) return VTC_OK;
  int64_t blocks = ceil_div(count, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(slab_reduce_minus_kernel, dim3((unsigned)blocks),
                     dim3(256), 0, st, slabs, slices, count, X, out, max_out);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

int launch_slab_reduce(const float* slabs, int slices, int64_t count,
                       float* out, hipStream_t st) {
  if (count <= 0) return VTC_OK;
  int64_t blocks = ceil_div(count, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     st, slabs, slices, count, out);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

// Batch slices for the C^T E contraction: enough blocks to fill 256 CUs,
// at least 64 batch rows per slice.
static int gradient_slices(int64_t b, int64_t n, int64_t s) {
  const int64_t tiles = ceil_div(s, kGemmBM) * ceil_div(n, kGemmBN);
  int64_t want = ceil_div(1024, tiles);
  const int64_t cap = ceil_div(b, 64);
  if (want > cap) want = cap;
  if (want > 128) want = 128;
  if (want < 1) want = 1;
  return gemm_effective_slices(b, (int)want);
}

// ------------------------------------------------------------------ apply
__device__ __forceinline__ void fc_apply_row(
    float* __restrict__ D, const float* __restrict__ G,
    const float* __restrict__ hess, const float* __restrict__ P,
    float alignment_penalty, float batch_f, float stepsize, float lowest,
    int normalize, int64_t row, int64_t n, int lane) {
  float* d = D + row * n;
  const float* g = G + row * n;
  const float* p = P ? P + row * n : nullptr;
  const float denom = hess ? add_rn(hess[row], lowest) : 1.f;
  float sumsq = 0.f;
  for (int64_t j = lane; j < n; j += 64) {
    float grad = g[j] / batch_f;
    if (p) grad = add_rn(grad, mul_rn(alignment_penalty, p[j]));
    float step = mul_rn(stepsize, grad);
    if (hess) step = step / denom;
    const float v = sub_rn(d[j], step);
    d[j] = v;
    sumsq = fmaf(v, v, sumsq);
  }
  if (!normalize) return;
  const float norm = sqrtf(wave_sum(sumsq));
  for (int64_t j = lane; j < n; j += 64) d[j] = d[j] / norm;
}

// One wave per atom (dictionary row); a block owns `rows_per_block`
// consecutive rows.  The update is in place, and two blocks on different XCDs
// must not read-modify-write parts of one 128-byte line (the L2s are not
// coherent within a launch): the host picks rows_per_block so that a block's
// footprint is a whole number of lines (4 rows for 16x16 patches, 32 for an
// odd pixel count).
__global__ __launch_bounds__(256) void fc_apply_kernel(
    float* __restrict__ D, const float* __restrict__ G,
    const float* __restrict__ hess, const float* __restrict__ P,
    float alignment_penalty, float batch_f, float stepsize, float lowest,
    int normalize, int64_t s, int64_t n, int rows_per_block) {
  const int lane = threadIdx.x & 63;
  const int64_t first = (int64_t)blockIdx.x * rows_per_block;
  const int64_t last = first + rows_per_block < s ? first + rows_per_block : s;
  for (int64_t row = first + (threadIdx.x >> 6); row < last; row += 4)
    fc_apply_row(D, G, hess, P, alignment_penalty, batch_f, stepsize, lowest,
                 normalize, row, n, lane);
}
// ------------------------------------------------- alignment penalty (a7)
// One block per group.  Rows of the group are copied to LDS, the m x m table
// of cosines is formed, then every thread owns columns of the gradient rows.
// Per-slot gradients go to `slot_grad` (slots, n); penalty_accumulate_kernel
// sums the slots of each atom in increasing slot order.
constexpr int kMaxGroup = 32;

__global__ __launch_bounds__(256) void alignment_slot_kernel(
    const float* __restrict__ D, const int32_t* __restrict__ index,
    const uint8_t* __restrict__ valid, float* __restrict__ slot_grad,
    int64_t n, int m, int normalized) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* rows = lds;                        // m * n
  float* cosines = lds + (size_t)m * n;     // m * m
  float* norms = cosines + m * m;           // m
  const int g = blockIdx.x;
  const int32_t* idx = index + (int64_t)g * m;
  const uint8_t* ok = valid + (int64_t)g * m;
  int members = 0;
  for (int j = 0; j < m; ++j) members += ok[j] ? 1 : 0;  // valid slots lead
  for (int64_t e = threadIdx.x; e < (int64_t)members * n; e += blockDim.x) {
    const int r = (int)(e / n);
    const int64_t c = e % n;
    rows[e] = D[(int64_t)idx[r] * n + c];
  }
  __syncthreads();
  // dot products, one wave per (i, j) pair in turn
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int pair = wave; pair < members * members; pair += 4) {
    const int i = pair / members, j = pair % members;
    float acc = 0.f;
    for (int64_t c = lane; c < n; c += 64)
      acc = fmaf(rows[(int64_t)i * n + c], rows[(int64_t)j * n + c], acc);
    acc = wave_sum(acc);
    if (lane == 0) cosines[i * m + j] = acc;
  }
  __syncthreads();
  if (!normalized) {
    if (threadIdx.x < members)
      norms[threadIdx.x] = sqrtf(cosines[threadIdx.x * m + threadIdx.x]);
    __syncthreads();
  }
  for (int64_t e = threadIdx.x; e < (int64_t)members * n; e += blockDim.x) {
    const int i = (int)(e / n);
    const int64_t c = e % n;
    const float di = rows[(int64_t)i * n + c];
    float acc = 0.f;
    for (int j = 0; j < members; ++j) {
      const float dj = rows[(int64_t)j * n + c];
      float cs, toward_other, toward_self;
      if (normalized) {
        cs = cosines[i * m + j];
        toward_other = dj;
        toward_self = mul_rn(cs, di);
      } else {
        const float outer = mul_rn(norms[i], norms[j]);
        cs = cosines[i * m + j] / outer;
        toward_other = dj / outer;
        toward_self = mul_rn(cs / mul_rn(norms[i], norms[i]), di);
      }
      acc = add_rn(acc, mul_rn(sign_of(cs), sub_rn(toward_other, toward_self)));
    }
    slot_grad[((int64_t)g * m + i) * n + c] = acc;
  }
}

// out[a, :] = sum over the slots of atom a (CSR: atom_ptr, atom_slots)
__global__ void rows_by_atom_sum_kernel(const float* __restrict__ slot_rows,
                                        const int32_t* __restrict__ atom_ptr,
                                        const int32_t* __restrict__ atom_slots,
                                        float* __restrict__ out, int64_t s,
                                        int64_t n) {
  const int64_t a = blockIdx.x;
  if (a >= s) return;
  const int beg = atom_ptr[a], end = atom_ptr[a + 1];
  for (int64_t c = threadIdx.x; c < n; c += blockDim.x) {
    float acc = 0.f;
    for (int t = beg; t < end; ++t)
      acc = add_rn(acc, slot_rows[(int64_t)atom_slots[t] * n + c]);
    out[a * n + c] = acc;
  }
}

// ----------------------------------------------------------- code energy
// positions == 1: codes (b, s); partial[chunk][j] = sum over the chunk's rows.
__global__ __launch_bounds__(256) void energy_rows_kernel(
    const float* __restrict__ codes, int64_t b, int64_t s, int rows_per_chunk,
    float* __restrict__ partial) {
  const int64_t chunk = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= s) return;
  const int64_t r0 = chunk * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > b) r1 = b;
  float acc = 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    const float v = codes[r * s + j];
    acc = fmaf(v, v, acc);
  }
  partial[chunk * s + j] = acc;
}

// positions > 1: codes (b, s, positions); partial[image][j].
__global__ __launch_bounds__(256) void energy_maps_kernel(
    const float* __restrict__ codes, int64_t s, int64_t positions,
    float* __restrict__ partial) {
  __shared__ float red[4];
  const int64_t map = blockIdx.x;  // = image * s + j
  const float* p = codes + map * positions;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < positions; i += blockDim.x) {
    const float v = p[i];
    acc = fmaf(v, v, acc);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[map] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void hessian_ema_kernel(float* __restrict__ h,
                                   const float* __restrict__ energy,
                                   float batch_f, int64_t s) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= s) return;
  // h.mul_(0.99).add_(mean / 100)
  const float mean = energy[j] / batch_f;
  h[j] = add_rn(mul_rn(h[j], 0.99f), mean / 100.f);
}

static int energy_chunk_rows(int64_t b) {
  // about 512 chunks, at least 32 rows each
  int64_t rows = ceil_div(b, 512);
  if (rows < 32) rows = 32;
  return (int)rows;
}

}  // namespace vtc

using namespace vtc;

// --------------------------------------------------------------------------
extern "C" int vtc_gram(const float* a, int64_t rows, int64_t cols,
                        int transpose_a, float* gram, void* stream) {
  VTC_REQUIRE(a && gram, "vtc_gram: null pointer");
  VTC_REQUIRE(rows > 0 && cols > 0, "vtc_gram: bad sizes");
  hipStream_t st = as_stream(stream);
  if (transpose_a) {
    // (cols, cols) = A^T A : both operands indexed [k][line]
    EpiStore e{gram, cols};
    return launch_gemm_f32<false, false>(a, cols, a, cols, cols, cols, rows, 1,
                                         e, st);
  }
  EpiStore e{gram, rows};
  return launch_gemm_f32<true, true>(a, cols, a, cols, rows, rows, cols, 1, e,
                                     st);
}

extern "C" size_t vtc_fc_dict_gradient_workspace_bytes(int64_t b, int64_t n,
                                                       int64_t s) {
  if (b <= 0 || n <= 0 || s <= 0) return 256;
  size_t bytes = align_up((size_t)b * n * sizeof(float), 256);  // residual
  bytes += align_up((size_t)gradient_slices(b, n, s) * s * n * sizeof(float),
                    256);
  return bytes;
}

extern "C" int vtc_fc_dict_gradient(const float* images,
                                    const float* dictionary,
                                    const float* codes, float* grad_sum,
                                    int64_t b, int64_t n, int64_t s,
                                    void* workspace, size_t workspace_bytes,
                                    void* stream) {
  VTC_REQUIRE(images && dictionary && codes && grad_sum,
              "vtc_fc_dict_gradient: null pointer");
  VTC_REQUIRE(b > 0 && n > 0 && s > 0, "vtc_fc_dict_gradient: bad sizes");
  if (!workspace ||
      workspace_bytes < vtc_fc_dict_gradient_workspace_bytes(b, n, s)) {
    set_error("vtc_fc_dict_gradient: workspace too small");
    return VTC_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  Carver ws(workspace);
  float* E = ws.take<float>((size_t)b * n);
  const int slices = gradient_slices(b, n, s);
  float* slabs = ws.take<float>((size_t)slices * s * n);
  // E = C D - X
  EpiMinus e1{E, images, n, n};
  int rc = launch_gemm_f32<true, false>(codes, s, dictionary, n, b, n, s, 1,
                                        e1, st);
  if (rc != VTC_OK) return rc;
  // G = C^T E.  Small batches of a small dictionary (the reference's example
  // sizes: 250 patches, 256 atoms): one launch of the 32x32-tile kernel
  // straight into grad_sum -- no slabs, no reduction pass.
  if (b <= 512 && gemm_prefers_small(s, n)) {
    EpiStore direct{grad_sum, n};
    return launch_gemm_f32<false, false>(codes, s, E, n, s, n, b, 1, direct,
                                         st);
  }
  // otherwise K = b split into slabs
  EpiSlab e2{slabs, s * n, n};
  rc = launch_gemm_f32<false, false>(codes, s, E, n, s, n, b, slices, e2, st);
  if (rc != VTC_OK) return rc;
  return launch_slab_reduce(slabs, slices, s * n, grad_sum, st);
}

// ---------------------------------------------- ICA natural gradient (f4)
// dict_update_rules/fully_connected/ica_natural_gradient.py:26-35:
//   D += stepsize * ((C^T sign(C)) / b - I) D
// split like the sparse-coding update: the (s, s) moment C^T sign(C) is the
// quantity a data-parallel caller sums over ranks.
__global__ void sign_kernel(const float* __restrict__ x, float* __restrict__ y,
                            int64_t count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += stride)
    y[i] = sign_of(x[i]);
}

// T = M / b - I
__global__ void ica_center_kernel(const float* __restrict__ moment_sum,
                                  float* __restrict__ T, float batch_f,
                                  int64_t s) {
  const int64_t total = s * s;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += stride) {
    const float eye = (i / s == i % s) ? 1.f : 0.f;
    T[i] = sub_rn(moment_sum[i] / batch_f, eye);
  }
}

// D += stepsize * U   (the reference ascends: dictionary.add_)
__global__ void axpy_rn_kernel(float* __restrict__ D,
                               const float* __restrict__ U, float stepsize,
                               int64_t count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += stride)
    D[i] = add_rn(D[i], mul_rn(stepsize, U[i]));
}

static unsigned stream_grid(int64_t total) {
  int64_t blocks = ceil_div(total, 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// E = codes * dictionary - images  (the reconstruction residual; the trainer's
// validation metrics, training/sparse_coding.py:181-203)
extern "C" int vtc_fc_residual(const float* images, const float* dictionary,
                               const float* codes, float* residual, int64_t b,
                               int64_t n, int64_t s, void* stream) {
  VTC_REQUIRE((images && dictionary && codes && residual) || b == 0,
              "vtc_fc_residual: null pointer");
  VTC_REQUIRE(b >= 0 && n > 0 && s > 0, "vtc_fc_residual: bad sizes");
  if (b == 0) return VTC_OK;
  EpiMinus e{residual, images, n, n};
  return launch_gemm_f32<true, false>(codes, s, dictionary, n, b, n, s, 1, e,
                                      as_stream(stream));
}

extern "C" size_t vtc_subspace_alignment_gradient_workspace_bytes(
    int64_t slots, int64_t n) {
  return align_up((size_t)(slots > 0 ? slots : 1) * n * sizeof(float), 256);
}

extern "C" int vtc_subspace_alignment_gradient(
    const float* dictionary, const int32_t* index, const uint8_t* valid,
    const int32_t* atom_ptr, const int32_t* atom_slots, float* penalty_grad,
    int64_t s, int64_t n, int64_t groups, int64_t m, int dict_is_normalized,
    void* workspace, size_t workspace_bytes, void* stream) {
  VTC_REQUIRE(dictionary && index && valid && atom_ptr && atom_slots &&
                  penalty_grad, "vtc_subspace_alignment_gradient: null");
  VTC_REQUIRE(s > 0 && n > 0 && groups > 0 && m > 0,
              "vtc_subspace_alignment_gradient: bad sizes");
  const size_t lds_bytes = ((size_t)m * n + (size_t)m * m + m) * sizeof(float);
  if (m > kMaxGroup || lds_bytes > 160 * 1024) {
    set_error("vtc_subspace_alignment_gradient: group of %lld atoms x %lld "
              "pixels exceeds the LDS tile", (long long)m, (long long)n);
    return VTC_ERR_UNSUPPORTED;
  }
  if (!workspace || workspace_bytes <
      vtc_subspace_alignment_gradient_workspace_bytes(groups * m, n)) {
    set_error("vtc_subspace_alignment_gradient: workspace too small");
    return VTC_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  float* slot_grad = static_cast<float*>(workspace);
  VTC_HIP_CHECK(hipMemsetAsync(slot_grad, 0,
                               (size_t)groups * m * n * sizeof(float), st));
  if (lds_bytes > 64 * 1024)
    VTC_HIP_CHECK(hipFuncSetAttribute(
        reinterpret_cast<const void*>(alignment_slot_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(alignment_slot_kernel, dim3((unsigned)groups), dim3(256),
                     lds_bytes, st, dictionary, index, valid, slot_grad, n,
                     (int)m, dict_is_normalized);
  VTC_LAUNCH_CHECK();
  hipLaunchKernelGGL(rows_by_atom_sum_kernel, dim3((unsigned)s), dim3(256), 0,
                     st, slot_grad, atom_ptr, atom_slots, penalty_grad, s, n);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

extern "C" int vtc_fc_dict_apply(float* dictionary, const float* grad_sum,
                                 const float* hessian_diagonal,
                                 const float* penalty_grad,
                                 float alignment_penalty, int64_t global_batch,
                                 float stepsize, float lowest_code_val,
                                 int normalize, int64_t s, int64_t n,
                                 void* stream) {
  VTC_REQUIRE(dictionary && grad_sum, "vtc_fc_dict_apply: null pointer");
  VTC_REQUIRE(s > 0 && n > 0 && global_batch > 0,
              "vtc_fc_dict_apply: bad sizes");
  // rows per block: a multiple of 4 whose bytes are a multiple of 128 (see the
  // kernel); a dictionary that does not start on a line goes to one block
  int64_t rows = 32;
  while (rows > 4 && ((rows / 2) * n * 4) % 128 == 0) rows /= 2;
  if (reinterpret_cast<uintptr_t>(dictionary) % 128 != 0) rows = s;
  hipLaunchKernelGGL(fc_apply_kernel, dim3((unsigned)ceil_div(s, rows)),
                     dim3(256), 0, as_stream(stream), dictionary, grad_sum,
                     hessian_diagonal, penalty_grad, alignment_penalty,
                     (float)global_batch, stepsize, lowest_code_val, normalize,
                     s, n, (int)rows);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

extern "C" size_t vtc_code_energy_workspace_bytes(int64_t b, int64_t s,
                                                  int64_t positions) {
  if (b <= 0 || s <= 0) return 256;
  const int64_t chunks =
      positions > 1 ? b : ceil_div(b, energy_chunk_rows(b));
  return align_up((size_t)chunks * s * sizeof(float), 256);
}

extern "C" int vtc_code_energy(const float* codes, int64_t b, int64_t s,
                               int64_t positions, float* energy,
                               void* workspace, size_t workspace_bytes,
                               void* stream) {
  VTC_REQUIRE(codes && energy, "vtc_code_energy: null pointer");
  VTC_REQUIRE(b > 0 && s > 0 && positions > 0, "vtc_code_energy: bad sizes");
  if (!workspace ||
      workspace_bytes < vtc_code_energy_workspace_bytes(b, s, positions)) {
    set_error("vtc_code_energy: workspace too small");
    return VTC_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  float* partial = static_cast<float*>(workspace);
  int64_t chunks;
  if (positions == 1) {
    const int rows = energy_chunk_rows(b);
    chunks = ceil_div(b, rows);
    hipLaunchKernelGGL(energy_rows_kernel,
                       dim3((unsigned)ceil_div(s, 256), (unsigned)chunks),
                       dim3(256), 0, st, codes, b, s, rows, partial);
  } else {
    chunks = b;
    hipLaunchKernelGGL(energy_maps_kernel, dim3((unsigned)(b * s)), dim3(256),
                       0, st, codes, s, positions, partial);
  }
  VTC_LAUNCH_CHECK();
  return launch_slab_reduce(partial, (int)chunks, s, energy, st);
}

extern "C" int vtc_hessian_ema(float* hessian_diagonal, const float* energy,
                               int64_t global_batch, int64_t s, void* stream) {
  VTC_REQUIRE(hessian_diagonal && energy, "vtc_hessian_ema: null pointer");
  VTC_REQUIRE(s > 0 && global_batch > 0, "vtc_hessian_ema: bad sizes");
  hipLaunchKernelGGL(hessian_ema_kernel, dim3((unsigned)ceil_div(s, 256)),
                     dim3(256), 0, as_stream(stream), hessian_diagonal, energy,
                     (float)global_batch, s);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}

extern "C" size_t vtc_ica_moment_workspace_bytes(int64_t b, int64_t s) {
  if (b <= 0 || s <= 0) return 256;
  return align_up((size_t)b * s * sizeof(float), 256) +
         align_up((size_t)gradient_slices(b, s, s) * s * s * sizeof(float),
                  256);
}

extern "C" int vtc_ica_moment(const float* codes, float* moment_sum, int64_t b,
                              int64_t s, void* workspace,
                              size_t workspace_bytes, void* stream) {
  VTC_REQUIRE(codes && moment_sum, "vtc_ica_moment: null pointer");
  VTC_REQUIRE(b > 0 && s > 0, "vtc_ica_moment: bad sizes");
  if (!workspace || workspace_bytes < vtc_ica_moment_workspace_bytes(b, s)) {
    set_error("vtc_ica_moment: workspace too small");
    return VTC_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  Carver ws(workspace);
  float* S = ws.take<float>((size_t)b * s);
  const int slices = gradient_slices(b, s, s);
  float* slabs = ws.take<float>((size_t)slices * s * s);
  hipLaunchKernelGGL(sign_kernel, dim3(stream_grid(b * s)), dim3(256), 0, st,
                     codes, S, b * s);
  VTC_LAUNCH_CHECK();
  // M = C^T sign(C), K = b split into slabs summed in a fixed order
  EpiSlab e{slabs, s * s, s};
  int rc = launch_gemm_f32<false, false>(codes, s, S, s, s, s, b, slices, e,
                                         st);
  if (rc != VTC_OK) return rc;
  return launch_slab_reduce(slabs, slices, s * s, moment_sum, st);
}

extern "C" size_t vtc_ica_apply_workspace_bytes(int64_t s, int64_t n) {
  if (s <= 0 || n <= 0) return 256;
  return align_up((size_t)s * s * sizeof(float), 256) +
         align_up((size_t)s * n * sizeof(float), 256);
}

extern "C" int vtc_ica_apply(float* dictionary, const float* moment_sum,
                             int64_t global_batch, int64_t s, int64_t n,
                             float stepsize, void* workspace,
                             size_t workspace_bytes, void* stream) {
  VTC_REQUIRE(dictionary && moment_sum, "vtc_ica_apply: null pointer");
  VTC_REQUIRE(global_batch > 0 && s > 0 && n > 0, "vtc_ica_apply: bad sizes");
  if (!workspace || workspace_bytes < vtc_ica_apply_workspace_bytes(s, n)) {
    set_error("vtc_ica_apply: workspace too small");
    return VTC_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  Carver ws(workspace);
  float* T = ws.take<float>((size_t)s * s);
  float* U = ws.take<float>((size_t)s * n);
  hipLaunchKernelGGL(ica_center_kernel, dim3(stream_grid(s * s)), dim3(256), 0,
                     st, moment_sum, T, (float)global_batch, s);
  VTC_LAUNCH_CHECK();
  EpiStore e{U, n};
  int rc = launch_gemm_f32<true, false>(T, s, dictionary, n, s, n, s, 1, e,
                                        st);
  if (rc != VTC_OK) return rc;
  hipLaunchKernelGGL(axpy_rn_kernel, dim3(stream_grid(s * n)), dim3(256), 0,
                     st, dictionary, U, stepsize, s * n);
  VTC_LAUNCH_CHECK();
  return VTC_OK;
}
