/*
 * vtc_hip.h -- C ABI of the MI355X (gfx950) sparse-coding engine.
 *
 * One shared library, libvtc_hip.so, replaces the arithmetic behind the
 * plugin layer of spencerkent/vision-transform-codes:
 *
 *   analysis_transforms/fully_connected/ista_fista.py:14-148          -> vtc_fc_ista_fista
 *   analysis_transforms/fully_connected/subspace_ista_fista.py:23-192 -> vtc_subspace_ista_fista
 *                                                                        (+ vtc_group_gather_rows / vtc_group_gather_cols /
 *                                                                           vtc_group_scatter_add)
 *   analysis_transforms/convolutional/ista_fista.py:18-197            -> vtc_conv_ista_fista
 *   dict_update_rules/fully_connected/sc_steepest_descent.py:9-41,
 *     sc_cheap_quadratic_descent.py:11-48,
 *     subspace_sc_cheap_quadratic_descent.py:13-127                   -> vtc_fc_dict_gradient, vtc_subspace_alignment_gradient,
 *                                                                        vtc_fc_dict_apply
 *   dict_update_rules/convolutional/sc_steepest_descent.py:12-72,
 *     sc_cheap_quadratic_descent.py:14-79                             -> vtc_conv_dict_gradient, vtc_conv_dict_apply
 *   training/sparse_coding.py:154,160-161 (Hessian-diagonal EMA)      -> vtc_code_energy, vtc_hessian_ema
 *   the `torch.mm(dictionary.t(), dictionary)` of the Lipschitz step
 *     (ista_fista.py:73-74) and its `torch.symeig(...)[0][-1]`         -> vtc_gram, vtc_lambda_max
 *   training/sparse_coding.py:177-229 (validation metrics)            -> vtc_fc_residual, vtc_conv_residual, vtc_row_stats,
 *                                                                        vtc_group_norm_sum, vtc_window_minmax,
 *                                                                        vtc_rows_mean_abs_diff
 *   utils/image_processing.py:267-308, utils/dataset_generation.py
 *     :169-222 (range standardisation, whitening, patch positions
 *     and extraction)                                                 -> vtc_standardize_data_range, vtc_whiten_center_surround,
 *                                                                        vtc_draw_patch_positions, vtc_extract_patches
 *   dict_update_rules/fully_connected/ica_natural_gradient.py:6-35    -> vtc_ica_moment, vtc_ica_apply
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major float32 unless
 *     said otherwise; sizes are element counts; `stream` is a hipStream_t
 *     passed as void* (NULL = the null stream).
 *   - functions only enqueue work on `stream` and return; the single exception
 *     is an inference call with early_stopping_epsilon >= 0, which has to read
 *     one flag back per iteration exactly like the reference's
 *     `stop_early = (avg < eps)` does (ista_fista.py:143-144).
 *   - no allocation inside: scratch comes from the caller as `workspace`, sized
 *     by the matching *_workspace_bytes() query.  The per-device constants of
 *     the library (the 64 KiB FISTA momentum table) are placed by vtc_init().
 *   - return value: VTC_OK or a VTC_ERR_* code; vtc_last_error() gives text.
 *   - the gradient/apply split of the dictionary update is where a data-parallel
 *     caller places its all-reduce: gradient sums are un-normalised sums over
 *     the local batch, `global_batch` in the apply call is the divisor.
 */
#ifndef VTC_HIP_H_
#define VTC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VTC_ABI_VERSION 4

enum vtc_status {
  VTC_OK = 0,
  VTC_ERR_INVALID_ARGUMENT = 1, /* bad size / enum / null pointer           */
  VTC_ERR_UNSUPPORTED = 2,      /* valid in the reference, not built here    */
  VTC_ERR_WORKSPACE = 3,        /* workspace too small                       */
  VTC_ERR_HIP = 4               /* a HIP runtime call failed                 */
};

enum vtc_variant { VTC_ISTA = 0, VTC_FISTA = 1 };

/* ista_fista.py:107-120 */
enum vtc_threshold {
  VTC_SOFT = 0,        /* sign(c) * max(|c| - lambda*eta, 0)                 */
  VTC_SOFT_NONNEG = 1, /* max(c - lambda*eta, 0)                             */
  VTC_HARD = 2,        /* c if |c| >= lambda*eta else 0                      */
  VTC_HARD_NONNEG = 3  /* c if c >= lambda*eta else 0                        */
};

/* arithmetic used by the two contractions of an inference iteration */
enum vtc_precision {
  VTC_F32 = 0,    /* exact-f32 MFMA (v_mfma_f32_32x32x2_f32): parity mode    */
  VTC_BF16X3 = 1, /* bf16 hi/lo split, 3 MFMA products: ~f32 accuracy        */
  VTC_BF16 = 2,   /* single bf16 MFMA product, f32 accumulate: fast mode     */
  VTC_F16X3 = 3   /* f16 hi/lo split (11 + 11 bits) in power-of-two scaled
                   * units, 3 MFMA products: ~2^-21 per product at the bf16x3
                   * cost.  Fused fully-connected kernels and the convolutional
                   * inference; the tiled contractions and the convolutional
                   * dictionary gradient run it as VTC_BF16X3.                */
};

const char* vtc_version(void);
const char* vtc_last_error(void);
int vtc_abi_version(void);
/* Places the library's per-device constants on the CURRENT HIP device: the
 * FISTA momentum table beta_k = (t_k - 1) / t_{k+1} (ista_fista.py:123-127,
 * float64 recurrence rounded to f32, 16384 entries: one hipMalloc and one
 * blocking 64 KiB copy).  Thread-safe and idempotent; call it once per device
 * before enqueueing work if the first inference call must not block.  An
 * inference entry point that finds the device unprepared calls it itself. */
int vtc_init(void);

/* ---- Lipschitz step ---------------------------------------------------- */
/* gram = A^T A (transpose_a = 1, A is (rows, cols), gram is (cols, cols)) or
 * A A^T (transpose_a = 0, gram is (rows, rows)).  ista_fista.py:73-74 uses the
 * first form on the (s, n) dictionary, convolutional/ista_fista.py:104-105 the
 * second on the (s, c*kh*kw) flattened kernels. */
int vtc_gram(const float* a, int64_t rows, int64_t cols, int transpose_a,
             float* gram, void* stream);
/* Largest eigenvalue of a symmetric (n,n) matrix, n <= 1024, replacing the
 * `torch.symeig(...)[0][-1]` of ista_fista.py:73-74: single-workgroup Lanczos
 * + Sturm counts.  out (device, 3 floats) = [lambda_max, 1/lambda_max,
 * converged].  A Ritz value approaches lambda_max from below, so the kernel
 * checks itself: the top Ritz value of all steps against that of the steps up
 * to 8 earlier, going on (n <= 256: up to min(max(2n, n + 48), 256) steps) while they differ
 * by more than 1e-6 relative; converged = 1.0 when they agree or the Krylov
 * space is exhausted, 0.0 when the step limit ended the iteration -- the
 * reference's symeig is exact (ista_fista.py:72-80), a caller should treat 0
 * like its failure.  n <= 256 needs no workspace (NULL, 0); 256 < n <= 1024
 * keeps the Krylov basis in the workspace.  Larger n: VTC_ERR_UNSUPPORTED (the
 * caller then uses a library eigen-solver). */
size_t vtc_lambda_max_workspace_bytes(int64_t n);
int vtc_lambda_max(const float* symmetric, int64_t n, float* out,
                   void* workspace, size_t workspace_bytes, void* stream);
/* Same; the kernel also stores the three floats through host_mirror (may be
 * NULL), a device-visible HOST pointer (pinned memory): a caller that keeps
 * eta on the device can still notice a failed eigen-solve -- the reference's
 * symeig raises there, ista_fista.py:75-79 -- without a copy or a wait. */
int vtc_lambda_max_mirrored(const float* symmetric, int64_t n, float* out,
                            float* host_mirror, void* workspace,
                            size_t workspace_bytes, void* stream);

/* ---- fully-connected inference (row a1) ------------------------------- */
size_t vtc_fc_ista_fista_workspace_bytes(int64_t b, int64_t n, int64_t s,
                                         int precision);
/* images (b,n), dictionary (s,n), initial_codes (b,s) or NULL, codes (b,s) out.
 * stepsize = eta = 1/L.  The threshold is float(sparsity_weight)*eta in f32.
 * early_stopping_epsilon < 0 disables the stopping test (sync-free).
 * iters_run (host int*, may be NULL) receives the iterations executed.
 * Kernels behind it: 16x16 patches (n = 256) with 256 / 512 / 1024 atoms take
 * the fused persistent kernel in the split precisions, more atoms (multiples
 * of 256) the fused kernel with streamed state; 8x8 patches (n = 64) with 64 /
 * 128 / 192 atoms take an on-chip exact-f32 kernel under VTC_F32; every other
 * shape, and early stopping, the tiled contractions with the proximal step in
 * their epilogue.  `codes` is only ever written, never read-modified in
 * place. */
int vtc_fc_ista_fista(const float* images, const float* dictionary,
                      const float* initial_codes, float* codes, int64_t b,
                      int64_t n, int64_t s, float stepsize,
                      float sparsity_weight, int num_iters, int variant,
                      int threshold, float early_stopping_epsilon,
                      int precision, void* workspace, size_t workspace_bytes,
                      int* iters_run, void* stream);
/* Same, with the step size read from DEVICE memory (e.g. out + 1 of
 * vtc_lambda_max): the reference keeps `stepsize` as a 0-d device tensor and
 * never brings it to the host (ista_fista.py:72-80), so a training step
 * without early stopping needs no host synchronisation at all.  The threshold
 * is float(sparsity_weight) * *stepsize_dev, formed on the device in f32. */
int vtc_fc_ista_fista_dev(const float* images, const float* dictionary,
                          const float* initial_codes, float* codes, int64_t b,
                          int64_t n, int64_t s, const float* stepsize_dev,
                          float sparsity_weight, int num_iters, int variant,
                          int threshold, float early_stopping_epsilon,
                          int precision, void* workspace,
                          size_t workspace_bytes, int* iters_run, void* stream);

/* ---- subspace inference (row a3) --------------------------------------- */
/* index/valid describe the padded (G, m) layout: slot g*m+j holds dictionary
 * row index[g*m+j] when valid[g*m+j] != 0 and is padding otherwise. */
int vtc_group_gather_rows(const float* dictionary, const int32_t* index,
                          const uint8_t* valid, float* grouped_dictionary,
                          int64_t slots, int64_t n, void* stream);
int vtc_group_gather_cols(const float* codes, const int32_t* index,
                          const uint8_t* valid, float* grouped_codes,
                          int64_t b, int64_t s, int64_t slots, void* stream);
/* codes (b,s): codes[:, a] = sum of grouped_codes[:, t] over the slots t of
 * atom a, listed in increasing order in CSR form: atom_slots[atom_ptr[a] ..
 * atom_ptr[a+1]) (subspace_ista_fista.py:184-190 adds them in that order). */
int vtc_group_scatter_add(const float* grouped_codes, const int32_t* atom_ptr,
                          const int32_t* atom_slots, float* codes, int64_t b,
                          int64_t s, int64_t slots, void* stream);
size_t vtc_subspace_ista_fista_workspace_bytes(int64_t b, int64_t n,
                                               int64_t groups, int64_t m);
/* grouped_dictionary (G*m, n); initial_grouped (b, G*m) or NULL;
 * grouped_codes (b, G*m) out.  Proximal step scales each group of m slots by
 * max(1 - lambda*eta/||group||_2, 0) (subspace_ista_fista.py:149-156).
 * precision: VTC_F32 (exact-f32 MFMA) or VTC_BF16X3 (bf16 hi/lo split tiles,
 * needs n and G*m to be multiples of 4). */
int vtc_subspace_ista_fista(const float* images,
                            const float* grouped_dictionary,
                            const float* initial_grouped, float* grouped_codes,
                            int64_t b, int64_t n, int64_t groups, int64_t m,
                            float stepsize, float sparsity_weight,
                            int num_iters, int variant,
                            float early_stopping_epsilon, int precision,
                            void* workspace, size_t workspace_bytes,
                            int* iters_run, void* stream);

/* ---- convolutional inference (row a4) ---------------------------------- */
typedef struct vtc_conv_geometry {
  int64_t b;            /* images in the batch                               */
  int32_t c, h, w;      /* channels, padded height, padded width             */
  int32_t s, kh, kw;    /* kernels, kernel height, kernel width              */
  int32_t stride_v, stride_h;
  int32_t has_padding;  /* 0: padding_dims was None (mask of ones)           */
  int32_t pad_lead_v, pad_trail_v, pad_lead_h, pad_trail_h;
} vtc_conv_geometry;

int vtc_conv_code_dims(const vtc_conv_geometry* g, int32_t* code_h,
                       int32_t* code_w);
size_t vtc_conv_ista_fista_workspace_bytes(const vtc_conv_geometry* g);
/* 1 when the geometry has a split-precision (VTC_F16X3 / VTC_BF16X3) path:
 * stride 1, square kernels of 5, 8, 11 or 16, operand planes within the
 * 160 KiB LDS (1 to 3 image channels at 11x11 and 16x16). */
int vtc_conv_x3_supported(const vtc_conv_geometry* g);
/* images_padded (b,c,h,w), dictionary (s,c,kh,kw), codes (b,s,code_h,code_w).
 * precision: VTC_F32 (direct f32 convolutions, fixed summation order),
 * VTC_F16X3 or VTC_BF16X3 (both convolutions as hi/lo split MFMA
 * contractions, f16 in power-of-two scaled units or bf16;
 * VTC_ERR_UNSUPPORTED unless vtc_conv_x3_supported).  With kernels up to
 * 11x11, more than 32 of them and one image channel the split modes run one
 * fused launch per iteration (analysis, proximal step and the next residual) on code maps kept
 * in an internal tile order; the caller's layout is written by the last one. */
int vtc_conv_ista_fista(const float* images_padded, const float* dictionary,
                        const float* initial_codes, float* codes,
                        const vtc_conv_geometry* g, float stepsize,
                        float sparsity_weight, int num_iters, int variant,
                        int threshold, float early_stopping_epsilon,
                        int precision, void* workspace,
                        size_t workspace_bytes, int* iters_run, void* stream);

/* ---- dictionary update, fully-connected (rows a5, a6, a7) --------------- */
size_t vtc_fc_dict_gradient_workspace_bytes(int64_t b, int64_t n, int64_t s);
/* grad_sum (s,n) = codes^T (codes dictionary - images): NOT divided by b. */
int vtc_fc_dict_gradient(const float* images, const float* dictionary,
                         const float* codes, float* grad_sum, int64_t b,
                         int64_t n, int64_t s, void* workspace,
                         size_t workspace_bytes, void* stream);
/* penalty_grad (s,n) = sum over groups of the |cos| alignment gradients
 * (subspace_sc_cheap_quadratic_descent.py:91-127); index/valid and the CSR
 * inverse map atom_ptr/atom_slots as above.  Groups of up to 32 atoms. */
size_t vtc_subspace_alignment_gradient_workspace_bytes(int64_t slots,
                                                       int64_t n);
int vtc_subspace_alignment_gradient(
    const float* dictionary, const int32_t* index, const uint8_t* valid,
    const int32_t* atom_ptr, const int32_t* atom_slots, float* penalty_grad,
    int64_t s, int64_t n, int64_t groups, int64_t m, int dict_is_normalized,
    void* workspace, size_t workspace_bytes, void* stream);
/* D -= (stepsize * (grad_sum/global_batch + alignment_penalty*penalty_grad))
 *      / (hessian_diagonal + lowest_code_val)         [divide iff hessian given]
 * then D /= ||row||_2 iff normalize.  hessian_diagonal, penalty_grad may be
 * NULL.  In place on `dictionary` (s,n). */
int vtc_fc_dict_apply(float* dictionary, const float* grad_sum,
                      const float* hessian_diagonal, const float* penalty_grad,
                      float alignment_penalty, int64_t global_batch,
                      float stepsize, float lowest_code_val, int normalize,
                      int64_t s, int64_t n, void* stream);

/* ---- dictionary update, convolutional (rows a8, a9) --------------------- */
size_t vtc_conv_dict_gradient_workspace_bytes(const vtc_conv_geometry* g);
/* grad_sum (s,c,kh,kw) = sum_{b,p,q} codes[b,s,p,q] * r[b,c,p*sv+dy,q*sh+dx],
 * r = mask * (synthesis(codes) - images_padded): NOT divided by b.
 * precision: VTC_F32 (direct kernels) or VTC_BF16X3 (residual and gradient as
 * bf16 hi/lo split MFMA contractions; needs vtc_conv_x3_supported). */
int vtc_conv_dict_gradient(const float* images_padded, const float* dictionary,
                           const float* codes, float* grad_sum,
                           const vtc_conv_geometry* g, int precision,
                           void* workspace, size_t workspace_bytes,
                           void* stream);
/* g = grad_sum/global_batch; g /= (h + lowest_code_val) iff hessian given;
 * g *= ||D||_F/||g||_F; D -= stepsize*g; per-kernel l2 normalise iff normalize.
 * scratch: s*kernel_elems floats of device memory. */
int vtc_conv_dict_apply(float* dictionary, const float* grad_sum,
                        const float* hessian_diagonal, int64_t global_batch,
                        float stepsize, float lowest_code_val, int normalize,
                        int64_t s, int64_t kernel_elems, float* scratch,
                        void* stream);

/* ---- Hessian-diagonal EMA (training/sparse_coding.py:154,160-161) ------- */
size_t vtc_code_energy_workspace_bytes(int64_t b, int64_t s,
                                       int64_t positions);
/* energy[j] = sum_b sum_positions codes[b, j, :]^2  (positions = 1 for the
 * fully-connected (b,s) layout, code_h*code_w for (b,s,h,w)). */
int vtc_code_energy(const float* codes, int64_t b, int64_t s,
                    int64_t positions, float* energy, void* workspace,
                    size_t workspace_bytes, void* stream);
/* h = 0.99*h + (energy/global_batch)/100 */
int vtc_hessian_ema(float* hessian_diagonal, const float* energy,
                    int64_t global_batch, int64_t s, void* stream);

/* ---- validation metrics ---------------------
 * Device reductions behind training/sparse_coding.py:177-229 `compute_metrics`
 * (LASSO loss terms, normalised L0, pSNR, dictionary change); the host side
 * (training/sparse_coding.py of this package) turns them into the reference's
 * dictionary of scalars. */
/* residual (b,n) = codes dictionary - images */
int vtc_fc_residual(const float* images, const float* dictionary,
                    const float* codes, float* residual, int64_t b, int64_t n,
                    int64_t s, void* stream);
/* residual (b,c,h,w) = mask * (conv_transpose2d(codes, dictionary) -
 * images_padded); the mask zeroes the padding the reference crops away */
int vtc_conv_residual(const float* images_padded, const float* dictionary,
                      const float* codes, float* residual,
                      const vtc_conv_geometry* g, void* stream);
/* per row of x (rows, cols): sum x^2, sum |x|, count of non-zeros; any of the
 * three outputs (rows floats each) may be NULL */
int vtc_row_stats(const float* x, int64_t rows, int64_t cols, float* sumsq,
                  float* l1, float* l0, void* stream);
/* out[r] = sum over groups of ||codes[r, group]||_2 ; index/valid are the
 * padded (G, m) group tables of vtc_group_gather_cols */
int vtc_group_norm_sum(const float* codes, const int32_t* index,
                       const uint8_t* valid, float* out, int64_t b, int64_t s,
                       int64_t groups, int64_t m, void* stream);
/* out_min_max[0..1] = min, max over the window of outer x rows x cols
 * elements at x[o*outer_pitch + r*row_pitch + c] (pitches in elements) */
size_t vtc_window_minmax_workspace_bytes(void);
int vtc_window_minmax(const float* x, int64_t outer, int64_t rows,
                      int64_t cols, int64_t outer_pitch, int64_t row_pitch,
                      float* out_min_max, void* workspace,
                      size_t workspace_bytes, void* stream);
/* out[r] = mean over columns of |a[r,c] - b[r,c]| */
int vtc_rows_mean_abs_diff(const float* a, const float* b, int64_t rows,
                           int64_t cols, float* out, void* stream);

/* ---- patch pipeline --------------------------
 * utils/image_processing.py:267-308 whiten_center_surround (float64 FFT
 * filtering as at :63-92, through hipFFT opened at first use) and the 'patch'
 * operation of utils/dataset_generation.py:184-222.  Images are channel-last
 * (count, h, w, c) float32 as in the reference. */
size_t vtc_whiten_center_surround_workspace_bytes(int64_t count, int32_t h,
                                                  int32_t w, int32_t c);
/* norm_and_threshold != 0 (the reference function's default,
 * image_processing.py:302-304): the transfer function is divided by its
 * maximum over the frequency grid and values below 1e-3 are raised to 1e-3;
 * 0 is what the dataset pipeline passes (dataset_generation.py:231-238) */
int vtc_whiten_center_surround(const float* images, float* out, int64_t count,
                               int32_t h, int32_t w, int32_t c,
                               float cutoff_low, float cutoff_high,
                               int norm_and_threshold, void* workspace,
                               size_t workspace_bytes, void* stream);
/* 'standardize_data_range' of utils/dataset_generation.py:169-183, the first
 * operation of every example pipeline: out = (images - min) / (max - min)
 * over the whole array (count floats, any shape), float32 arithmetic as
 * numpy's.  min_max (device, 2 floats) receives [min, max]; the reference
 * asserts max > min, a caller reads them back for that.  workspace:
 * vtc_window_minmax_workspace_bytes(). */
int vtc_standardize_data_range(const float* images, float* out, int64_t count,
                               float* min_max, void* workspace,
                               size_t workspace_bytes, void* stream);
/* HOST-side helper (no device work): the patch positions of
 * utils/dataset_generation.py:205-214 -- per patch np.random.randint(0,
 * num_images), randint(edge_buffer, max_vert[image]), randint(edge_buffer,
 * max_horz[image]), in that order -- drawn from numpy's legacy generator
 * state: key (624 words) and *pos are RandomState.get_state()[1:3], updated
 * in place.  img_index / vert / horz: num_samples int32 each, host memory.
 * 131 072 positions take ~2 ms; the reference's Python loop 0.3-0.4 s. */
int vtc_draw_patch_positions(uint32_t* key, int32_t* pos, int64_t num_samples,
                             int32_t num_images, int32_t edge_buffer,
                             const int32_t* max_vert, const int32_t* max_horz,
                             int32_t* img_index, int32_t* vert, int32_t* horz);
/* patches (num, ph*pw*c): patch p = images[img_index[p],
 * vert[p]:vert[p]+ph, horz[p]:horz[p]+pw, :] flattened; positions come from
 * the caller's random number generator (int32 device arrays) */
int vtc_extract_patches(const float* images, const int32_t* img_index,
                        const int32_t* vert, const int32_t* horz,
                        float* patches, int64_t num, int32_t h, int32_t w,
                        int32_t c, int32_t ph, int32_t pw, void* stream);

/* ---- ICA natural gradient (the sibling update
 * rule dict_update_rules/fully_connected/ica_natural_gradient.py:26-35):
 *   D += stepsize * ((codes^T sign(codes)) / b - I) D
 * split at the quantity a data-parallel caller sums over ranks. */
size_t vtc_ica_moment_workspace_bytes(int64_t b, int64_t s);
/* moment_sum (s,s) = codes^T sign(codes): NOT divided by b */
int vtc_ica_moment(const float* codes, float* moment_sum, int64_t b, int64_t s,
                   void* workspace, size_t workspace_bytes, void* stream);
size_t vtc_ica_apply_workspace_bytes(int64_t s, int64_t n);
int vtc_ica_apply(float* dictionary, const float* moment_sum,
                  int64_t global_batch, int64_t s, int64_t n, float stepsize,
                  void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VTC_HIP_H_ */
