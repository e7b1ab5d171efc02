/*
 * eofx.h -- C ABI of the MI355X-native EOF / randomized-SVD engine (libeofx.so).
 *
 * This is the drop-in boundary for the one hot path of xarray-contrib/xeofs
 * (reference v3.0.4, paths relative to /root/reference):
 *
 *   preprocess  : Scaler.fit/transform      xeofs/preprocessing/scaler.py:69-154
 *                 Sanitizer.fit/transform   xeofs/preprocessing/sanitizer.py:46-126
 *                 total_variance            xeofs/utils/xarray_utils.py:236-253
 *   decompose   : the callable handed to xr.apply_ufunc in Decomposer._svd
 *                 xeofs/linalg/decomposer.py:141-146,252-263
 *                 (= sklearn.utils.extmath.randomized_svd(X, n_components, random_state))
 *                 + sign rule xeofs/utils/xarray_utils.py:273-301
 *   project     : xr.dot(X, components)     xeofs/single/eof.py:129, cross/cpcca.py:204-205
 *   cross-cov   : X^H Y/(n-1) + its rSVD    xeofs/cross/cpcca.py:168-225,1007-1015
 *   hilbert     : analytic signal + padding xeofs/utils/hilbert_transform.py:40-114
 *   and, for the callers either side of the path (panel-level building blocks):
 *   rotation    : Varimax / Promax step     xeofs/linalg/_numpy/_rotation.py:6-187
 *   PCA         : small-side Gram matrix    xeofs/preprocessing/pca.py:94-171
 *   bootstrap   : row resampling + centring xeofs/validation/bootstrapper.py:78-91
 *   patterns    : per-feature norms         xeofs/utils/optional/statistics.py:50-54
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / xarray types.
 *   - every function returns an int status: 0 = ok, negative = error
 *     (see EOFX_ERR_*; eofx_last_error() gives the message).  The Python shell
 *     maps the codes onto the reference's exception types.
 *   - data pointers documented "host|device" may point to host memory or to HIP
 *     device memory of the context's GPU; the library detects which
 *     (hipPointerGetAttributes) and never frees caller memory.
 *   - matrices are row-major.  `sample` = rows (time), `feature` = columns (space).
 *   - all work is enqueued on the context's HIP stream; functions that write host
 *     outputs synchronise that stream before returning, the others do not.
 *   - deterministic: no floating-point atomics, fixed reduction trees; the same
 *     inputs give bitwise identical outputs (reference contract
 *     tests/linalg/test_decomposer.py:164-192).
 */
#ifndef EOFX_H
#define EOFX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EOFX_ABI_VERSION 1
#define EOFX_MAX_SKETCH 256 /* widest sketch k + n_oversamples; <= 64 is factorised on the device */

#define EOFX_OK 0
#define EOFX_ERR_ARG (-1)          /* bad argument                       -> ValueError        */
#define EOFX_ERR_HIP (-2)          /* HIP runtime failure                -> RuntimeError      */
#define EOFX_ERR_PARTIAL_NAN (-3)  /* sanitizer.py:115-122               -> ValueError        */
#define EOFX_ERR_NAN_MISMATCH (-4) /* sanitizer.py:109-113               -> ValueError        */
#define EOFX_ERR_RANK (-5)         /* decomposer.py:97-100               -> ValueError        */
#define EOFX_ERR_LINALG (-6)       /* decomposer.py:265-270              -> LinAlgError       */
#define EOFX_ERR_NOMEM (-7)        /* device allocation failed           -> MemoryError       */
#define EOFX_ERR_SHAPE (-8)        /* cpcca.py:1012-1014 sample mismatch -> ValueError        */

typedef struct eofx_ctx eofx_ctx; /* one GPU + one stream + scratch            */
typedef struct eofx_mat eofx_mat; /* a resident preprocessed (sample x feature) matrix */

/* ---- context ----------------------------------------------------------- */
int eofx_abi_version(void);
/* stream: a hipStream_t (NULL = the device's default stream). */
int eofx_ctx_create(int device, void *stream, eofx_ctx **out);
int eofx_ctx_destroy(eofx_ctx *ctx);
int eofx_ctx_synchronize(eofx_ctx *ctx);
/* Re-bind the context to another hipStream_t (work queued on the old stream is awaited first).                  */
int eofx_ctx_set_stream(eofx_ctx *ctx, void *stream);
const char *eofx_last_error(const eofx_ctx *ctx);
/* Destroyed resident matrices leave their HBM buffers in a per-context cache (allocating
 * tens of GB costs more than a fit); eofx_ctx_trim returns that cache to the device. */
int eofx_ctx_trim(eofx_ctx *ctx);
/* Measurement aid: when enabled, every launch of the dominant kernel (atb_f32, the
 * A^T B product that streams the matrix) is bracketed by HIP events on the context
 * stream.  _read synchronises, returns and resets: the number of launches, their summed
 * duration, the flops issued (2*K*M*L on padded sizes) and the matrix bytes streamed.  */
/* Arithmetic of the matrix passes.  Every pass reads the f32 matrix once from HBM; what
 * differs is how the tall-skinny product is issued on the matrix cores:
 *   EOFX_PREC_F32     exact-f32 MFMA (v_mfma_f32_32x32x2_f32), bitwise an fmaf chain
 *   EOFX_PREC_BF16X3  operands split into 2 bf16 terms, 3 cross products  (~2^-16 per product)
 *   EOFX_PREC_BF16X6  operands split into 3 bf16 terms, 6 cross products  (~2^-23, f32 class)
 * power_passes applies to the 2*n_iter power-iteration products and the range-basis product A Z
 * (they only have to find a subspace), final_passes to the projection A^T Q, which decides the
 * singular values, and to eofx_project.
 * Default: (F16X3, F16X3) -- every pass HBM-bound, f32-class accuracy (singular values within 1e-6
 * of the float64 oracle on converged modes, like the exact-f32 kernel). */
#define EOFX_PREC_F32 0
#define EOFX_PREC_BF16X3 1
#define EOFX_PREC_BF16X6 2
#define EOFX_PREC_F16X3 3 /* operands scaled by exact powers of two and split into 2 fp16 terms (11+11
                            bits), 3 cross products: ~2^-22 per product at the cost of BF16X3 */
#define EOFX_PREC_F64 4   /* float64 multiply-accumulate on the fp64 matrix cores (v_mfma_f64_16x16x4_f64): exact products
                           * of the float32 data and panel, float64 sums -- the reference's arithmetic after it promotes
                           * the field; MFMA-bound, about 3x the time of the split-fp16 pass.  For spectra whose wanted
                           * modes lie more than ~500x below the leading one.                       */
int eofx_ctx_set_precision(eofx_ctx *ctx, int power_passes, int final_passes);
/* 1 when the drivers re-normalise the tall (rows_pad x L) panel between the two products of EVERY power iteration:
 * small panels (<= 16 MiB) and the float64 mode (the first iteration always does).  Exported so that panel-level
 * drivers follow the same rule.                                                                                  */
int eofx_orth_tall_rule(int64_t tall_rows_pad, int L, int precision_power);
/* 1 when the spectrum is peaked enough (sqrt of the ratio of the extreme eigenvalues of the small-side Gram matrix of
 * the first power iteration > 30) for the drivers to keep that step in the remaining iterations as well.  G: host,
 * leading l x l block, row stride ld.                                                                           */
int eofx_peaked_spectrum(const double *G, int ld, int l);
int eofx_ctx_profile(eofx_ctx *ctx, int enable);
int eofx_ctx_profile_read(eofx_ctx *ctx, int64_t *launches, double *total_ms, double *flops,
                          double *bytes);
/* the same launches of the LAST eofx_ctx_profile_read by streaming kernel: [0] the atb kernels (operand read with the
 * reduction axis strided: X^T Z on the field -- including the statistics-carrying first pass of eofx_fit_f32 --, X Y on
 * the sample-contiguous layout), [1] axb_f16_kernel (X Y on the field in place), [2] unused */
int eofx_ctx_profile_by_kernel(const eofx_ctx *ctx, int64_t *launches3, double *ms3);

/* ---- resident matrix ---------------------------------------------------
 * An eofx_mat holds the preprocessed matrix twice in HBM, zero padded:
 *   X  [n_pad x p_pad] feature-contiguous   (streams X^T Z)
 *   Xt [p_pad x n_pad] sample-contiguous    (streams X Y)
 * with n_pad, p_pad multiples of 512.  See DESIGN.md "Data layout".       */

/* Scaler + Sanitizer + total variance, fused (R1,R2,R4,R6 of SURVEY.md 8a).
 *   X              host|device, n x P float32, may contain NaN
 *   center/standardize  Scaler with_center / with_std
 *   feat_weights   host, P doubles (coslat*weights) or NULL (= ones)
 *   check_nans     1: raise EOFX_ERR_PARTIAL_NAN on isolated NaNs, drop all-NaN
 *                  features and samples
 * outputs (host, each may be NULL):
 *   mean[P], std[P]      fitted statistics (NaN for all-NaN features)
 *   valid_feature[P], valid_sample[n]   0/1 masks
 *   n_out, p_out         shape of the compacted matrix
 *   total_variance       sum_f var(X_f, ddof=1) of the transformed matrix        */
int eofx_preprocess_f32(eofx_ctx *ctx, const float *X, int64_t n, int64_t P, int center,
                        int standardize, const double *feat_weights, int check_nans,
                        eofx_mat **out, double *mean, double *std, uint8_t *valid_feature,
                        uint8_t *valid_sample, int64_t *n_out, int64_t *p_out,
                        double *total_variance);

/* Preprocessor.transform on new data with fitted state (scaler.py:128-154,
 * sanitizer.py:80-126).  mean/std may be NULL (no centring / no scaling).
 * Raises EOFX_ERR_NAN_MISMATCH if the NaN-feature pattern differs from
 * valid_feature, EOFX_ERR_PARTIAL_NAN on isolated NaNs.                       */
int eofx_apply_f32(eofx_ctx *ctx, const float *X, int64_t n, int64_t P, const double *mean,
                   const double *std, const double *feat_weights, const uint8_t *valid_feature,
                   int check_nans, eofx_mat **out, uint8_t *valid_sample, int64_t *n_out);

/* Adopt an already preprocessed dense matrix (host|device, n x p, leading dim ld). */
int eofx_mat_from_dense_f32(eofx_ctx *ctx, const float *X, int64_t n, int64_t p, int64_t ld,
                            eofx_mat **out);
int eofx_mat_destroy(eofx_ctx *ctx, eofx_mat *m);
int eofx_mat_shape(const eofx_mat *m, int64_t *n, int64_t *p, int64_t *n_pad, int64_t *p_pad);
/* copy the resident matrix back (host|device dst, n x p dense).                */
int eofx_mat_download_f32(eofx_ctx *ctx, const eofx_mat *m, float *dst);

/* ---- layout policy -------------------------------------------------------
 * mode 0  a resident matrix holds the preprocessed field twice (feature- and sample-contiguous): every pass streams a
 *         plain float32 matrix.
 * mode 1  ("raw") eofx_preprocess_f32 / eofx_apply_f32 write the sample-contiguous layout only: the products that stream
 *         the feature-contiguous layout (X^T Z) read the RAW field and apply the Scaler map
 *         (xeofs/preprocessing/scaler.py:153, (x - mean) / std * weights) on the fly -- one third less traffic in the
 *         preprocessor, one layout less in HBM.
 * mode 2  ("in place") nothing is written: the preprocessor is the statistics pass alone, X^T Z streams the field as in
 *         mode 1 and X Y streams it along its rows (axb_f16_kernel) -- the field is read where it lies, 1x its size in
 *         HBM instead of 3x.  Entry points that need a layout (other precisions than EOFX_PREC_F16X3, the Hilbert
 *         transform, Gram matrices, download, resampling) build it on demand from the field through the same map.
 * Modes 1 and 2 apply whenever nothing is dropped (no all-NaN feature or sample) and the raw field is 16-byte aligned
 * with P % 4 == 0; otherwise mode 0 is used silently (but see mode 3).  A DEVICE field handed to the preprocessor must stay alive and
 * unmodified until eofx_mat_release_raw (which first builds what only the field could provide) or eofx_mat_destroy; a
 * host field is staged and owned by the matrix.  eofx_mat_layout: bit 0 / bit 1 of *layouts = feature- /
 * sample-contiguous layout present.
 * mode 3  mode 2, and a field with all-NaN grid points (a land / sea mask; sanitizer.py:80-126 drops them) may keep them
 *         as ZERO columns of the in-place matrix instead of being compacted into a second copy: their scale is 0 in the
 *         map and the streaming kernels' MASK variants AND their bits to +0, so no product, Gram matrix or norm changes.
 *         Taken when fewer than 40 % of the features are masked and n < (valid features).  eofx_mat_shape then reports
 *         p = P (the physical column count), eofx_mat_masked the number of valid ones; factors with a feature axis come
 *         back with P rows (zeros at the masked features) and are expected that way: the CALLER compacts / scatters
 *         (xeofs_amd/engine.py does).  Only for callers prepared for that.                                        */
int eofx_ctx_set_layout(eofx_ctx *ctx, int mode);
/* on != 0: the next eofx_preprocess_f32 in the in-place layout (mode 2) also writes the RAW field in the sample-contiguous
 * layout while it takes the column statistics (one pass, 14.3 ms at 8000 x 1 036 800 instead of 5.5 ms + a 14 ms copy later),
 * for the eofx_hilbert_f32 call that follows: the Hilbert stage (xeofs/utils/hilbert_transform.py:40-72) reads it through the
 * Scaler map and releases it.  Series of up to 8192 samples (the one-kernel route, two features per transform); otherwise
 * without effect.  Reset it to 0 after the call. */
int eofx_ctx_set_sample_raw(eofx_ctx *ctx, int on);
/* masked: 1 for a mode-3 matrix with zero columns; p_valid: its number of valid features (= p otherwise). */
int eofx_mat_masked(const eofx_mat *m, int *masked, int64_t *p_valid);
int eofx_mat_release_raw(eofx_ctx *ctx, eofx_mat *m);
int eofx_mat_layout(const eofx_mat *m, int *layouts, int *has_raw);
/* Build the sample-contiguous layout ahead of the passes (an in-place matrix needs none: this is for callers that
 * decompose the same matrix many times -- the bootstrapper (xeofs/validation/bootstrapper.py:78-100) -- whose X Y passes
 * run faster over it).  only_if_room: do nothing unless HBM holds one more copy of the field with 8 GB to spare;
 * *built (may be NULL) = 1 when the layout exists afterwards.  Masked in-place matrices are left as they are.        */
int eofx_mat_ensure_sample_layout(eofx_ctx *ctx, eofx_mat *m, int only_if_room, int *built);
/* The inverse: drop that layout again when the matrix can rebuild it (in-place / raw mode keeps the raw field and its
 * map); the memory returns to the context's pool.  No-op for a matrix whose only data is that layout. */
int eofx_mat_release_sample_layout(eofx_ctx *ctx, eofx_mat *m);

/* ---- randomized SVD (the decomposer seam) ------------------------------
 * Replaces randomized_svd(X, n_components=k, random_state) at decomposer.py:146.
 *   omega   host, (min(n,p) x (k+n_oversamples)) float32 Gaussian test matrix, drawn
 *           by the caller exactly as sklearn does so results are seed-compatible:
 *           RandomState(seed).normal(size=(min(n,p), k+n_oversamples)).astype(float32)
 *   n_iter  power iterations; <0 = sklearn "auto" (7 if k < 0.1*min(n,p) else 4)
 *   flip    1: apply the xeofs sign rule (xarray_utils.py:273-301) to U and V
 * outputs host|device: U [n x k], s [k], V [p x k] (V = VT^T, decomposer.py:226). */
int eofx_rsvd_f32(eofx_ctx *ctx, const eofx_mat *m, int k, int n_oversamples, int n_iter,
                  const float *omega, int flip, float *U, float *s, float *V);

/* ---- the fused fit (Scaler.fit + Sanitizer + Decomposer.fit in ONE call) ----
 * Replaces, for the model classes, the pair eofx_preprocess_f32 -> eofx_rsvd_f32, i.e. xeofs'
 * Preprocessor.fit_transform (xeofs/single/base_model_single_set.py:123-161 -> preprocessing/scaler.py:69-154,
 * sanitizer.py:46-126) followed by Decomposer.fit (xeofs/single/eof.py:85-97 -> linalg/decomposer.py:76-226).
 * Same arguments and outputs as the two calls it replaces; what changes is the traffic: the column statistics are taken
 * DURING the first pass of the randomized SVD (Y = X'^T Omega is linear in the data, so it is computed with a
 * provisional per-feature shift and corrected by a rank-one term once the mean is known; xeofs_amd/csrc/eofx_fit.hpp),
 * and the field is read 2 n_iter + 2 times instead of 2 n_iter + 3.  Taken when the layout policy is "in place"
 * (eofx_ctx_set_layout 2), the passes run in EOFX_PREC_F16X3, n < P, P % 4 == 0, k + n_oversamples < 64, != 32 and < n, and the
 * field holds no NaN / inf; otherwise (and whenever the first pass meets a NaN or exceeds its provisional fp16 range)
 * the call runs the two-step path itself -- the NaN policies are the Sanitizer's either way.  *fused tells which.
 *   omega       host, omega_rows x (k + n_oversamples): the sketch for the UNcompacted shape (omega_rows >= min(n, P));
 *               numpy fills it row by row, so its leading rows are the reference's draw when samples / features drop out
 *   U, V        host|device, room for n x k and P x k; filled densely with n_out x k and (*out)->p x k values
 *   *out        the resident (in-place when possible) matrix, as from eofx_preprocess_f32                          */
int eofx_fit_f32(eofx_ctx *ctx, const float *X, int64_t n, int64_t P, int center, int standardize,
                 const double *feat_weights, int check_nans, int k, int n_oversamples, int n_iter,
                 const float *omega, int64_t omega_rows, int flip, eofx_mat **out, double *mean, double *std,
                 uint8_t *valid_feature, uint8_t *valid_sample, int64_t *n_out, int64_t *p_out,
                 double *total_variance, float *U, float *s, float *V, int *fused);
/* The statistics-carrying first pass on its own, for drivers that put collectives between the passes (the
 * feature-sharded fit, SURVEY.md 8e: a rank's X_g^T Z needs no communication, so every rank takes the statistics of its
 * shard while it computes it).  Yp [round_up(P, 512) x L] (device) = X'^T Zn for the device panel Zn [n_pad x L] whose
 * first l < L columns are in use, plus everything eofx_preprocess_f32 returns (same arguments).  Same eligibility and
 * fallback as eofx_fit_f32 (the fallback is eofx_preprocess_f32 + eofx_panel_tmul_f32; if it dropped samples, *n_out < n
 * and Yp is NOT computed: the caller re-imports the rows of Z that survive).                                      */
int eofx_fit_first_f32(eofx_ctx *ctx, const float *X, int64_t n, int64_t P, int center, int standardize,
                       const double *feat_weights, int check_nans, const float *Zn, int L, int l, float *Yp,
                       eofx_mat **out, double *mean, double *std, uint8_t *valid_feature, uint8_t *valid_sample,
                       int64_t *n_out, int64_t *p_out, double *total_variance, int *fused);
/* info3: [0] 1 when the last eofx_fit_f32 / eofx_fit_first_f32 took the fused path, [1] milliseconds of its non-pass work (probe kernel,
 * statistics finalisation, rank-one correction; HIP events, only while profiling is on), [2] why it did not: 0 fused,
 * -1 not eligible, 1 NaN / constant data in the sampled rows (a feature that is NaN in SOME of them; or in all of them
 * while layout mode 3 is not selected), 3 NaN or inf in the field, 4 provisional fp16 range exceeded (or a finite value
 * in a feature whose sampled rows were all NaN), 5 both (also a value so far outside the provisional range that its fp16
 * conversion is infinite), 6 all-NaN grid points outside the range of the masked in-place
 * layout (40 % of the features or more, or n >= valid features), 7 standardize with a feature whose standard deviation
 * is below 2^-14 of the field's largest value (mixed units: the first pass splits raw values against one scale; the
 * two-step path maps every feature to unit variance first).  In layout mode 3 a field whose NaNs are all-NaN grid
 * points stays on the fused path: the first pass confines and verifies them by itself (xeofs_amd/csrc/eofx_fit.hpp).  */
int eofx_ctx_fit_info(const eofx_ctx *ctx, double *info3);
/* ---- feature-sharded fit: one process per GPU, the space axis split over the ranks ---------------------
 * Rank g holds X_g = X[:, p_g] (all n samples, its slice of the stacked feature axis).  With a communicator attached to
 * the context, eofx_fit_sharded_f32 is eofx_fit_f32 on that slice: the statistics of the slice ride on its
 * (communication-free) first product X_g^T Omega, every later sample-side product X_g Y_g is followed by ONE all-reduce of
 * the n x L float32 panel, every Gram matrix of a feature-side panel by one of L x L float64, the sign rule by one of 2 k
 * extrema, the total variance by one scalar -- all enqueued on the context's stream between the kernels, no host round
 * trip and no second stream.  Results: U, s replicated (bit-identical on every rank), V = this rank's rows.
 *
 * Two bindings of the collective.  RCCL: rank 0 calls eofx_comm_unique_id and hands the 128 bytes to every rank (any side
 * channel: torch.distributed, MPI, a file); every rank then calls eofx_ctx_comm_init_rccl (ncclCommInitRank; the library
 * is opened at run time -- an already loaded librccl.so.1 is reused, e.g. PyTorch's --, nothing links against it).
 * Callback: `fn` must all-reduce `count` elements of the DEVICE buffer in place, ordered after the work already queued on
 * `stream` and before what is queued next (e.g. synchronise, reduce through a host library, copy back): for tests and for
 * hosts that own their communicator.  dtype: 0 float32, 1 float64, 2 int32; op: 0 sum, 1 max, 2 min; returns 0 on success. */
typedef int (*eofx_allreduce_fn)(void *user, void *device_buf, int64_t count, int dtype, int op, void *stream);
int eofx_comm_unique_id(char *id128);
int eofx_ctx_comm_init_rccl(eofx_ctx *ctx, const char *id128, int world, int rank);
int eofx_ctx_comm_set_callback(eofx_ctx *ctx, eofx_allreduce_fn fn, void *user, int world, int rank);
int eofx_ctx_comm_clear(eofx_ctx *ctx);
/* collectives issued on this context since the last call, their bytes, and (while eofx_ctx_profile is on) their
 * milliseconds from events on the context's stream; resets the counters */
int eofx_ctx_comm_stats(eofx_ctx *ctx, int64_t *calls, int64_t *bytes, double *ms);
/* one round of every collective the sharded fit uses (float32 sum / max / min, float64 sum, int32 max) on known values;
 * *ok = 1 when all results are what `world` ranks must produce.  Collective: every rank of the communicator calls it. */
int eofx_ctx_comm_selftest(eofx_ctx *ctx, int *ok);
/* Self-diagnosis of the attached communicator (collective): ranks_seen = all-reduce(sum) of 1.0 per rank, us[i] = mean
 * microseconds (HIP events, `reps` calls after a warm-up) of an all-reduce(sum) of counts[i] elements of dtypes[i]
 * (0 float32, 1 float64, 2 int32) -- the collectives of one sharded fit (n x l float32 per pass, l x l
 * float64 Gram matrices, one int32 vote).  bench.py --gpus N prints both on its JSON line (`comm.ranks_seen`). */
int eofx_ctx_comm_probe(eofx_ctx *ctx, int ncases, const int64_t *counts, const int *dtypes, int reps, double *ranks_seen,
                        double *us);
/* eofx_fit_f32 on this rank's slice [n x P_local] of a field with P_total features (arguments as there; omega: the global
 * sketch [n x (k + n_oversamples)], identical on every rank; needs n < P_total, i.e. the sketch on the sample side).
 * total_variance is the global one; mean / std / valid_feature / V are those of the slice.  Returns 0, or 1 when the
 * fused path is not available on SOME rank (NaN fields, shapes outside eofx_fit_first_f32's range -- the ranks agree on
 * this by a vote): nothing is built then and the caller takes the panel-level route (xeofs_amd/sharded.py).
 * Round 6: with layout mode 3 selected, a field whose NaNs are all-NaN grid points (a land / sea mask, sanitizer.py:80-126)
 * stays on this entry -- every slice keeps its masked features as zero columns, the ranks sum their valid-feature counts
 * (the decomposition needs n < that sum) and V comes back with P_local rows, zeros at the masked features; and modes beyond
 * the numerical rank keep both factors orthonormal as in eofx_fit_f32 (the feature-side one through its all-reduced Gram). */
int eofx_fit_sharded_f32(eofx_ctx *ctx, const float *X, int64_t n, int64_t P_local, int64_t P_total, int center,
                         int standardize, const double *feat_weights, int k, int n_oversamples, int n_iter,
                         const float *omega, int64_t omega_rows, int flip, eofx_mat **out, double *mean, double *std,
                         uint8_t *valid_feature, double *total_variance, float *U, float *s, float *V);

/* ---- the other two sharded decompositions (round 6: the 8-GPU forms of BASELINE configs 3 and 5) -------------------------
 * All-reduce of a small HOST vector of doubles over the context's communicator, in stream order: the global facts a sharded
 * preprocess needs between engine calls (feature counts per rank, the Sanitizer's sample votes -- sanitizer.py:58-126 --,
 * variances), so that a sharded model issues EVERY collective through the engine.  op: 0 sum, 1 max, 2 min.           */
int eofx_ctx_comm_allreduce_f64(eofx_ctx *ctx, double *host_buf, int64_t count, int op);
/* eofx_crosscov_rsvd_f32 (cross/cpcca.py:168-225, 991-1015) with BOTH fields sharded along their own feature axes: x / y
 * are this rank's slices [n x p1_local], [n x p2_local] of fields with p1_total / p2_total features over all ranks, the
 * slice's first feature at p1_offset / p2_offset of the global (valid-feature) axis.  C = X^T Y is never formed; both of
 * its sides are sharded: every sample-side panel (Y Z, X W) is all-reduced (n x L float32), every L x L float64 Gram
 * matrix of a feature-side panel too, and with tsc != NULL the two n x n sample-space Gram matrices once (the power
 * iterations then run replicated in sample space without touching the fields or the links).
 *   omega   host, THIS RANK'S ROWS of the global sketch [min(p1_total, p2_total) x (k + n_oversamples)] (one draw, the
 *           same on every rank): the rows of the features of the narrower field's slice, [p_local x (k + n_oversamples)]
 *           (for a masked in-place slice: its physical rows, zero rows at the masked features)
 * outputs: Q1 / Q2 = this rank's rows [p1_local x k] / [p2_local x k]; s, scores, norms, tsc replicated (bit-identical
 * on every rank).  At world size 1 the call reproduces eofx_crosscov_rsvd_f32 bit for bit.                           */
int eofx_crosscov_rsvd_sharded_f32(eofx_ctx *ctx, const eofx_mat *x, const eofx_mat *y, int64_t p1_total,
                                   int64_t p1_offset, int64_t p2_total, int64_t p2_offset, int k, int n_oversamples,
                                   int n_iter, const float *omega, int flip, float *Q1, float *s, float *Q2,
                                   float *scores1, float *scores2, float *norm1, float *norm2, double *tsc);
/* eofx_rsvd_c64 / eofx_rsvd_hilbert_c64 (decomposer.py:149-160; single/eof.py:433-447,546-555) on this rank's slice of
 * the feature axis of a complex field / of the analytic signal of a real field with p_total (valid) features over all
 * ranks, n < p_total.  The block-Krylov recurrence lives on the replicated sample side; per product Z Y one all-reduce of
 * the n x LP float32 panel, per factorised feature-side panel one of LP x LP float64, two small ones for the sign rule.
 * The Hilbert operator Hc (n x n, resident on every rank) acts on the replicated sample-side panel, so the operator
 * route -- every pass streams the REAL slice once, the imaginary part is never written -- shards without further
 * exchange.  omega: host [n x (k + n_oversamples)], identical on every rank.  U [n x k], s replicated; V = this rank's
 * rows [p_local x k] (for a masked in-place slice: its physical rows, zeros at masked features).                    */
int eofx_rsvd_sharded_c64(eofx_ctx *ctx, const eofx_mat *A, const eofx_mat *B, int64_t p_total, int k,
                          int n_oversamples, int n_iter, const float *omega, int flip_signs, float *U, float *s,
                          float *V);
int eofx_rsvd_hilbert_sharded_c64(eofx_ctx *ctx, const eofx_mat *A, int64_t p_total, int padding,
                                  double decay_factor, int k, int n_oversamples, int n_iter, const float *omega,
                                  int flip_signs, float *U, float *s, float *V);

/* power iterations the last eofx_rsvd_c64 on this context made (n_iter < 0 there = iterate until the Ritz values stand
 * still: the reference's complex branch, scipy svds(solver="lobpcg"), converges to a tolerance -- decomposer.py:149-160) */
int eofx_ctx_last_iterations(const eofx_ctx *ctx, int *iterations);

/* scores = X V (eof.py:129).  V host|device [p x k]; out host|device [n x k].   */
int eofx_project_f32(eofx_ctx *ctx, const eofx_mat *m, const float *V, int k, float *out);
/* Xhat = S V^T (eof.py:151-153).  S [n x k], V [p x k] -> out [n x p] host|device. */
int eofx_reconstruct_f32(eofx_ctx *ctx, const float *S, const float *V, int64_t n, int64_t p,
                         int k, float *out);

/* ---- cross-covariance path (MCA) ---------------------------------------
 * Matrix-free rSVD of C = X^T Y/(n-1) (cpcca.py:1007-1015) -- C is never formed.
 *   omega   host, (min(p1,p2) x (k+n_oversamples)) as above for C's shape (p1 x p2)
 * outputs host|device: Q1 [p1 x k], s [k], Q2 [p2 x k], scores1/2 [n x k],
 * norm1/2 [k] (cpcca.py:204-208); tsc = sum |C|^2 (cpcca.py:991-1000), host.   */
int eofx_crosscov_rsvd_f32(eofx_ctx *ctx, const eofx_mat *x, const eofx_mat *y, int k,
                           int n_oversamples, int n_iter, const float *omega, int flip, float *Q1,
                           float *s, float *Q2, float *scores1, float *scores2, float *norm1,
                           float *norm2, double *tsc);
/* The same call with the sketch handed over by a callback that the engine invokes when it first needs the matrix --
 * with tsc != NULL and both fields wider than long the two sample-space Gram matrices (which the total squared
 * covariance needs, cpcca.py:991-1000, and through which the power iterations then run in sample space) are queued
 * first, so a host generator (decomposer.py:142-145 passes random_state to scikit-learn, whose legacy stream takes
 * ~5 ms for a 129 600 x 30 sketch) runs beside them.  omega_fn returns the matrix (host, layout as above; it must stay
 * valid until the call returns) or NULL on failure; it is called at most once, on the calling thread.   */
typedef const float *(*eofx_sketch_fn)(void *user);
int eofx_crosscov_rsvd_lazy_f32(eofx_ctx *ctx, const eofx_mat *x, const eofx_mat *y, int k, int n_oversamples,
                                int n_iter, eofx_sketch_fn omega_fn, void *omega_user, int flip, float *Q1, float *s,
                                float *Q2, float *scores1, float *scores2, float *norm1, float *norm2, double *tsc);

/* ---- panel-level steps (device pointers only) ---------------------------
 * Building blocks of eofx_rsvd_f32, exported so a host shell can insert
 * collectives between them when the feature axis is sharded over GPUs
 * (one RCCL all-reduce of the n x L panel per pass).
 * A panel is a row-major float32 [rows_pad x L] device buffer, L a multiple of
 * 32, rows_pad = the matrix's n_pad or p_pad; pad rows/columns hold zeros.    */
int eofx_panel_tmul_f32(eofx_ctx *ctx, const eofx_mat *m, const float *Zn, float *Yp, int L,
                        int precision); /* Yp[p_pad x L] = X^T Zn[n_pad x L]; EOFX_PREC_* */
int eofx_panel_mul_f32(eofx_ctx *ctx, const eofx_mat *m, const float *Yp, float *Wn, int L,
                       int precision);
/* G[L x L] (device, float64) = P^T P, accumulated in float64 with a fixed tree. */
int eofx_panel_gram_f64(eofx_ctx *ctx, const float *P, int64_t rows_pad, int L, double *G);
/* Cholesky-QR step from a (possibly all-reduced) Gram matrix: out = P R^-1 with
 * G = R^T R restricted to the leading l x l block; rank-deficient columns -> 0. */
int eofx_panel_cholqr_f32(eofx_ctx *ctx, const float *P, int64_t rows_pad, int L, int l,
                          const double *G, float *out);
/* Rinv[L x L] (device float64) = R^-1 of the Cholesky factor G = R^T R, leading l x l block (zero elsewhere;
 * rank-deficient columns -> 0): the factor eofx_panel_cholqr_f32 applies, on its own -- the second step of
 * CholeskyQR2 is applied on the small side and folded into the final rotation (rsvd_core), not to the tall panel. */
int eofx_panel_rinv_f64(eofx_ctx *ctx, const double *G, int L, int l, double *Rinv);
/* out[rows_pad x Lo] = P[rows_pad x L] * M[L x Lo]  (M device float64 row-major). */
int eofx_panel_matmul_f32(eofx_ctx *ctx, const float *P, int64_t rows_pad, int L,
                          const double *M, int Lo, float *out);
/* per-column max and min over the first `rows` rows: mx[L], mn[L] device float32. */
int eofx_panel_colminmax_f32(eofx_ctx *ctx, const float *P, int64_t rows, int L, float *mx,
                             float *mn);
/* dst[rows x k] (host|device, dense) = P[:, :k] * sign[k] (host doubles or NULL). */
int eofx_panel_export_f32(eofx_ctx *ctx, const float *P, int64_t rows, int L, int k,
                          const double *sign, float *dst);
/* P[rows_pad x L] (device) <- src[rows x l] (host|device dense), zero padded.   */
int eofx_panel_import_f32(eofx_ctx *ctx, const float *src, int64_t rows, int l, float *P,
                          int64_t rows_pad, int L);

/* ---- complex / Hilbert path (ComplexEOF, HilbertEOF: single/eof.py:243-560) ----------
 * A complex (sample x feature) matrix is held as two resident real matrices (Re, Im).
 *
 * eofx_hilbert_f32: analytic signal along the sample axis of every feature of `a`
 * (utils/hilbert_transform.py:40-72): optional exponential padding to 3n around a per-feature
 * linear fit (:75-114, padding != 0, decay_factor), batched FFT (hipFFT), zero negative / double
 * positive frequencies, inverse FFT, middle n samples, minus the per-feature mean.
 * out_imag receives Im; out_real (may be NULL) receives Re = a - mean(a) (equal to `a` itself
 * when `a` is already centred).                                                              */
int eofx_hilbert_f32(eofx_ctx *ctx, const eofx_mat *a, int padding, double decay_factor,
                     eofx_mat **out_imag, eofx_mat **out_real);
/* sum of squares of the resident matrix (float64, fixed reduction tree).                     */
int eofx_mat_sumsq_f64(eofx_ctx *ctx, const eofx_mat *m, double *out);
/* Euclidean row norms (float64, host|device out[rows]) of a panel, and the per-feature norms
 * sqrt(sum_t X[t,j]^2) of a resident matrix (out[p]): the standard deviations the correlation patterns of
 * the cross models divide by (utils/optional/statistics.py:50-54, cross/cpcca.py:642-845).            */
int eofx_panel_rownorm_f64(eofx_ctx *ctx, const float *P, int64_t rows, int L, double *out);
int eofx_mat_feature_norms_f64(eofx_ctx *ctx, const eofx_mat *m, double *out);
/* Euclidean norm of every SAMPLE (row) of the resident matrix, out[n]; an in-place matrix is read through its Scaler map
 * (no layout is built).  The bootstrap members' total variance is a count-weighted sum of their squares.      */
int eofx_mat_sample_norms_f64(eofx_ctx *ctx, const eofx_mat *m, double *out);
/* Bootstrap resampling (validation/bootstrapper.py:78-91): out = rows `rows[0..n_rows)` (host indices
 * into src, drawn with replacement) of the resident matrix, re-centred per feature when `center`
 * (the bootstrap model is `EOF(n_modes)` with its default center=True).  mean (host, [p], may be NULL)
 * receives the per-feature mean that was removed, total_variance the ddof=1 variance sum.          */
int eofx_resample_f32(eofx_ctx *ctx, const eofx_mat *src, const int64_t *rows, int64_t n_rows, int center,
                      eofx_mat **out, double *mean, double *total_variance);
/* A bootstrap member as an operator on the RESIDENT matrix (no resampled copy): rows drawn with replacement and
 * re-centred are X_b = H X, H = G - 1 c^T / n, so the member's two products are X^T (H^T Z) and H (X Y).  This applies H
 * (transpose = 0: out[i] = P[idx[i]] - mean of the gathered rows) or H^T (transpose = 1: out[r] = sum of the draws of row
 * r - c_r * mean-like term) to an n x L sample-side panel; idx [n] = the draw, order [n] = its stable argsort,
 * rowptr [n + 1] = segment starts of `order` per source row (device int64).  Float64 sums in draw order: reproducible. */
int eofx_panel_bootstrap_f32(eofx_ctx *ctx, const float *P_in, int64_t n, int64_t rows_pad, int L, const int64_t *idx,
                             const int64_t *order, const int64_t *rowptr, int transpose, float *P_out);
/* Gram matrix of a resident matrix (float32, device): side 0 = sample space G[n_pad x n_pad] = X X^T,
 * side 1 = feature space G[p_pad x p_pad] = X^T X (rows/columns beyond n / p are zero).  Used for
 * (a) the total squared covariance sum(|X^T Y|^2) = <X X^T, Y Y^T> (cross/cpcca.py:991-1000) when X and
 * Y are sharded over GPUs (with eofx_vec_dot_f64), (b) the exact PCA pre-reduction of the cross models
 * (preprocessing/pca.py:94-123): eigenvectors of the small-side Gram matrix span the PCA subspace.   */
int eofx_mat_gram_f32(eofx_ctx *ctx, const eofx_mat *m, int side, float *G);
/* G = A_a A_b^T (side 0, [n_pad x n_pad]) or A_a^T A_b (side 1, [p_pad x p_pad]) of two resident matrices of one shape:
 * the imaginary blocks of the Hermitian Gram matrix of a complex field a + i b (complex PCA pre-reduction of the complex
 * cross models, xeofs/cross/cpcca.py:1023-1173 with preprocessing/pca.py:94-123). */
int eofx_mat_cross_gram_f32(eofx_ctx *ctx, const eofx_mat *a, const eofx_mat *b, int side, float *G);
int eofx_vec_dot_f64(eofx_ctx *ctx, const float *a, const float *b, int64_t count, double *out);
/* ---- complex randomized SVD (the decomposer's complex branch) -----------------------------------------------
 * Z = A + iB (two resident real matrices of equal shape) ~ U diag(s) V^H.  Replaces
 * scipy.sparse.linalg.svds(X, k, solver="lobpcg") at xeofs/linalg/decomposer.py:149-160 (+ the sign rule,
 * xarray_utils.py:273-301, with numpy's lexicographic complex max / min).  A pass over the data is one launch of the
 * streaming kernel in its two-matrix form; the l x l Hermitian factorisations run on the host in float64.
 * omega: host [min(n,p) x (k + n_oversamples)] REAL start matrix (the Gaussian numpy's RandomState draws; the
 * identity when k + n_oversamples >= min(n, p)); n_iter = -1: scikit-learn's "auto" count (7 or 4 products of the block
 * Krylov recurrence), n_iter = -2 ("converge"): continue the recurrence until every wanted singular value is good to 2e-6
 * and every wanted vector separated from its neighbours by 2 % to |cos| >= 1 - 5e-6, judged from the Ritz values' own
 * convergence history (a host Rayleigh-Ritz solve every third product), at most 20 products -- lobpcg converges to a
 * tolerance, within at most 20 iterations under svds, which matters when wanted modes sit next to a flat noise bulk
 * (profiles/r06_r9_evidence.txt); eofx_ctx_last_iterations reports the count.  U [n x k], V [p x k]:
 * complex64, row-major, interleaved (re, im), host|device; s [k] float32.  k + n_oversamples <= 64
 * (EOFX_ERR_ARG beyond); vectors are defined up to a unit phase per mode, as in the reference.                   */
int eofx_rsvd_c64(eofx_ctx *ctx, const eofx_mat *A, const eofx_mat *B, int k, int n_oversamples, int n_iter,
                  const float *omega, int flip_signs, float *U, float *s, float *V);

/* The same decomposition for the ANALYTIC SIGNAL of a resident real matrix -- Z = A + i H(A), H the Hilbert stage of
 * eofx_hilbert_f32 (padding / decay_factor as there; reference single/eof.py:433-447 -> utils/hilbert_transform.py:10-44
 * -> decomposer.py:149-160) -- without writing the imaginary part.  The stage is linear along the samples, Im = Hc A
 * with one n x n matrix Hc per (n, padding, decay) (built in float64 on the host, resident in both layouts, cached on
 * the context), so Z^H W = A^T (W - i Hc^T W) and Z Y = (I + i Hc)(A Y): every product streams the REAL field once
 * and applies Hc to the sample-side panel -- half the bytes per pass of the two-part form, no second resident field.
 * Arguments, rules, outputs and errors as eofx_rsvd_c64 on (A, eofx_hilbert_f32(A)); n <= 16384 (EOFX_ERR_ARG beyond:
 * use the two calls).  A may be an in-place or masked in-place matrix.                                             */
int eofx_rsvd_hilbert_c64(eofx_ctx *ctx, const eofx_mat *A, int padding, double decay_factor, int k, int n_oversamples,
                          int n_iter, const float *omega, int flip_signs, float *U, float *s, float *V);
/* The operator itself: out [n x n] (host, row-major float32) = Hc with Im = Hc A for the Hilbert stage of eofx_hilbert_f32
 * along the samples (utils/hilbert_transform.py:40-114 is linear in the series; built in float64).  For panel-level
 * drivers that apply it to a sample-side panel themselves (the feature-sharded fallback driver, xeofs_amd/complex_svd.py). */
int eofx_hilbert_operator_f32(eofx_ctx *ctx, int64_t n, int padding, double decay_factor, float *out);
/* sum of squares of the imaginary part eofx_hilbert_f32 would write for `a` (total variance of the analytic signal =
 * (eofx_mat_sumsq_f64(a) + this) / (n - 1); reference single/eof.py:93 on the complex field), computed by the same
 * kernel with its stores switched off; consumes the transposed raw layout of eofx_ctx_set_sample_raw like the stage. */
int eofx_hilbert_sumsq_f64(eofx_ctx *ctx, const eofx_mat *a, int padding, double decay_factor, double *out);

/* One pass of the complex operator Z = A + iB on a [Re | Im] panel of L = 64 or 128 real columns (device pointers):
 *   conj_left = 1: out [p_pad x L] = Z^H W, W [n_pad x L];   conj_left = 0: out [n_pad x L] = Z Y, Y [p_pad x L].
 * One launch of the streaming kernel over both parts in the default precision (the step a feature-sharded driver
 * all-reduces around; the reference's svds(lobpcg) call at xeofs/linalg/decomposer.py:149-160 has no counterpart --
 * it never exposes its matvec).  final_pass selects the context's final-pass precision.                          */
int eofx_cmat_mul_f32(eofx_ctx *ctx, const eofx_mat *A, const eofx_mat *B, int conj_left, const float *P_in, int L,
                      int final_pass, float *P_out);

/* Complex panels are real panels [Re | Im] (Re in columns [0, L/2), Im in [L/2, L)).
 * With P1 = op(A) W and P2 = op(B) W for a complex matrix Z = A + iB (A, B real resident):
 *   conj_left = 1:  out = Z^H W :  out.re = P1.re + P2.im, out.im = P1.im - P2.re
 *   conj_left = 0:  out = Z   W :  out.re = P1.re - P2.im, out.im = P1.im + P2.re          */
int eofx_cpanel_combine_f32(eofx_ctx *ctx, const float *P1, const float *P2, int conj_left,
                            int64_t rows_pad, int L, float *out);
/* row index of the per-column max and min over the first `rows` rows (device int64[L]);
 * used for the lexicographic complex max/min of the sign rule (xarray_utils.py:294-296).     */
int eofx_panel_colargminmax_f32(eofx_ctx *ctx, const float *P, int64_t rows, int L, int64_t *amax,
                                int64_t *amin);

/* ---- rotation of loadings (EOFRotator: linalg/_numpy/_rotation.py:6-187), up to 256 modes -----
 * eofx_panel_row_normalize_f32: Kaiser normalisation out[r,:] = P[r,:] / (||P[r,:]|| + eps).
 * eofx_panel_rot_step_f64: one pass over the normalised loadings X (rows_pad x L):
 *   per row b = x R (float64); mode 0 (one Varimax iteration, _rotation.py:166-176):
 *   G = X^T (b * (b^2 - aux)), aux[j] = alpha * sum_r b_rj^2;  mode 1 (Promax regression terms,
 *   _rotation.py:62-69): G = B^T ((b/aux) |b/aux|^(power-1)), aux[j] = max_r |b_rj|.
 *   R, aux, G are device float64 (L x L, L, L x L); L = 32, 64 (one workgroup per row tile) or 128, 256 (G cut
 *   into column blocks of 64, one workgroup each; aux must be padded with 0 (even modes) / 1 (odd modes)).
 *   modes 2 / 3: the same two steps for COMPLEX loadings held as a [Re (L/2) | Im (L/2)] panel, L >= 64 (the rotation of
 *   ComplexEOF / HilbertEOF models, eof_rotator.py:294-400): R is the real L x L embedding [[Rr, Ri], [-Ri, Rr]] of the
 *   complex rotation matrix (L x L), aux[j] = aux[j + L/2] the per-mode value, |b_j|^2 pairs column j with j + L/2, and G holds
 *   the four real blocks of X^H T = (Xr^T Tr + Xi^T Ti) + i (Xr^T Ti - Xi^T Tr).
 * eofx_cpanel_colabsmax_f32: max over the rows of |column| for the L/2 complex columns of such a panel (device out). */
int eofx_panel_row_normalize_f32(eofx_ctx *ctx, const float *P, int64_t rows_pad, int L, float *out);
int eofx_panel_rot_step_f64(eofx_ctx *ctx, const float *X, int64_t rows_pad, int L, const double *R,
                            const double *aux, int mode, double power, double *G);
int eofx_cpanel_colabsmax_f32(eofx_ctx *ctx, const float *P, int64_t rows, int L, float *out);

/* ---- the sketch matrix ------------------------------------------------------------------
 * out[rows x cols] (host float32) = np.random.RandomState(seed).normal(size=(rows, cols))
 * .astype(float32), bit for bit -- the Gaussian test matrix scikit-learn's randomized_svd draws
 * (MT19937 + numpy's legacy polar method), so `random_state=seed` means what it means in the
 * reference (decomposer.py:141-146).                                                        */
int eofx_sketch_gaussian_f32(uint32_t seed, int64_t rows, int64_t cols, float *out);

/* ---- small host linear algebra used by the drivers ---------------------- */
/* symmetric eigen-decomposition (Householder tridiagonalisation + implicit QL, float64):
 * A[n x n] row-major ->
 * eigenvalues w[n] descending, eigenvectors as columns of Vec[n x n].          */
int eofx_host_eigh_f64(const double *A, int n, double *w, double *Vec);
/* leading `nev` eigenpairs of the complex Hermitian matrix Hr + i Hi (row-major m x m, float64): Householder reduction to a
 * real tridiagonal matrix, QL eigenvalues, inverse iteration, back-transformation (xeofs_amd/csrc/eofx_hosteig.hpp).  The
 * Rayleigh-Ritz step of the block-Krylov complex decomposition (eofx_rsvd_c64; the reference's counterpart is the dense
 * eigen-problem inside scipy's lobpcg, reached from xeofs/linalg/decomposer.py:149-160).
 * w[nev] descending, Xr / Xi [m x nev] row-major with orthonormal columns.     */
int eofx_host_zheigh_top_f64(const double *Hr, const double *Hi, int m, int nev, double *w, double *Xr, double *Xi);

#ifdef __cplusplus
}
#endif
#endif /* EOFX_H */
