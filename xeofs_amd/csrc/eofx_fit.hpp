// eofx_fit.hpp -- the fused fit: the Scaler's column statistics ride on the FIRST pass of the randomized SVD.
//
// xeofs fits in two steps: Scaler.fit (per-feature mean / std, xeofs/preprocessing/scaler.py:69-126) reads the field
// once, then the decomposer (xeofs/linalg/decomposer.py:141-146) streams it 2 n_iter + 2 times.  The first of those
// passes is Y = X'^T Omega with X' = (X - mean) * scale -- linear in the data, so it can run BEFORE the mean is known:
//
//     Y[j, :] = scale_j * ( sum_i (x_ij - c_j) Omega[i, :]  -  (mean_j - c_j) * sum_i Omega[i, :] )
//
// with any provisional shift c_j (here: the mean of nine sampled rows, which keeps x - c small, so the split-fp16
// product loses nothing to cancellation).  atb_f16_fit_kernel computes the first sum on the matrix cores exactly like
// atb_f16_kernel<NB, true>; the per-feature sum of (x - c) comes out of the SAME matrix product -- the sketch panel is
// 64 (32) columns wide and the sketch uses fewer, so a spare column of the B operand is set to ones --, and the sums of
// (x - c)^2 and max |x - c| are taken from the registers the kernel converts anyway; fit_finalize_kernel turns those
// into mean / std / shift / scale (the same formulas as colstats_finalize_kernel) and fit_reduce_kernel applies the
// rank-one correction while it reduces the split-K partials.  A fit then reads the field 2 n_iter + 2 times instead
// of 2 n_iter + 3.
//
// NaN anywhere, a non-finite value or an overflow of the provisional fp16 scaling shows up in the statistics
// (NaN / inf propagate through the sums; the maximum is compared with the fp16 range) and sends the caller back to
// the two-step path, which owns the NaN policies of the Sanitizer.  gfx950 only.
#pragma once
#include "eofx_kernels.hpp"

namespace eofx {

// Provisional shift c_j = mean of nine sampled rows (first, last, seven in between) and an estimate of max |x - c|
// from the same rows.  One thread per four adjacent features (P % 4 == 0, 16-byte aligned rows).  flags bit 0: a NaN or
// an infinity was seen in SOME of the sampled rows of a feature; bit 1: a feature is NaN in ALL nine -- a candidate for
// an all-NaN grid point (land / sea mask).  Its shift is written as NaN, the marker the first pass and
// fit_finalize_kernel read.
__global__ __launch_bounds__(256) void fit_probe_kernel(const float* __restrict__ X, int64_t n, int64_t P, int64_t ld,
                                                         int64_t p_pad, float* __restrict__ cshift,
                                                         unsigned* __restrict__ est, int* __restrict__ flags) {
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  float m = 0.f;
  bool bad = false, cand = false;
  if (c < P) {
    f32x4 v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int64_t r = t == 8 ? n - 1 : (n * t) / 8;
      v[t] = *reinterpret_cast<const f32x4*>(X + (r < n ? r : n - 1) * ld + c);
    }
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
    int nn[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      c0 += v[t];
#pragma unroll
      for (int e = 0; e < 4; ++e) nn[e] += v[t][e] != v[t][e];
    }
    c0 *= (1.f / 9.f);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (nn[e] == 9) {                     // all-NaN candidate: marker
        cand = true;
        c0[e] = NAN;
      } else if (!(fabsf(c0[e]) < INFINITY)) {     // NaN or inf among the samples
        bad = true;
        c0[e] = 0.f;
      }
    *reinterpret_cast<f32x4*>(cshift + c) = c0;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(v[t][e] - c0[e]));
  } else if (c < p_pad) {
    *reinterpret_cast<f32x4*>(cshift + c) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  const int anybad = (__any(bad) ? 1 : 0) | (__any(cand) ? 2 : 0);
  if ((threadIdx.x & 63) == 0) {
    if (m > 0.f) atomicMax(est, __float_as_uint(m));      // inf included: the host checks  (NaN differences are skipped)
    if (anybad) atomicOr(flags, anybad);
  }
}

// ---------------------------------------------------------------------------------
// atb_f16_fit: C = A'^T B with A' = (A - c) * a_scale, A the RAW field [a_rows x a_cols] (lda), plus per-feature
// statistics of A'.  Structure, operand layout and split-fp16 arithmetic are those of atb_f16_kernel<NB, true>
// (eofx_kernels.hpp); what differs:
//   * the map is one subtraction and one exact power-of-two scale (2 instead of 3 VALU per element);
//   * column 32 NB - 1 of the B operand -- a padding column of the sketch panel, zero in B itself -- is staged as ONES for
//     the rows inside the field: C[:, 32 NB - 1] = (FIT_ONE / b_scale) sum_i (x - c) comes out of the matrix product at
//     no cost;
//   * per slab and feature the lane adds its 8 squares to a float32 accumulator and folds max |v| into one register
//     shared by its four features (the maximum only fixes a power-of-two scale: a bound over neighbours will do); every
//     FIT_FLUSH slabs (128 terms: no accumulation error worth the name) the squares move to float64 accumulators that
//     live in LDS -- the kernel has no registers to spare -- by fire-and-forget LDS atomics on addresses only this lane
//     touches (program order, hence deterministic).  The hot loop carries no row test: the K range is padded to a
//     multiple of 32 rows and the rows beyond the field re-read its LAST row (their B rows are zero, ones column
//     included), so the squares hold K - a_rows extra copies of that row's term; fit_finalize_kernel takes them out;
//   * the epilogue adds the two half-waves (rows 2t and 2t+1) and writes the partial statistics of this split:
//     st_sq [split][st_ld] float64, st_max [split][st_ld] float32.
// grid = (M / 512, splits, 1), NB = 1 or 2 (L = 32 NB columns, of which the sketch uses at most L - 1).
// ---------------------------------------------------------------------------------
constexpr int FIT_FLUSH = 16;   // slabs between two flushes of the float32 sums of squares (even)
// The "ones" of the spare column as the fp16 operand sees them: 2^13, whatever the panel's own scale (a literal 1.0
// times the panel scale could leave fp16's range for panels with tiny entries); fit_finalize_kernel divides it out.
constexpr float FIT_ONE = 8192.f;

template <int NB>
__global__ __launch_bounds__(256, 2) void atb_f16_fit_kernel(const float* __restrict__ A, int64_t lda, int a_rows,
                                                              int64_t a_cols, const float* __restrict__ cshift,
                                                              const float* __restrict__ B, int ldb,
                                                              float* __restrict__ C, int ldc, int64_t M, int64_t K,
                                                              int64_t k_per_split, float a_scale,
                                                              const float* __restrict__ b_absmax,
                                                              double* __restrict__ st_sq,
                                                              float* __restrict__ st_max, int64_t st_ld) {
  __shared__ __attribute__((aligned(16))) _Float16 Bs[2][2][2][32 * NB][8];
  __shared__ double Sq[4][256];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * ATB_BM + wave * ATB_WM;
  const int64_t kb = (int64_t)blockIdx.y * k_per_split;
  const int64_t ke = (kb + k_per_split < K) ? kb + k_per_split : K;
  const int nchunks = (int)((ke - kb) / ATB_KC);
  const float b_scale = f16_scale_for(*b_absmax);
  const float out_scale = 1.f / (a_scale * b_scale);   // exact: both are powers of two

  f32x16 acc[4][NB];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][b][r] = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) Sq[j][tid] = 0.0;
  float qa[4] = {0.f, 0.f, 0.f, 0.f};
  float mall = 0.f;        // max |v| over this lane's FOUR features: one register; an upper bound is all the scaling needs

  const bool colok = m0 + 4 * li < a_cols;
  const float* Ap = A + (colok ? m0 + 4 * li : 0);
  f32x4 cs_ = {0.f, 0.f, 0.f, 0.f};
  if (colok) cs_ = *reinterpret_cast<const f32x4*>(cshift + m0 + 4 * li);
  const float asc_ = colok ? a_scale : 0.f;      // 16-byte chunks beyond the field read chunk 0 and count as zeros
  f32x4 ncs_ = -cs_ * asc_;                      // (x - c) a = fma(x, a, -c a): one rounding, one instruction (a = 2^k)
  // All-NaN candidates (shift marker NaN, fit_probe_kernel): the addend +inf keeps a NaN a NaN -- confined to this
  // feature's row of C, its sums and nothing else: the running maximum skips NaN operands -- and turns any FINITE value
  // of the column into +inf, which the maximum does not skip: fit_finalize_kernel then sees an overflow and the caller
  // takes the two-step path with its isolated-NaN policy.  No instruction in the hot loop changes.
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (cs_[e] != cs_[e]) ncs_[e] = INFINITY;
  float m1 = -1.f;   // opaque to the optimiser: x - (float)h stays ONE v_fma_mix_f32 instead of a conversion and a subtraction
  asm volatile("" : "+v"(m1));
  // B staging as in atb_f16_kernel: item = (column pair cp, parity lh, row pair tp), two 8-byte loads, one packed 4-byte
  // LDS store per column and plane, bank-conflict free; three registers of per-thread state
  constexpr int CP = 16 * NB;
  const bool b_odd = (tid >> 4) & 1;
  int b_off, b_w0, b_w1, b_r0;
  bool b_loader, b_ones;
  {
    const int tp_ = tid & 3, lh_ = (tid >> 4) & 1;
    const int cp_ = ((tid >> 2) & 3) | ((tid >> 5) << 2);
    b_loader = cp_ < CP;
    b_ones = cp_ == CP - 1;            // this loader stages columns 32 NB - 2 and 32 NB - 1: the latter is the ones column
    b_r0 = (int)kb + 4 * tp_ + lh_;    // its rows of slab 0: b_r0 and b_r0 + 2
    b_off = (4 * tp_ + lh_) * ldb + 2 * cp_;
    const int w_ = ((lh_ * 32 * NB + 2 * cp_) * 8 + 2 * tp_) * 2;
    b_w0 = w_ + 16 * lh_;
    b_w1 = w_ + 16 * (1 - lh_);
  }
  char* const Bsb = reinterpret_cast<char*>(&Bs[0][0][0][0][0]);
  constexpr int B_PLANE = 2 * 32 * NB * 8 * 2, B_BUF = 2 * B_PLANE;

  f32x4 a0[8], a1[8];
  f32x4 bn = {0.f, 0.f, 0.f, 0.f};      // {row r0: col 2cp, 2cp+1; row r0 + 2: col 2cp, 2cp+1}
#define EOFX_FIT_LOAD_B(chunk)                                                                   \
  do {                                                                                           \
    if (b_loader) {                                                                              \
      const float* bp_ = B + (kb + (int64_t)(chunk) * ATB_KC) * ldb + b_off;                       \
      const f32x2 lo_ = *reinterpret_cast<const f32x2*>(bp_);                                    \
      const f32x2 hi_ = *reinterpret_cast<const f32x2*>(bp_ + 2 * ldb);                          \
      bn = f32x4{lo_[0], lo_[1], hi_[0], hi_[1]};                                                \
      if (b_ones) {                                                                              \
        const int row_ = b_r0 + (chunk) * ATB_KC;                                                \
        bn[1] = row_ < a_rows ? FIT_ONE / b_scale : 0.f;                                         \
        bn[3] = row_ + 2 < a_rows ? FIT_ONE / b_scale : 0.f;                                     \
      }                                                                                          \
    }                                                                                            \
  } while (0)
#define EOFX_FIT_LOAD_A(areg, chunk)                                                             \
  do {                                                                                           \
    const int r0_ = (int)kb + (chunk) * ATB_KC + lh;                                             \
    if ((int)kb + (chunk) * ATB_KC + ATB_KC <= a_rows) {   /* whole slab inside the field */      \
      const float* pa_ = Ap + (int64_t)r0_ * lda;                                                \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) areg[i] =                                    \
          __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pa_ + (int64_t)(2 * i) * lda)); \
    } else {                                                                                     \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                            \
        const int r_ = r0_ + 2 * i < a_rows ? r0_ + 2 * i : a_rows - 1;                          \
        areg[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Ap + (int64_t)r_ * lda)); \
      }                                                                                          \
    }                                                                                            \
  } while (0)
#define EOFX_FIT_STORE_B(buf)                                                                    \
  do {                                                                                           \
    if (b_loader) _Pragma("unroll") for (int e2 = 0; e2 < 2; ++e2) {                             \
      const bool sec_ = (e2 != 0) != b_odd;                                                      \
      const float v0_ = (sec_ ? bn[1] : bn[0]) * b_scale, v1_ = (sec_ ? bn[3] : bn[2]) * b_scale; \
      const fp16x2_t h_ = cvt_pk_rn(v0_, v1_);                                  \
      fp16x2_t l_;                                                                               \
      l_[0] = (__fp16)__builtin_fmaf((float)h_[0], m1, v0_);                                     \
      l_[1] = (__fp16)__builtin_fmaf((float)h_[1], m1, v1_);                                     \
      char* w_ = Bsb + (buf) * B_BUF + (e2 ? b_w1 : b_w0);                                       \
      *reinterpret_cast<unsigned*>(w_) = __builtin_bit_cast(unsigned, h_);                       \
      *reinterpret_cast<unsigned*>(w_ + B_PLANE) = __builtin_bit_cast(unsigned, l_);             \
    }                                                                                            \
  } while (0)
  /* Left to itself the SLP vectoriser packs the fmas below (across t, across j) into v_pk_fma_f32: slower beside MFMAs,
     and the register pairs push the kernel into scratch.  The empty asm statements make each value opaque, which keeps
     the scalar forms without pinning the instruction order (spelled-out asm made the scheduler pad with s_nop and bunch
     the MFMAs). */
#define EOFX_FIT_CONVERT_J(areg, j, af_)                                                         \
  do {                                                                                           \
    f32x8 x_;                                                                                    \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                              \
      float xv_ = __builtin_fmaf(areg[t][j], asc_, ncs_[j]);                                     \
      asm("" : "+v"(xv_));                                                                       \
      x_[t] = xv_;                                                                               \
    }                                                                                            \
    {                                                                                            \
      float q_ = qa[j], mm_ = mall;                                                              \
      _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                            \
        q_ = __builtin_fmaf(x_[t], x_[t], q_);                                                   \
        mm_ = __builtin_fmaxf(mm_, __builtin_fabsf(x_[t]));     /* NaN operands are skipped */   \
      }                                                                                          \
      asm("" : "+v"(q_), "+v"(mm_));                                                             \
      qa[j] = q_;                                                                                \
      mall = mm_;                                                                                \
    }                                                                                            \
    split_f16_mix(x_, m1, af_);                                                                  \
  } while (0)
#define EOFX_FIT_MFMA_J(j, af_)                                                                  \
  _Pragma("unroll") for (int q = 0; q < NB; ++q) {                                               \
    acc[j][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_[1], bf_[0][q], acc[j][q], 0, 0, 0);   \
    acc[j][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_[0], bf_[1][q], acc[j][q], 0, 0, 0);   \
    acc[j][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_[0], bf_[0][q], acc[j][q], 0, 0, 0);   \
  }
  // as in atb_f16_kernel: the slab's registers are refilled right after its last conversion
#define EOFX_FIT_COMPUTE(areg, buf, refill)                                                      \
  do {                                                                                           \
    f16x8 bf_[2][NB];                                                                            \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int q = 0; q < NB; ++q) \
        bf_[s][q] = *reinterpret_cast<const f16x8*>(&Bs[buf][s][lh][32 * q + li][0]);            \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                              \
      f16x8 af_[2];                                                                              \
      EOFX_FIT_CONVERT_J(areg, j, af_);                                                          \
      EOFX_FIT_MFMA_J(j, af_)                                                                    \
    }                                                                                            \
    f16x8 al_[2];                                                                                \
    EOFX_FIT_CONVERT_J(areg, 3, al_);                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    EOFX_FIT_LOAD_A(areg, refill);                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    EOFX_FIT_MFMA_J(3, al_)                                                                      \
  } while (0)
#define EOFX_FIT_FLUSH()                                                                         \
  do {                                                                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                              \
      (void)__hip_atomic_fetch_add(&Sq[j][tid], (double)qa[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
      qa[j] = 0.f;                                                                               \
    }                                                                                            \
  } while (0)

  // The statistics see every slab once: the two refills past the end of the K range (harmless re-reads for the
  // product) are made of the LAST pair again but never converted.
  if (nchunks > 0) {
    EOFX_FIT_LOAD_B(0);
    EOFX_FIT_LOAD_A(a0, 0);
    EOFX_FIT_LOAD_A(a1, 1);
    EOFX_FIT_STORE_B(0);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
      const int c2 = (c + 2 < nchunks) ? c + 2 : c;
      EOFX_FIT_LOAD_B(c + 1);
      EOFX_FIT_COMPUTE(a0, 0, c2);
      EOFX_FIT_STORE_B(1);
      __syncthreads();
      EOFX_FIT_LOAD_B(c2);
      EOFX_FIT_COMPUTE(a1, 1, c2 + 1);
      EOFX_FIT_STORE_B(0);
      if ((c & (FIT_FLUSH - 2)) == FIT_FLUSH - 2) EOFX_FIT_FLUSH();
      __syncthreads();
    }
  }
  EOFX_FIT_FLUSH();
#undef EOFX_FIT_LOAD_B
#undef EOFX_FIT_LOAD_A
#undef EOFX_FIT_COMPUTE
#undef EOFX_FIT_CONVERT_J
#undef EOFX_FIT_MFMA_J
#undef EOFX_FIT_STORE_B
#undef EOFX_FIT_FLUSH

  float* Cs = C + (int64_t)blockIdx.y * M * ldc;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ii = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int64_t m = m0 + 4 * ii + j;
        Cs[m * ldc + 32 * q + li] = acc[j][q][r] * out_scale;
      }
  // statistics of this split: rows 2t (lh = 0) + rows 2t + 1 (lh = 1), in that order
  {
    f64x4 q4;
    f32x4 m4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double q_ = Sq[j][tid];
      const double qo_ = __shfl_xor(q_, 32);
      const float mo_ = __shfl_xor(mall, 32);
      q4[j] = q_ + qo_;      // only the lh == 0 lanes store: (rows 2t) + (rows 2t + 1)
      m4[j] = fmaxf(mall, mo_);
    }
    if (lh == 0 && colok) {
      const int64_t o = (int64_t)blockIdx.y * st_ld + m0 + 4 * li;
      *reinterpret_cast<f64x4*>(st_sq + o) = q4;
      *reinterpret_cast<f32x4*>(st_max + o) = m4;
    }
  }
}

// Statistics of the provisionally shifted values -> the Scaler's state.  One thread per feature c < p_pad.
//   S1 = sum (x - cshift) over all n samples: column L - 1 of the split-K partials of the first pass (the ones column,
//        times FIT_ONE / b_scale), summed over the splits in fixed order;
//   S2 = sum v^2, M = max |v| with v = (x - cshift) * a_scale (split partials in fixed order); the kernel's K range
//        is padded with `extra` = K - n re-reads of the last row, whose squares are subtracted here.
//   mean = cshift + S1 / n;  M2 = S2 / a^2 - S1^2 / n;  std = sqrt(M2 / n) clipped at eps        (scaler.py:101-108)
//   shift = center ? mean : 0;  scale = (standardize ? 1 / std : 1) * weight                   (scaler.py:128-154)
// flags: bit 0 a sum is not finite (NaN / inf in the data), bit 1 the provisional fp16 scaling overflowed (or a masked
// candidate held a finite value), bit 2 all-NaN grid points are present (count 0, zero map: the masked in-place layout),
// bit 3 (standardize only) a feature's standard deviation is below 2^-14 of the field's largest |x - c|: the first pass
// splits the RAW values against ONE scale, so such a feature reaches the matrix cores -- and its sum S1 -- with fewer than
// 16 of its 22 bits (below 2^-33: with none), which 1 / std then magnifies (mixed-unit fields: pressure in Pa next to
// specific humidity; tools/scale_probe.py: 7e-6 on the values at 8 orders of magnitude between features, nonsense at 16).
// The two-step path maps every feature to unit variance BEFORE the split and is exact there: the fit goes back to it.
// Also written: the float triples of the in-place view (aff_pack_kernel's layout), dcorr = shift - cshift for
// fit_reduce_kernel, and max |(x - shift) * scale| (an upper bound within a factor of two) into *absmax.
__global__ __launch_bounds__(256) void fit_finalize_kernel(
    const float* __restrict__ part, int64_t part_rows, int L, const double* __restrict__ st_sq,
    const float* __restrict__ st_max, int64_t st_ld, int splits, const float* __restrict__ X, int64_t ld, int64_t n,
    int64_t extra, int64_t P, int64_t p_pad, const float* __restrict__ cshift, float a_scale, int center, int standardize,
    const double* __restrict__ weights, double eps, const float* __restrict__ b_absmax, int* __restrict__ cnt,
    double* __restrict__ mean, double* __restrict__ stdv, double* __restrict__ shift, double* __restrict__ scale,
    double* __restrict__ m2, float* __restrict__ aff, double* __restrict__ dcorr, unsigned* __restrict__ absmax,
    int* __restrict__ flags) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float amax = 0.f;
  int fl = 0;
  if (c < P) {
    const double unit = (double)f16_scale_for(*b_absmax) / (double)FIT_ONE;   // exact: powers of two
    double S1 = 0.0, q = 0.0;
    float mx = 0.f;
    for (int sp = 0; sp < splits; ++sp) {
      S1 += (double)part[((int64_t)sp * part_rows + c) * L + (L - 1)];
      q += st_sq[(int64_t)sp * st_ld + c];
      mx = fmaxf(mx, st_max[(int64_t)sp * st_ld + c]);
    }
    S1 *= unit;
    const float cs = cshift[c];
    if (cs != cs) {       // an all-NaN grid point (no finite value: that would have set mx = inf): invalid feature, zero column
      if (!(mx < 60000.f)) fl |= 2;
      fl |= 4;
      const double w = weights ? weights[c] : 1.0;
      cnt[c] = 0;
      mean[c] = NAN;
      stdv[c] = NAN;
      shift[c] = center ? NAN : 0.0;      // as colstats_finalize_kernel leaves a feature without data
      scale[c] = (standardize ? NAN : 1.0) * w;
      m2[c] = 0.0;
      dcorr[c] = 0.0;
      aff[c] = 0.f;
      aff[p_pad + c] = 0.f;
      aff[2 * p_pad + c] = 0.f;
    } else {
    if (extra > 0) {
      const float v = __builtin_fmaf(X[(n - 1) * ld + c], a_scale, -cs * a_scale);    // the kernel's own expression
      q -= (double)extra * ((double)v * (double)v);
    }
    if (!(fabs(S1) < INFINITY) || !(fabs(q) < INFINITY)) fl |= 1;
    if (!(mx < 60000.f)) fl |= 2;
    const double ia = 1.0 / (double)a_scale;          // exact: a power of two
    const double nn = (double)n;
    const double mu = (double)cs + S1 / nn;
    double M2 = q * ia * ia - S1 * S1 / nn;
    if (!(M2 > 0.0)) M2 = 0.0;
    double sd = sqrt(M2 / nn);
    if (standardize && sd > eps && sd * (double)a_scale < 0.015625) fl |= 8;
    if (sd < eps) sd = eps;
    const double w = weights ? weights[c] : 1.0;
    const double sh = center ? mu : 0.0, sc = (standardize ? 1.0 / sd : 1.0) * w;
    cnt[c] = (int)n;
    mean[c] = mu;
    stdv[c] = sd;
    shift[c] = sh;
    scale[c] = sc;
    m2[c] = M2;
    dcorr[c] = sh - (double)cs;
    float hi, lo;
    aff_split(sh, hi, lo);
    aff[c] = hi;
    aff[p_pad + c] = lo;
    aff[2 * p_pad + c] = (float)sc;
    amax = (float)(((double)mx * ia + fabs(sh - (double)cs)) * fabs(sc) * 1.000001);
    }
  } else if (c < p_pad) {
    aff[c] = 0.f;
    aff[p_pad + c] = 0.f;
    aff[2 * p_pad + c] = 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  const int anyfl = (__any(fl & 1) ? 1 : 0) | (__any(fl & 2) ? 2 : 0) | (__any(fl & 4) ? 4 : 0) | (__any(fl & 8) ? 8 : 0);
  if ((threadIdx.x & 63) == 0) {
    if (amax > 0.f && amax < INFINITY) atomicMax(absmax, __float_as_uint(amax));
    if (anyfl) atomicOr(flags, anyfl);
  }
}

// column sums of the first `rows` rows of a [rows_pad x L] panel in float64: partials over row ranges, then one
// workgroup adds them in a fixed order.  (sum_i Omega[i, :] of the rank-one correction.)
__global__ __launch_bounds__(256) void panel_colsum_part_kernel(const float* __restrict__ P, int64_t rows, int L,
                                                                 double* __restrict__ part) {
  __shared__ double red[4][64];
  const int col = threadIdx.x & 63, rq = threadIdx.x >> 6;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  for (int cb = 0; cb < L; cb += 64) {
    double s = 0.0;
    if (cb + col < L)
      for (int64_t r = r0 + rq; r < r1; r += 4) s += (double)P[r * L + cb + col];
    red[rq][col] = s;
    __syncthreads();
    if (rq == 0 && cb + col < L) part[(int64_t)blockIdx.x * L + cb + col] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void panel_colsum_final_kernel(const double* __restrict__ part, int nparts, int L,
                                                                  double* __restrict__ out) {
  for (int c = threadIdx.x; c < L; c += 256) {
    double s = 0.0;
    for (int p = 0; p < nparts; ++p) s += part[(int64_t)p * L + c];
    out[c] = s;
  }
}

// Y[j, c] = scale_j * (sum_splits part[s][j, c] - dcorr_j * wbar[c]) for j < P and c < l, 0 for the padding rows and
// columns (the ones column of the first pass among them) and for features without data (cnt[j] == 0); float64
// arithmetic, fixed order.  part may alias out when
// splits == 1.  amax_out (may be null): max |Y| by atomicMax on the float bits (order independent).
__global__ __launch_bounds__(256) void fit_reduce_kernel(const float* part, float* out, int64_t rows_pad, int L, int l,
                                                          int splits, int64_t P, const double* __restrict__ dcorr,
                                                          const double* __restrict__ scale,
                                                          const double* __restrict__ wbar,
                                                          unsigned* __restrict__ amax_out, const int* __restrict__ cnt = nullptr) {
  // thread (row group, quad): quad = 4 adjacent columns, fixed for the thread, so its four wbar values are loaded once;
  // rows stride by the number of row slots of the grid.  L / 4 is 8 or 16: 32 or 16 rows per 256-thread workgroup.
  const int l4 = L / 4;
  const int quad = threadIdx.x % l4, rslot = threadIdx.x / l4, rper = 256 / l4;
  const int c = 4 * quad;
  const int64_t count4 = rows_pad * l4;
  double w[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) w[e] = c + e < l ? wbar[c + e] : 0.0;
  float m = 0.f;
  for (int64_t j = (int64_t)blockIdx.x * rper + rslot; j < rows_pad; j += (int64_t)gridDim.x * rper) {
    const int64_t i = j * l4 + quad;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (j < P && !(cnt && cnt[j] == 0)) {      // (the partial sums of an all-NaN grid point are NaN: its row of Y is zero)
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      for (int s = 0; s < splits; ++s) {
        const f32x4 v = reinterpret_cast<const f32x4*>(part)[(int64_t)s * count4 + i];
        s0 += v[0];
        s1 += v[1];
        s2 += v[2];
        s3 += v[3];
      }
      const double d = dcorr[j], sc = scale[j];
      o[0] = c < l ? (float)((s0 - d * w[0]) * sc) : 0.f;
      o[1] = c + 1 < l ? (float)((s1 - d * w[1]) * sc) : 0.f;
      o[2] = c + 2 < l ? (float)((s2 - d * w[2]) * sc) : 0.f;
      o[3] = c + 3 < l ? (float)((s3 - d * w[3]) * sc) : 0.f;
      m = fmaxf(m, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
    }
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
  if (amax_out) amax_commit(m, amax_out);
}

// ---------------------------------------------------------------------------------
// Bootstrap member as an operator (validation/bootstrapper.py:78-91): rows drawn with replacement and re-centred are
// X_b = H X with H = G - 1 c^T / n (G the row selector of the draw idx, c its counts), so the member's products are the
// original matrix' products with the small sample-side panel transformed:
//   H   W : out[i] = W[idx[i]] - (sum_i W[idx[i]]) / n                         (bst_gather_kernel, then the column sums)
//   H^T Z : out[r] = sum_{i : idx[i] = r} Z[i] - c_r (sum_i Z[i]) / n          (bst_segsum_kernel over the sorted draw)
// Float64 sums in a fixed order (segments are walked in draw order): a member is reproducible bit for bit.  These two
// kernels replace a dense n x n library GEMM per pass.
// ---------------------------------------------------------------------------------
// one thread per (row, 4 columns): out[i, :] = P[idx[i], :] for i < n, 0 for the padding rows
__global__ __launch_bounds__(256) void bst_gather_kernel(const float* __restrict__ P, const int64_t* __restrict__ idx,
                                                          int64_t n, int64_t rows_pad, int L, float* __restrict__ out) {
  const int l4 = L / 4;
  const int64_t total = rows_pad * l4;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t i = t / l4;
    const int q = (int)(t - i * l4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < n) v = reinterpret_cast<const f32x4*>(P)[idx[i] * l4 + q];
    reinterpret_cast<f32x4*>(out)[t] = v;
  }
}
// out[r, :] = sum over the draws i in [rowptr[r], rowptr[r+1]) of Z[order[i], :]  (float64, draw order) for r < n
__global__ __launch_bounds__(256) void bst_segsum_kernel(const float* __restrict__ Z, const int64_t* __restrict__ order,
                                                          const int64_t* __restrict__ rowptr, int64_t n, int64_t rows_pad,
                                                          int L, float* __restrict__ out) {
  const int l4 = L / 4;
  const int64_t total = rows_pad * l4;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t r = t / l4;
    const int q = (int)(t - r * l4);
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (r < n)
      for (int64_t i = rowptr[r]; i < rowptr[r + 1]; ++i) {
        const f32x4 v = reinterpret_cast<const f32x4*>(Z)[order[i] * l4 + q];
        s0 += v[0];
        s1 += v[1];
        s2 += v[2];
        s3 += v[3];
      }
    reinterpret_cast<f32x4*>(out)[t] = f32x4{(float)s0, (float)s1, (float)s2, (float)s3};
  }
}
// out[r, :] -= w_r * colsum[:] / n for r < n, with w_r = rowptr[r+1] - rowptr[r] (the draw count c_r) or 1 (rowptr null)
__global__ __launch_bounds__(256) void bst_rankone_kernel(float* __restrict__ out, const double* __restrict__ colsum,
                                                           const int64_t* __restrict__ rowptr, int64_t n, int L) {
  const int l4 = L / 4;
  const int64_t total = n * l4;
  const double inv = 1.0 / (double)n;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t r = t / l4;
    const int c = (int)(t - r * l4) * 4;
    const double w = rowptr ? (double)(rowptr[r + 1] - rowptr[r]) * inv : inv;
    f32x4 v = reinterpret_cast<f32x4*>(out)[t];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)((double)v[e] - w * colsum[c + e]);
    reinterpret_cast<f32x4*>(out)[t] = v;
  }
}

}  // namespace eofx
