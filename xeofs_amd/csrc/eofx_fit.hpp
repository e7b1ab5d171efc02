// eofx_fit.hpp -- the fused fit: the Scaler's column statistics ride on the FIRST pass of the randomized SVD.
//
// xeofs fits in two steps: Scaler.fit (per-feature mean / std, xeofs/preprocessing/scaler.py:69-126) reads the field
// once, then the decomposer (xeofs/linalg/decomposer.py:141-146) streams it 2 n_iter + 2 times.  The first of those
// passes is Y = X'^T Omega with X' = (X - mean) * scale -- linear in the data, so it can run BEFORE the mean is known:
//
//     Y[j, :] = scale_j * ( sum_i (x_ij - c_j) Omega[i, :]  -  (mean_j - c_j) * sum_i Omega[i, :] )
//
// with any provisional shift c_j (here: the feature's first sample, which keeps x - c small, so the split-fp16
// product loses nothing to cancellation).  atb_f16_fit_kernel computes the first sum on the matrix cores exactly like
// atb_f16_kernel<NB, true> and, from the registers it converts anyway, the per-feature sums of (x - c) and
// (x - c)^2 and max |x - c|; fit_finalize_kernel turns those into mean / std / shift / scale (the same formulas as
// colstats_finalize_kernel) and fit_reduce_kernel applies the rank-one correction while it reduces the split-K
// partials.  A fit then reads the field 2 n_iter + 2 times instead of 2 n_iter + 3.
//
// NaN anywhere, a non-finite value or an overflow of the provisional fp16 scaling shows up in the statistics
// (NaN / inf propagate through the sums; the maximum is compared with the fp16 range) and sends the caller back to
// the two-step path, which owns the NaN policies of the Sanitizer.  gfx950 only.
#pragma once
#include "eofx_kernels.hpp"

namespace eofx {

// Provisional shift c_j = x[0][j] and an estimate of max |x - c| from eight sampled rows.  One thread per four
// adjacent features (P % 4 == 0, 16-byte aligned rows).  flags bit 0: a NaN was seen in the sampled rows.
__global__ __launch_bounds__(256) void fit_probe_kernel(const float* __restrict__ X, int64_t n, int64_t P, int64_t ld,
                                                         int64_t p_pad, float* __restrict__ cshift,
                                                         unsigned* __restrict__ est, int* __restrict__ flags) {
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  float m = 0.f;
  bool bad = false;
  if (c < P) {
    f32x4 c0 = *reinterpret_cast<const f32x4*>(X + c);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (!(c0[e] == c0[e]) || fabsf(c0[e]) == INFINITY) {
        bad = true;
        c0[e] = 0.f;
      }
    *reinterpret_cast<f32x4*>(cshift + c) = c0;
    f32x4 v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int64_t r = t == 7 ? n - 1 : (n * (t + 1)) / 8;
      v[t] = *reinterpret_cast<const f32x4*>(X + (r < n ? r : n - 1) * ld + c);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bad = bad || !(v[t][e] == v[t][e]);
        m = fmaxf(m, fabsf(v[t][e] - c0[e]));
      }
  } else if (c < p_pad) {
    *reinterpret_cast<f32x4*>(cshift + c) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  const bool anybad = __any(bad);
  if ((threadIdx.x & 63) == 0) {
    if (m > 0.f) atomicMax(est, __float_as_uint(m));      // inf included: the host checks
    if (anybad) atomicOr(flags, 1);
  }
}

// ---------------------------------------------------------------------------------
// atb_f16_fit: C = A'^T B with A' = (A - c) * a_scale, A the RAW field [a_rows x a_cols] (lda), plus per-feature
// statistics of A' over the 16-row slabs that lie entirely inside the field (rows [0, 16 floor(a_rows / 16)); the
// < 16 remaining rows are added by fit_finalize_kernel).  Structure, operand layout and split-fp16 arithmetic are
// those of atb_f16_kernel<NB, true> (eofx_kernels.hpp); what differs:
//   * the map is one subtraction and one exact power-of-two scale (2 instead of 3 VALU per element);
//   * per slab and feature the lane sums its 8 values, their squares and max |v| in float32 (8 terms: no
//     accumulation error worth the name) and adds them to float64 accumulators that live in LDS -- the kernel has no
//     registers to spare (246 of 256) -- with fire-and-forget LDS atomics on addresses only this lane touches
//     (program order, hence deterministic);
//   * the epilogue adds the two half-waves (rows 2t and 2t+1) and writes the partial statistics of this split:
//     st_sum / st_sq [split][st_ld] float64, st_max [split][st_ld] float32.
// grid = (M / 512, splits, 1), NB = 1 or 2 (L = 32 NB columns).
// ---------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256, 2) void atb_f16_fit_kernel(const float* __restrict__ A, int64_t lda, int a_rows,
                                                              int64_t a_cols, const float* __restrict__ cshift,
                                                              const float* __restrict__ B, int ldb,
                                                              float* __restrict__ C, int ldc, int64_t M, int64_t K,
                                                              int64_t k_per_split, float a_scale,
                                                              const float* __restrict__ b_absmax,
                                                              double* __restrict__ st_sum, double* __restrict__ st_sq,
                                                              float* __restrict__ st_max, int64_t st_ld) {
  __shared__ __attribute__((aligned(16))) _Float16 Bs[2][2][2][32 * NB][8];
  __shared__ double Ss[4][256], Sq[4][256];
  __shared__ unsigned Sm[4][256];
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
  for (int j = 0; j < 4; ++j) {
    Ss[j][tid] = 0.0;
    Sq[j][tid] = 0.0;
    Sm[j][tid] = 0u;
  }

  const bool colok = m0 + 4 * li < a_cols;
  const float* Ap = A + (colok ? m0 + 4 * li : 0);
  f32x4 cs_ = {0.f, 0.f, 0.f, 0.f};
  if (colok) cs_ = *reinterpret_cast<const f32x4*>(cshift + m0 + 4 * li);
  const float asc_ = colok ? a_scale : 0.f;      // 16-byte chunks beyond the field read chunk 0 and count as zeros
  constexpr int BV = 8 * NB;
  const bool b_loader = tid < 16 * BV;
  const int brow0 = tid / BV, bc4 = tid % BV;
  const float* Bp = B + (kb + brow0) * (int64_t)ldb + 4 * bc4;

  f32x4 a0[8], a1[8];
  f32x4 bn = {0.f, 0.f, 0.f, 0.f};
#define EOFX_FIT_LOAD(areg, chunk)                                                               \
  do {                                                                                           \
    if (b_loader) bn = *reinterpret_cast<const f32x4*>(Bp + (int64_t)(chunk) * ATB_KC * ldb);    \
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
    if (b_loader) {                                                                              \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
        const float r_ = bn[e] * b_scale;                                                        \
        const fp16x2_t h_ = __builtin_amdgcn_cvt_pkrtz(r_, 0.f);                                 \
        const _Float16 m_ = (_Float16)(r_ - (float)h_[0]);                                       \
        Bs[buf][0][brow0 & 1][4 * bc4 + e][brow0 >> 1] = (_Float16)h_[0];                        \
        Bs[buf][1][brow0 & 1][4 * bc4 + e][brow0 >> 1] = m_;                                     \
      }                                                                                          \
    }                                                                                            \
  } while (0)
#define EOFX_FIT_COMPUTE(areg, buf, chunk)                                                       \
  do {                                                                                           \
    const bool st_ok_ = (int)kb + (chunk) * ATB_KC + ATB_KC <= a_rows;   /* uniform */           \
    f16x8 bf_[2][NB];                                                                            \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int q = 0; q < NB; ++q) \
        bf_[s][q] = *reinterpret_cast<const f16x8*>(&Bs[buf][s][lh][32 * q + li][0]);            \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                              \
      f32x8 x_;                                                                                  \
      _Pragma("unroll") for (int t = 0; t < 8; ++t) x_[t] = (areg[t][j] - cs_[j]) * asc_;        \
      if (st_ok_) {                                                                              \
        const float s_ = ((x_[0] + x_[1]) + (x_[2] + x_[3])) + ((x_[4] + x_[5]) + (x_[6] + x_[7])); \
        float q_ = x_[0] * x_[0];                                                                \
        float m_ = fabsf(x_[0]);                                                                 \
        _Pragma("unroll") for (int t = 1; t < 8; ++t) {                                          \
          q_ = __builtin_fmaf(x_[t], x_[t], q_);                                                 \
          m_ = fmaxf(m_, fabsf(x_[t]));                                                          \
        }                                                                                        \
        (void)__hip_atomic_fetch_add(&Ss[j][tid], (double)s_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        (void)__hip_atomic_fetch_add(&Sq[j][tid], (double)q_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        (void)__hip_atomic_fetch_max(&Sm[j][tid], __float_as_uint(m_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
      }                                                                                          \
      f16x8 af_[2];                                                                              \
      split_f16(x_, af_);                                                                        \
      _Pragma("unroll") for (int q = 0; q < NB; ++q) {                                           \
        acc[j][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_[1], bf_[0][q], acc[j][q], 0, 0, 0); \
        acc[j][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_[0], bf_[1][q], acc[j][q], 0, 0, 0); \
        acc[j][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_[0], bf_[0][q], acc[j][q], 0, 0, 0); \
      }                                                                                          \
    }                                                                                            \
  } while (0)

  if (nchunks > 0) {
    EOFX_FIT_LOAD(a0, 0);
    EOFX_FIT_STORE_B(0);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
      EOFX_FIT_LOAD(a1, c + 1);
      __builtin_amdgcn_sched_barrier(0);
      EOFX_FIT_COMPUTE(a0, 0, c);
      EOFX_FIT_STORE_B(1);
      __syncthreads();
      const int c2 = (c + 2 < nchunks) ? c + 2 : c + 1;
      EOFX_FIT_LOAD(a0, c2);
      __builtin_amdgcn_sched_barrier(0);
      EOFX_FIT_COMPUTE(a1, 1, c + 1);
      EOFX_FIT_STORE_B(0);
      __syncthreads();
    }
  }
#undef EOFX_FIT_LOAD
#undef EOFX_FIT_COMPUTE
#undef EOFX_FIT_STORE_B

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
    f64x4 s4, q4;
    f32x4 m4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double s_ = Ss[j][tid], q_ = Sq[j][tid];
      const float m_ = __uint_as_float(Sm[j][tid]);
      const double so_ = __shfl_xor(s_, 32), qo_ = __shfl_xor(q_, 32);
      const float mo_ = __shfl_xor(m_, 32);
      s4[j] = s_ + so_;      // only the lh == 0 lanes store: (rows 2t) + (rows 2t + 1)
      q4[j] = q_ + qo_;
      m4[j] = fmaxf(m_, mo_);
    }
    if (lh == 0 && colok) {
      const int64_t o = (int64_t)blockIdx.y * st_ld + m0 + 4 * li;
      *reinterpret_cast<f64x4*>(st_sum + o) = s4;
      *reinterpret_cast<f64x4*>(st_sq + o) = q4;
      *reinterpret_cast<f32x4*>(st_max + o) = m4;
    }
  }
}

// Statistics of the scaled, provisionally shifted values -> the Scaler's state.  One thread per feature c < p_pad.
//   v = (x - cshift) * a_scale;  S1 = sum v, S2 = sum v^2 over the n samples (split partials in fixed order + the
//   n - n_full tail rows read here), M = max |v|.
//   mean = cshift + S1 / (n a);  M2 = (S2 - S1^2 / n) / a^2;  std = sqrt(M2 / n) clipped at eps  (scaler.py:101-108)
//   shift = center ? mean : 0;  scale = (standardize ? 1 / std : 1) * weight                   (scaler.py:128-154)
// flags: bit 0 a sum is not finite (NaN / inf in the data), bit 1 the provisional fp16 scaling overflowed.
// Also written: the float triples of the in-place view (aff_pack_kernel's layout), dcorr = shift - cshift for
// fit_reduce_kernel, and max |(x - shift) * scale| (an upper bound within a factor of two) into *absmax.
__global__ __launch_bounds__(256) void fit_finalize_kernel(
    const double* __restrict__ st_sum, const double* __restrict__ st_sq, const float* __restrict__ st_max, int64_t st_ld,
    int splits, const float* __restrict__ X, int64_t ld, int64_t n, int64_t n_full, int64_t P, int64_t p_pad,
    const float* __restrict__ cshift, float a_scale, int center, int standardize, const double* __restrict__ weights,
    double eps, int* __restrict__ cnt, double* __restrict__ mean, double* __restrict__ stdv, double* __restrict__ shift,
    double* __restrict__ scale, double* __restrict__ m2, float* __restrict__ aff, double* __restrict__ dcorr,
    unsigned* __restrict__ absmax, int* __restrict__ flags) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float amax = 0.f;
  int fl = 0;
  if (c < P) {
    double s = 0.0, q = 0.0;
    float mx = 0.f;
    for (int sp = 0; sp < splits; ++sp) {
      s += st_sum[(int64_t)sp * st_ld + c];
      q += st_sq[(int64_t)sp * st_ld + c];
      mx = fmaxf(mx, st_max[(int64_t)sp * st_ld + c]);
    }
    const float cs = cshift[c];
    for (int64_t r = n_full; r < n; ++r) {
      const float v = (X[r * ld + c] - cs) * a_scale;
      s += (double)v;
      q += (double)v * (double)v;
      mx = fmaxf(mx, fabsf(v));
    }
    if (!(fabs(s) < INFINITY) || !(fabs(q) < INFINITY)) fl |= 1;
    if (!(mx < 60000.f)) fl |= 2;
    const double ia = 1.0 / (double)a_scale;          // exact: a power of two
    const double S1 = s * ia, nn = (double)n;
    const double mu = (double)cs + S1 / nn;
    double M2 = (q - s * s / nn) * ia * ia;
    if (!(M2 > 0.0)) M2 = 0.0;
    double sd = sqrt(M2 / nn);
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
  } else if (c < p_pad) {
    aff[c] = 0.f;
    aff[p_pad + c] = 0.f;
    aff[2 * p_pad + c] = 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  const int anyfl = __any(fl & 1) | (__any(fl & 2) << 1);
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

// Y[j, :] = scale_j * (sum_splits part[s][j, :] - dcorr_j * wbar[:]) for j < P, 0 for the padding rows; float64
// arithmetic, fixed order.  part may alias out when splits == 1.  amax_out (may be null): max |Y| by atomicMax on the
// float bits (order independent).  count4 = rows_pad * L / 4.
__global__ __launch_bounds__(256) void fit_reduce_kernel(const float* part, float* out, int64_t rows_pad, int L,
                                                          int splits, int64_t P, const double* __restrict__ dcorr,
                                                          const double* __restrict__ scale,
                                                          const double* __restrict__ wbar,
                                                          unsigned* __restrict__ amax_out) {
  const int64_t count4 = rows_pad * (L / 4);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int l4 = L / 4;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += stride) {
    const int64_t j = i / l4;
    const int c = (int)(i - j * l4) * 4;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (j < P) {
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      for (int s = 0; s < splits; ++s) {
        const f32x4 v = reinterpret_cast<const f32x4*>(part)[(int64_t)s * count4 + i];
        s0 += v[0];
        s1 += v[1];
        s2 += v[2];
        s3 += v[3];
      }
      const double d = dcorr[j], sc = scale[j];
      o[0] = (float)((s0 - d * wbar[c]) * sc);
      o[1] = (float)((s1 - d * wbar[c + 1]) * sc);
      o[2] = (float)((s2 - d * wbar[c + 2]) * sc);
      o[3] = (float)((s3 - d * wbar[c + 3]) * sc);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
    }
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
  if (amax_out) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax_out, __float_as_uint(m));
  }
}

}  // namespace eofx
