// eofx_kernels.hpp -- hand-written gfx950 (CDNA4) kernels of the EOF / randomized-SVD engine.
//
// Everything here is written for MI355X only: 64-wide wavefronts, the exact-f32
// MFMA v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak, bitwise an fmaf chain), LDS-staged
// sketch panels, coalesced 16 B/lane HBM streams.  See DESIGN.md for the roofline of
// each kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eofx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The Scaler map (xeofs/preprocessing/scaler.py:153) in float32 arithmetic: (x - mean) * scale with the float64 mean
// carried as a float pair, so the subtraction is exact whenever x and the mean are within a factor of two of each other
// (Sterbenz) and otherwise rounds like any float subtraction; about one ulp from the float64 formula.  ONE definition,
// used by apply_kernel (which writes the sample-contiguous layout) and by the raw view of atb_f16_kernel (which streams
// the raw field): both see the same float32 matrix.
__device__ __forceinline__ void aff_split(double sh, float& hi, float& lo) {
  hi = (float)sh;
  lo = (float)(sh - (double)hi);
}
__device__ __forceinline__ float aff_map(float x, float sh_hi, float sh_lo, float scale) {
  return ((x - sh_hi) - sh_lo) * scale;
}
// The same map as the streaming kernels issue it: two instructions instead of three -- (x - hi) s - lo s with the second
// step one fused multiply-add, nls = -(lo * s) prepared once per feature.  One rounding less than aff_map; identical to it
// whenever s is a power of two (no standardisation, no weights: the products are exact), within one ulp otherwise --
// the written layouts and the streamed view already differed by that much.
__device__ __forceinline__ float aff_fma(float x, float sh_hi, float s, float nls) {
  return __builtin_fmaf(x - sh_hi, s, nls);
}

// One atomicMax per WORKGROUP for a running maximum kept per thread.  Atomics on one address retire one every ~12 ns
// whatever the grid does meanwhile, so a kernel that lets every wave of a few thousand workgroups send its own becomes a
// 0.2 - 0.4 ms kernel however little data it moves (measured: splitk_reduce 193 us with 16 K atomics for 100 MB,
// fit_reduce 374 us at every size); callers also keep such grids at <= 1024 workgroups.  Every thread of the block
// must call it (it synchronises).
__device__ __forceinline__ void amax_commit(float mx, unsigned* __restrict__ amax_out) {
  __shared__ float amax_red_[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) amax_red_[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (int)(blockDim.x >> 6);
    float m = amax_red_[0];
    for (int w = 1; w < nw; ++w) m = fmaxf(m, amax_red_[w]);
    if (m > 0.f) atomicMax(amax_out, __float_as_uint(m));
  }
}

// ---------------------------------------------------------------------------------
// atb_f32: C[M x L] = A[K x M]^T * B[K x L]        (the dominant kernel)
//
//   A  row-major, M contiguous (lda), streamed from HBM exactly once per launch:
//      each lane loads 16 B (4 consecutive m) of two consecutive k-rows straight into
//      the MFMA A-operand registers -- for v_mfma_f32_32x32x2_f32 lane l supplies
//      A[i = l&31][k = l>>5], so the register image of a coalesced row read IS the
//      fragment (row i of sub-tile j is column m0 + 4 i + j).  No LDS round trip for A.
//   B  the sketch panel (K x L, L = 32*NB per z-block), staged through LDS in 16-row
//      slabs shared by the 4 waves of the workgroup, double buffered.
//   C  written once in the epilogue (or as a split-K partial, reduced by a fixed-order
//      second kernel: deterministic, no atomics).
//
//   X^T Z  (tall-skinny sketch/projection):  A = X  [n_pad x p_pad], B = Z [n_pad x L]
//   X  Y   (power-iteration back-product):   A = Xt [p_pad x n_pad], B = Y [p_pad x L]
//
//   grid  = (M/512, splits, ceil(L/64)) ; block = 256 = 4 waves x 128 columns of A.
//   Per wave and 2 k-rows: 1 KiB of A -> 4*NB MFMAs (64 cycles each on its SIMD).
// ---------------------------------------------------------------------------------
constexpr int ATB_KC = 16;    // k-rows per pipeline stage (slab)
constexpr int ATB_KG = 32;    // K granularity: slabs are consumed in pairs
constexpr int ATB_WM = 128;   // A columns per wave
constexpr int ATB_BM = 512;   // A columns per workgroup

template <int NB>
__global__ __launch_bounds__(256, 2) void atb_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                          const float* __restrict__ B, int ldb,
                                                          float* __restrict__ C, int ldc, int64_t M,
                                                          int64_t K, int64_t k_per_split,
                                                          int col_base) {
  __shared__ __attribute__((aligned(16))) float Bs[2][ATB_KC][32 * NB];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * ATB_BM + wave * ATB_WM;
  const int64_t kb = (int64_t)blockIdx.y * k_per_split;
  const int64_t ke = (kb + k_per_split < K) ? kb + k_per_split : K;
  const int nchunks = (int)((ke - kb) / ATB_KC);
  const int bcol0 = col_base + blockIdx.z * 64;

  f32x16 acc[4][NB];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][b][r] = 0.f;

  const float* Ap = A + (kb + lh) * lda + m0 + 4 * li;
  // B slab loader: 16 rows x (8*NB) float4 ; threads [0, 128*NB) take one float4 each
  constexpr int BV = 8 * NB;
  const bool b_loader = tid < 16 * BV;
  const int brow = tid / BV, bc4 = tid % BV;
  const float* Bp = B + (kb + brow) * (int64_t)ldb + bcol0 + 4 * bc4;

  // Two A-fragment register sets (a0/a1): the loads of slab c+1 are issued, and pinned with a
  // scheduling barrier, BEFORE the 32*NB MFMAs of slab c, so a full slab of matrix work
  // (~4k cycles) covers their HBM latency.  Slabs are processed in pairs (the host guarantees
  // an even slab count); the last prefetch index is clamped (a harmless re-read), so the loop
  // body has no data-dependent control flow.
  f32x4 a0[8], a1[8];
  f32x4 bn = {0.f, 0.f, 0.f, 0.f};
#define EOFX_LOAD_SLAB(areg, chunk)                                                              \
  do {                                                                                           \
    if (b_loader) bn = *reinterpret_cast<const f32x4*>(Bp + (int64_t)(chunk) * ATB_KC * ldb);   \
    const float* pa_ = Ap + (int64_t)(chunk) * ATB_KC * lda;                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) areg[i] =                                      \
        __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pa_ + (int64_t)(2 * i) * lda)); \
  } while (0)
#define EOFX_COMPUTE_SLAB(areg, buf)                                                             \
  do {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                              \
      float b_[NB];                                                                              \
      _Pragma("unroll") for (int q = 0; q < NB; ++q) b_[q] = Bs[buf][2 * i + lh][32 * q + li];   \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int q = 0; q < NB; ++q) \
          acc[j][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[i][j], b_[q], acc[j][q], 0, 0, 0); \
    }                                                                                            \
  } while (0)
#define EOFX_STORE_B(buf)                                                                        \
  do {                                                                                           \
    if (b_loader) *reinterpret_cast<f32x4*>(&Bs[buf][brow][4 * bc4]) = bn;                       \
  } while (0)

  if (nchunks > 0) {
    EOFX_LOAD_SLAB(a0, 0);
    EOFX_STORE_B(0);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
      EOFX_LOAD_SLAB(a1, c + 1);
      __builtin_amdgcn_sched_barrier(0);
      EOFX_COMPUTE_SLAB(a0, 0);
      EOFX_STORE_B(1);
      __syncthreads();
      const int c2 = (c + 2 < nchunks) ? c + 2 : c + 1;
      EOFX_LOAD_SLAB(a0, c2);
      __builtin_amdgcn_sched_barrier(0);
      EOFX_COMPUTE_SLAB(a1, 1);
      EOFX_STORE_B(0);
      __syncthreads();
    }
  }
#undef EOFX_LOAD_SLAB
#undef EOFX_COMPUTE_SLAB
#undef EOFX_STORE_B

  // epilogue: D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* Cs = C + (int64_t)blockIdx.y * M * ldc;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ii = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int64_t m = m0 + 4 * ii + j;
        Cs[m * ldc + bcol0 + 32 * q + li] = acc[j][q][r];
      }
}

// ---------------------------------------------------------------------------------
// atb_bf16: the same product C = A^T B on the bf16 matrix cores, with the f32 operands split
// on the fly into PARTS bf16 terms (x = h + m (+ l), each the RNE bf16 of the running
// remainder) and the cross products that matter kept:
//     PARTS = 2 ("bf16x3"):  hh + hm + mh            rel. error per product ~2^-16
//     PARTS = 3 ("bf16x6"):  hh + hm + mh + mm + hl + lh   ~2^-23 (f32 class)
// bf16 x bf16 products are exact in the f32 accumulator.  v_mfma_f32_32x32x16_bf16 retires
// 16 k-rows in 32 cycles (vs 2 rows in 64 for the exact-f32 MFMA), so even six products per
// operand pair leave the kernel HBM-bound: the matrix is still read once, as f32, straight
// into registers -- the same coalesced 16 B/lane row reads as atb_f32.
//
// Fragment mapping: for the 32x32x16 MFMA lane l supplies A[i = l&31][k = 8*(l>>5) + t],
// t = 0..7.  After the 8 row loads of a slab lane (li, lh) holds rows k0 + 2t + lh, so we
// simply *name* MFMA-k index 8*lh + t as slab row 2t + lh; the B fragments are stored in
// LDS under the same naming: Bs[part][kh][col][t] (16 B per (kh, col): one ds_read_b128).
// ---------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// two f32 -> one dword of two RNE bf16 (lo in bits 0-15).  hipcc scalarises vector f32->bf16
// conversions into one v_cvt_pk per element plus packing; the packed form halves the VALU work.
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// x[8] -> PARTS bf16x8 terms: term s is the RNE bf16 of the remainder left by terms < s.
template <int PARTS>
__device__ __forceinline__ void split_bf16(f32x8 r, bf16x8 (&out)[PARTS]) {
#pragma unroll
  for (int s = 0; s < PARTS; ++s) {
    u32x4 pk;
#pragma unroll
    for (int h = 0; h < 4; ++h) pk[h] = cvt_pk_bf16(r[2 * h], r[2 * h + 1]);
    out[s] = __builtin_bit_cast(bf16x8, pk);
    if (s + 1 < PARTS) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        r[2 * h] -= __uint_as_float(pk[h] << 16);
        r[2 * h + 1] -= __uint_as_float(pk[h] & 0xFFFF0000u);
      }
    }
  }
}

template <int NB, int PARTS>
__global__ __launch_bounds__(256, 2) void atb_bf16_kernel(const float* __restrict__ A, int64_t lda,
                                                           const float* __restrict__ B, int ldb,
                                                           float* __restrict__ C, int ldc, int64_t M,
                                                           int64_t K, int64_t k_per_split,
                                                           int col_base) {
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2][PARTS][2][32 * NB][8];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * ATB_BM + wave * ATB_WM;
  const int64_t kb = (int64_t)blockIdx.y * k_per_split;
  const int64_t ke = (kb + k_per_split < K) ? kb + k_per_split : K;
  const int nchunks = (int)((ke - kb) / ATB_KC);
  const int bcol0 = col_base + blockIdx.z * 64;

  f32x16 acc[4][NB];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][b][r] = 0.f;

  const float* Ap = A + (kb + lh) * lda + m0 + 4 * li;
  constexpr int BV = 8 * NB;
  const bool b_loader = tid < 16 * BV;
  const int brow = tid / BV, bc4 = tid % BV;
  const float* Bp = B + (kb + brow) * (int64_t)ldb + bcol0 + 4 * bc4;

  f32x4 a0[8], a1[8];
  f32x4 bn = {0.f, 0.f, 0.f, 0.f};
#define EOFX_LOAD_SLAB(areg, chunk)                                                              \
  do {                                                                                           \
    if (b_loader) bn = *reinterpret_cast<const f32x4*>(Bp + (int64_t)(chunk) * ATB_KC * ldb);   \
    const float* pa_ = Ap + (int64_t)(chunk) * ATB_KC * lda;                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) areg[i] =                                      \
        __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pa_ + (int64_t)(2 * i) * lda)); \
  } while (0)
  // B slab -> LDS, split into bf16 parts, stored fragment-ready: row r = 2t + kh
#define EOFX_STORE_B(buf)                                                                        \
  do {                                                                                           \
    if (b_loader) {                                                                              \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
        float r_ = bn[e];                                                                        \
        _Pragma("unroll") for (int s = 0; s < PARTS; ++s) {                                      \
          const __bf16 h_ = (__bf16)r_;                                                          \
          Bs[buf][s][brow & 1][4 * bc4 + e][brow >> 1] = h_;                                     \
          r_ -= (float)h_;                                                                       \
        }                                                                                        \
      }                                                                                          \
    }                                                                                            \
  } while (0)
#define EOFX_COMPUTE_SLAB(areg, buf)                                                             \
  do {                                                                                           \
    bf16x8 bf_[PARTS][NB];                                                                       \
    _Pragma("unroll") for (int s = 0; s < PARTS; ++s) _Pragma("unroll") for (int q = 0; q < NB; ++q) \
        bf_[s][q] = *reinterpret_cast<const bf16x8*>(&Bs[buf][s][lh][32 * q + li][0]);           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                              \
      f32x8 x_;                                                                                  \
      _Pragma("unroll") for (int t = 0; t < 8; ++t) x_[t] = areg[t][j];                          \
      bf16x8 af_[PARTS];                                                                         \
      split_bf16<PARTS>(x_, af_);                                                                \
      _Pragma("unroll") for (int q = 0; q < NB; ++q) {                                           \
        _Pragma("unroll") for (int sa = PARTS - 1; sa >= 0; --sa)                                \
            _Pragma("unroll") for (int sb = PARTS - 1; sb >= 0; --sb) {                          \
          if (sa + sb < PARTS || (PARTS == 3 && sa == 1 && sb == 1))                             \
            acc[j][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af_[sa], bf_[sb][q], acc[j][q], 0, 0, 0); \
        }                                                                                        \
      }                                                                                          \
    }                                                                                            \
  } while (0)

  if (nchunks > 0) {
    EOFX_LOAD_SLAB(a0, 0);
    EOFX_STORE_B(0);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
      EOFX_LOAD_SLAB(a1, c + 1);
      __builtin_amdgcn_sched_barrier(0);
      EOFX_COMPUTE_SLAB(a0, 0);
      EOFX_STORE_B(1);
      __syncthreads();
      const int c2 = (c + 2 < nchunks) ? c + 2 : c + 1;
      EOFX_LOAD_SLAB(a0, c2);
      __builtin_amdgcn_sched_barrier(0);
      EOFX_COMPUTE_SLAB(a1, 1);
      EOFX_STORE_B(0);
      __syncthreads();
    }
  }
#undef EOFX_LOAD_SLAB
#undef EOFX_COMPUTE_SLAB
#undef EOFX_STORE_B

  float* Cs = C + (int64_t)blockIdx.y * M * ldc;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ii = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int64_t m = m0 + 4 * ii + j;
        Cs[m * ldc + bcol0 + 32 * q + li] = acc[j][q][r];
      }
}

// ---------------------------------------------------------------------------------
// atb_f16: the same product with the operands split into TWO fp16 terms (hi: round-toward-zero
// fp16 of the scaled value, lo: fp16 of the exact remainder) and the three cross products
// hh + hm + mh on v_mfma_f32_32x32x16_f16.  fp16 carries 11 significant bits, so two terms hold 22:
// f32-class accuracy (~2^-22 per product) at the cost of bf16x3.  fp16's narrow exponent range is
// handled by exact power-of-two scaling: A is multiplied by a_scale (from the matrix' max |x|,
// host), B by b_scale (from the panel's max |b|, read from a device scalar written by
// panel_absmax_kernel) so that |scaled| <= 2^14; the epilogue multiplies by the exact inverse.
// Elements far below the maximum fall into fp16 subnormals in their lo term only: an ABSOLUTE
// error of 2^-25 * max, irrelevant against the f32 accumulation.
// ---------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
// The hi / lo split of the scaled split-fp16 products rounds TO NEAREST (v_cvt_pk_f16_f32, one instruction on gfx950 like
// v_cvt_pkrtz).  Round 5: with a truncating split every residual lo = x - hi has the sign of x, so the cross terms hi*lo +
// lo*hi of a coherent sum (a dominant mode) all push the same way; they are ~2^-11 of the main term, get swamped by a long
// float32 accumulation chain inside the MFMA accumulator, and the lost part shows up as a NEGATIVE bias of the product
// (measured: -2e-5 on the leading column of X Y over 1M features, tools/split_precision_probe.py, i.e. 2e-5 on the leading
// singular value -- above the 1e-5 parity tolerance).  With round-to-nearest the residuals have either sign and the same
// effect is zero-mean.
// (inline assembly: written as __builtin_convertvector the conversion costs atb_f16_kernel<2> 33 spilled VGPRs -- 6.4 -> 9.5 ms)
__device__ __forceinline__ fp16x2_t cvt_pk_rn(float a, float b) {
  fp16x2_t r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ void split_f16(f32x8 r, f16x8 (&out)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    u32x4 pk;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const fp16x2_t p = cvt_pk_rn(r[2 * h], r[2 * h + 1]);
      pk[h] = __builtin_bit_cast(unsigned, p);
      if (s == 0) {
        r[2 * h] -= (float)p[0];
        r[2 * h + 1] -= (float)p[1];
      }
    }
    out[s] = __builtin_bit_cast(f16x8, pk);
  }
}

// The same split with the residual x - (float)hi taken by ONE v_fma_mix_f32 (fp16 operand converted on the fly) instead
// of a conversion and a subtraction: m1 must hold -1.0f in a register the optimiser cannot see through
// (asm volatile("" : "+v"(m1))), otherwise it folds the product away and emits the two-instruction form.  Same values.
__device__ __forceinline__ void split_f16_mix(const f32x8 r, float m1, f16x8 (&out)[2]) {
  u32x4 hi, lo;
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const fp16x2_t p = cvt_pk_rn(r[2 * h], r[2 * h + 1]);
    const fp16x2_t q = cvt_pk_rn(__builtin_fmaf((float)p[0], m1, r[2 * h]),
                                                  __builtin_fmaf((float)p[1], m1, r[2 * h + 1]));
    hi[h] = __builtin_bit_cast(unsigned, p);
    lo[h] = __builtin_bit_cast(unsigned, q);
  }
  out[0] = __builtin_bit_cast(f16x8, hi);
  out[1] = __builtin_bit_cast(f16x8, lo);
}

// scale = 2^(14 - e) with |m| <= 2^e (m = max |value|); 1 for an all-zero operand
__device__ __forceinline__ float f16_scale_for(float m) {
  if (!(m > 0.f) || !(m < INFINITY)) return 1.f;
  int e;
  (void)frexpf(m, &e);
  return ldexpf(1.f, 14 - e);
}

// AFF: A is the RAW field [a_rows x a_cols] (lda); the preprocessing map aff_map(x, shift hi, shift lo, scale) -- the
// expression apply_kernel writes into the sample-contiguous layout, so both layouts hold the same float32 matrix -- is
// applied on the fly (the exact power-of-two a_scale folded into the scale), and the feature-contiguous copy of the
// matrix never exists.  Rows >= a_rows are read from the last row (their B rows are zero), 16-byte column chunks
// >= a_cols from chunk 0 with scale 0.  aff = {hi[aff_ld], lo[aff_ld], scale[aff_ld]} from aff_pack_kernel.
// NB = 32-column sub-tiles of B per workgroup: 2 (64 columns: the sketch-width passes, two workgroups per CU), 1 (a
// 32-column remainder) or 4 (128 columns, one workgroup per CU with 256 accumulator registers: the WIDE products -- Gram
// matrices, PCA panels -- re-read A half as often).  sym: C is a symmetric product (B = A): 128-column tiles lying
// entirely below the diagonal are skipped (symmetrize_lower_kernel fills them in afterwards).
// MASK (with AFF): features whose scale is 0 are the field's all-NaN grid points (land / sea masks, sanitizer.py:80-126)
// kept as zero columns instead of being compacted away: their bits are ANDed to +0 before the map, so NaN never reaches
// the matrix cores (one v_and per element; a NaN inside a VALID feature still propagates, as it must).
// amax_out (may be null; single split only): max |C| by atomicMax on the float bits -- the next pass' panel maximum.
template <int NB, bool AFF = false, bool MASK = false>
__global__ __launch_bounds__(256, NB > 2 ? 1 : 2) void atb_f16_kernel(const float* __restrict__ A, int64_t lda,
                                                          const float* __restrict__ B, int ldb,
                                                          float* __restrict__ C, int ldc, int64_t M,
                                                          int64_t K, int64_t k_per_split, int col_base,
                                                          float a_scale,
                                                          const float* __restrict__ b_absmax,
                                                          const float* __restrict__ aff = nullptr,
                                                          int64_t aff_ld = 0, int a_rows = 0, int64_t a_cols = 0,
                                                          const float* A2 = nullptr, const float* B2 = nullptr,
                                                          int s_half = 0, int sym = 0,
                                                          unsigned* __restrict__ amax_out = nullptr, int l_valid = 128) {
  // l_valid (NB = 4 only): columns of this 128-column tile that exist in B and C (96 for the tail of a 96 / 224-column
  // panel: one partial wide tile instead of a 64- and a 32-column launch that would each read the field again).
  // Two-matrix form (complex passes, eofx_rsvd_c64): splits [s_half, 2 s_half) stream a second matrix A2 (same shape)
  // against its own panel B2 -- C = A^T B + A2^T B2 in one launch, summed by the split-K reduction.
  __shared__ __attribute__((aligned(16))) _Float16 Bs[2][2][2][32 * NB][8];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * ATB_BM + wave * ATB_WM;
  int sy_ = blockIdx.y;
  if (A2 != nullptr && sy_ >= s_half) {
    A = A2;
    B = B2;
    sy_ -= s_half;
  }
  const int64_t kb = (int64_t)sy_ * k_per_split;
  const int64_t ke = (kb + k_per_split < K) ? kb + k_per_split : K;
  const int nchunks = (int)((ke - kb) / ATB_KC);
  const int bcol0 = col_base + blockIdx.z * (NB > 2 ? 32 * NB : 64);
  if (sym && bcol0 + 32 * NB <= (int64_t)blockIdx.x * ATB_BM) return;   // strictly below the diagonal (uniform)
  const float b_scale = f16_scale_for(*b_absmax);
  const float out_scale = 1.f / (a_scale * b_scale);   // exact: both are powers of two

  f32x16 acc[4][NB];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][b][r] = 0.f;

  const bool colok = !AFF || m0 + 4 * li < a_cols;
  const float* Ap = A + (AFF ? (colok ? m0 + 4 * li : 0) : (kb + lh) * lda + m0 + 4 * li);
  // {shift hi, shift lo, scale} of this lane's four columns (aff_pack_kernel; scale 0 beyond a_cols)
  f32x4 shh_ = {0.f, 0.f, 0.f, 0.f}, shl_ = {0.f, 0.f, 0.f, 0.f}, sla_ = {0.f, 0.f, 0.f, 0.f};
  if (AFF && colok) {
    shh_ = *reinterpret_cast<const f32x4*>(aff + m0 + 4 * li);
    shl_ = *reinterpret_cast<const f32x4*>(aff + aff_ld + m0 + 4 * li);
    sla_ = *reinterpret_cast<const f32x4*>(aff + 2 * aff_ld + m0 + 4 * li) * a_scale;   // exact: a power of two
  }
  const f32x4 nls_ = -(shl_ * sla_);
  // MASK: a wave whose 128 features are ALL masked streams nothing -- it only takes its part in staging B (its rows of C
  // stay zero) -- and a workgroup of four such waves leaves at once: where masked grid points come in runs (land in an
  // ocean field) their 512-byte row segments drop out of the stream.  (Skipping per LANE inside the common loop made
  // the slab registers conditional and the kernel spill.)
  const bool wave_live = !(AFF && MASK) || __any(sla_[0] != 0.f || sla_[1] != 0.f || sla_[2] != 0.f || sla_[3] != 0.f);
  float m1 = -1.f;   // opaque to the optimiser (split_f16_mix)
  asm volatile("" : "+v"(m1));
  // B staging.  The MFMA fragment of a lane is Bs[plane][lh][column][0..7] = slab rows lh, lh + 2, .., lh + 14 of one
  // column, so a loader item is (column PAIR cp, parity lh, row pair tp): two 8-byte loads (rows 4 tp + lh and + 2) give
  // two columns x two rows, and each column's two rows are ONE packed fp16 pair = one 4-byte LDS store per plane.
  // Item bits, low to high: tp (2), cp low (2), lh (1), cp high; the lh = 1 lanes write their two columns in the
  // opposite order, so the 64 lanes of one store instruction cover all 32 banks twice (free).  [One 2-byte store per
  // element, row per thread, was an 8- to 16-way bank conflict: 88 % of the kernel's LDS cycles,
  // profiles/r03_sq_counters.txt.]
  constexpr int CP = 16 * NB;                      // column pairs of the slab
  constexpr int BREP = (8 * CP + 255) / 256;       // items per thread: 1 (NB = 1: half the threads, NB = 2: all), 2 (NB = 4)
  // per-thread state kept to THREE registers (the kernel has none to spare; a spilled pointer is reloaded from scratch
  // inside the loop, and waiting for a scratch load drains every streaming load in flight): a 32-bit element offset
  // into the slab of B (the slab's first row is a uniform 64-bit base) and the LDS byte offsets of its two stores.
  const bool b_odd = (tid >> 4) & 1;                           // lh
  int b_off, b_w0, b_w1;
  {
    const int tp_ = tid & 3, lh_ = (tid >> 4) & 1;
    const int cp_ = ((tid >> 2) & 3) | ((tid >> 5) << 2);      // + 32 for the second item of the 128-column tile
    b_off = (4 * tp_ + lh_) * ldb + bcol0 + 2 * cp_;               // relative to the slab's first row (uniform part: SGPRs)
    const int w_ = ((lh_ * 32 * NB + 2 * cp_) * 8 + 2 * tp_) * 2;    // bytes of Bs[.][.][lh][2 cp][2 tp]
    b_w0 = w_ + 16 * lh_;
    b_w1 = w_ + 16 * (1 - lh_);
  }
  const bool b_item0 = (((tid >> 2) & 3) | ((tid >> 5) << 2)) < CP;   // NB = 1: only the first 128 threads stage B
  const int b_col0 = 2 * (((tid >> 2) & 3) | ((tid >> 5) << 2));      // first column (inside the tile) of this thread's item 0
  char* const Bsb = reinterpret_cast<char*>(&Bs[0][0][0][0][0]);
  constexpr int B_PLANE = 2 * 32 * NB * 8 * 2, B_BUF = 2 * B_PLANE;   // bytes per fp16 plane / per buffer

  f32x4 a0[8], a1[8];
  f32x4 bn[BREP];      // {row r0: col 2cp, 2cp+1; row r0 + 2: col 2cp, 2cp+1}
#pragma unroll
  for (int r = 0; r < BREP; ++r) bn[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#define EOFX_LOAD_B(chunk)                                                                       \
  do {                                                                                           \
    _Pragma("unroll") for (int r = 0; r < BREP; ++r) {                                           \
      if (b_item0 && (NB < 4 || b_col0 + 64 * r < l_valid)) {   /* (columns beyond l_valid stay zero) */ \
        const float* bp_ = B + (kb + (int64_t)(chunk) * ATB_KC) * ldb + (b_off + 64 * r);         \
        const f32x2 lo_ = *reinterpret_cast<const f32x2*>(bp_);                                  \
        const f32x2 hi_ = *reinterpret_cast<const f32x2*>(bp_ + 2 * ldb);                        \
        bn[r] = f32x4{lo_[0], lo_[1], hi_[0], hi_[1]};                                           \
      }                                                                                          \
    }                                                                                            \
  } while (0)
#define EOFX_LOAD_A(areg, chunk)                                                                 \
  do {                                                                                           \
    if (AFF) {                                                                                   \
      const int r0_ = (int)kb + (chunk) * ATB_KC + lh;                                           \
      if ((int)kb + (chunk) * ATB_KC + ATB_KC <= a_rows) {   /* whole slab inside the field */    \
        const float* pa_ = Ap + (int64_t)r0_ * lda;                                              \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) areg[i] =                                  \
            __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pa_ + (int64_t)(2 * i) * lda)); \
      } else {                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                          \
          const int r_ = r0_ + 2 * i < a_rows ? r0_ + 2 * i : a_rows - 1;                        \
          areg[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Ap + (int64_t)r_ * lda)); \
        }                                                                                        \
      }                                                                                          \
    } else {                                                                                     \
      const float* pa_ = Ap + (int64_t)(chunk) * ATB_KC * lda;                                   \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) areg[i] =                                    \
          __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pa_ + (int64_t)(2 * i) * lda)); \
    }                                                                                            \
  } while (0)
#define EOFX_STORE_B(buf)                                                                        \
  do {                                                                                           \
    _Pragma("unroll") for (int r = 0; r < BREP; ++r) {                                           \
      if (b_item0) _Pragma("unroll") for (int e2 = 0; e2 < 2; ++e2) {                            \
        const bool sec_ = (e2 != 0) != b_odd;        /* which column of the pair this store takes */ \
        const float v0_ = (sec_ ? bn[r][1] : bn[r][0]) * b_scale, v1_ = (sec_ ? bn[r][3] : bn[r][2]) * b_scale; \
        const fp16x2_t h_ = cvt_pk_rn(v0_, v1_);                                \
        fp16x2_t l_;                                                                             \
        l_[0] = (__fp16)__builtin_fmaf((float)h_[0], m1, v0_);                                   \
        l_[1] = (__fp16)__builtin_fmaf((float)h_[1], m1, v1_);                                   \
        char* w_ = Bsb + (buf) * B_BUF + (e2 ? b_w1 : b_w0) + 64 * 16 * r;                       \
        *reinterpret_cast<unsigned*>(w_) = __builtin_bit_cast(unsigned, h_);                     \
        *reinterpret_cast<unsigned*>(w_ + B_PLANE) = __builtin_bit_cast(unsigned, l_);           \
      }                                                                                          \
    }                                                                                            \
  } while (0)
#define EOFX_CONVERT_J(areg, j, af_)                                                             \
  do {                                                                                           \
    f32x8 x_;                                                                                    \
    const unsigned msk_ = (AFF && MASK) ? (sla_[j] != 0.f ? 0xffffffffu : 0u) : 0xffffffffu;     \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                              \
      const float xr_ = (AFF && MASK) ? __uint_as_float(__float_as_uint(areg[t][j]) & msk_) : areg[t][j]; \
      x_[t] = AFF ? aff_fma(xr_, shh_[j], sla_[j], nls_[j]) : xr_ * a_scale;                      \
    }                                                                                            \
    split_f16_mix(x_, m1, af_);                                                                  \
  } while (0)
#define EOFX_MFMA_J(j, af_)                                                                      \
  _Pragma("unroll") for (int q = 0; q < NB; ++q) {                                               \
    acc[j][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_[1], bf_[0][q], acc[j][q], 0, 0, 0);   \
    acc[j][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_[0], bf_[1][q], acc[j][q], 0, 0, 0);   \
    acc[j][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_[0], bf_[0][q], acc[j][q], 0, 0, 0);   \
  }
  // One slab: its registers are REFILLED (slab `refill`) as soon as the last of its four feature groups has been
  // converted -- before that group's MFMAs, the B staging and the barrier: a slab's registers are out of the memory
  // pipeline only for the length of their conversion.  (Refilling after the whole slab cost 4-5 % of the bandwidth:
  // two slabs per wave in flight against ~3.6 us of loaded latency is what paces the kernel.)
#define EOFX_COMPUTE_SLAB(areg, buf, refill)                                                     \
  do {                                                                                           \
    f16x8 bf_[2][NB];                                                                            \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int q = 0; q < NB; ++q) \
        bf_[s][q] = *reinterpret_cast<const f16x8*>(&Bs[buf][s][lh][32 * q + li][0]);            \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                              \
      f16x8 af_[2];                                                                              \
      EOFX_CONVERT_J(areg, j, af_);                                                              \
      EOFX_MFMA_J(j, af_)                                                                        \
    }                                                                                            \
    f16x8 al_[2];                                                                                \
    EOFX_CONVERT_J(areg, 3, al_);                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    EOFX_LOAD_A(areg, refill);                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    EOFX_MFMA_J(3, al_)                                                                          \
  } while (0)

  bool wg_live = true;
  if constexpr (AFF && MASK) wg_live = __syncthreads_or(wave_live) != 0;
  if (nchunks > 0 && wg_live && !wave_live) {   // (MASK) the B hand-over only: the same barriers as the streaming waves
    EOFX_LOAD_B(0);
    EOFX_STORE_B(0);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
      const int c2 = (c + 2 < nchunks) ? c + 2 : c;
      EOFX_LOAD_B(c + 1);
      EOFX_STORE_B(1);
      __syncthreads();
      EOFX_LOAD_B(c2);
      EOFX_STORE_B(0);
      __syncthreads();
    }
  } else if (nchunks > 0 && wg_live) {     // nchunks is even (K and k_per_split are multiples of ATB_KG = two slabs)
    EOFX_LOAD_B(0);
    EOFX_LOAD_A(a0, 0);
    EOFX_LOAD_A(a1, 1);
    EOFX_STORE_B(0);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
      const int c2 = (c + 2 < nchunks) ? c + 2 : c;      // past the end: harmless re-reads of this pair
      EOFX_LOAD_B(c + 1);
      EOFX_COMPUTE_SLAB(a0, 0, c2);
      EOFX_STORE_B(1);
      __syncthreads();
      EOFX_LOAD_B(c2);
      EOFX_COMPUTE_SLAB(a1, 1, c2 + 1);
      EOFX_STORE_B(0);
      __syncthreads();
    }
  }
#undef EOFX_LOAD_B
#undef EOFX_LOAD_A
#undef EOFX_COMPUTE_SLAB
#undef EOFX_CONVERT_J
#undef EOFX_MFMA_J
#undef EOFX_STORE_B

  float* Cs = C + (int64_t)blockIdx.y * M * ldc;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ii = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int64_t m = m0 + 4 * ii + j;
        if (NB < 4 || 32 * q + li < l_valid) Cs[m * ldc + bcol0 + 32 * q + li] = acc[j][q][r] * out_scale;
      }
  if (amax_out) {
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(acc[j][q][r]));
    amax_commit(mx * out_scale, amax_out);
  }
}

// ---------------------------------------------------------------------------------
// axb_f16: C[M x 64] = A'[M x K] B[K x 64] with A the RAW field read IN PLACE (row-major: rows = samples,
// the K = feature axis contiguous) and A' = aff_map(A) applied on the fly -- the sample-side product X Y of the power
// iteration without a sample-contiguous copy of the matrix.  Same scaled split-fp16 scheme as atb_f16 (hh + hl + lh).
//
//   A  a wave owns 64 rows.  It streams them with fully coalesced non-temporal 16-byte loads (one instruction = 8 rows
//      x one 128-byte line each: every line is touched exactly once and does not settle in L2, which stays free for
//      the B slabs), maps + splits its 4 features per lane in registers -- a lane keeps the same 4 features of the slab
//      for all 8 row groups, so its {shift hi, shift lo, scale} triples are three 16-byte loads per slab -- and
//      transposes through a PRIVATE LDS region ([plane][row][32 halves], 16-byte chunks XOR-swizzled by the row:
//      conflict-free 8-byte stores and 16-byte reads in the operand layout of v_mfma_f32_16x16x32_f16).  Wave-private, so no workgroup barrier guards
//      it.  (Loading in the operand layout directly -- one row per lane -- needs cached loads, the second touch of every
//      line then relies on L1 and the stream evicts the B slabs from L2: 17 % more HBM traffic, measured.)
//   B  32 x 64 slab through LDS as two fp16 planes [k-group][column][8], XOR-swizzled columns, written as packed
//      pairs, shared by the 4 waves, double buffered.
//   C  split-K partials [split][c_rows][ldc], reduced in fixed order by splitk_reduce_kernel.
//
//   grid = (8 * row_tiles * ceil(splits / 8), L/64); block = 256 = 4 waves x 64 rows.  Workgroup ids are dealt to the 8
//   XCDs round-robin, so id -> (xcd = id % 8, slot = id / 8), split = xcd + 8 (slot / row_tiles), row tile = slot %
//   row_tiles: all row tiles of one split run on ONE XCD and share its B slabs through that XCD's L2.
//   Rows >= a_rows read the last row and write zeros; 16-byte feature chunks >= a_cols read chunk 0 (their scale is 0).
// ---------------------------------------------------------------------------------
constexpr int AXB_KC = 32;    // features per slab
constexpr int AXB_KG = 64;    // K granularity: slabs are consumed in pairs
constexpr int AXB_BM = 256;   // rows per workgroup
constexpr int AXB_LDA = 32;   // halves per staged row: 64 bytes = four 16-byte chunks, chunk c of row r stored at c ^ ((r >> 1) & 3)

// DBG (tools/probes/axb_probe.hip only): 1 no MFMA, 2 no conversion either, 4 no B / map loads, 8 cached A loads,
// 16 no B conversion / staging stores, 32 B staged as raw bits (loads, waits and stores without the arithmetic)
// MASK: as in atb_f16_kernel -- features with scale 0 (all-NaN grid points kept as zero columns) are ANDed to +0.
template <int NQ, int DBG = 0, bool MASK = false>   // 16-column tiles per workgroup column block: 4 (64 columns) or 2 (a 32-column remainder)
__global__ __launch_bounds__(256, 2) void axb_f16_kernel(const float* __restrict__ A, int64_t lda, int a_rows,
                                                          int64_t a_cols, const float* __restrict__ aff, int64_t aff_ld,
                                                          const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                          int ldc, int64_t c_rows, int64_t K, int64_t k_per_split,
                                                          int splits, int row_tiles, int col_base, float a_scale,
                                                          const float* __restrict__ b_absmax,
                                                          const int* __restrict__ act = nullptr) {
  __shared__ __attribute__((aligned(16))) _Float16 Bs[2][2][8][64][8];    // [buffer][plane][k-group of 8][column slot][8]
  __shared__ __attribute__((aligned(16))) _Float16 As[4][2][64][AXB_LDA];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: scalar branches below
  const int ln = lane & 15, g = lane >> 4;
  const int lr = lane >> 3, lc = lane & 7;   // loader view: row inside a group of 8, 16-byte chunk of the 128-byte line
  // A staging without bank conflicts on either side: rows are 64 bytes (no padding) and the 16-byte chunk c of row r sits
  // at chunk c ^ ((r >> 1) & 3).  A store instruction (8-byte items, 16 lanes per LDS cycle = two adjacent rows x 64
  // bytes) then covers all 32 banks once, and so does a 16-byte fragment read (8 lanes per cycle = 8 rows, one chunk
  // each).  [Rows padded to 80 bytes were conflict-free for the reads only: every store took twice its cycles, 64 of the
  // 113 bank-conflict cycles per slab and wave in profiles/r03_sq_counters.txt.]  Halves: column offsets in a row.
  const int a_wc = 8 * ((lc >> 1) ^ ((lr >> 1) & 3)) + 4 * (lc & 1);   // store: halves 4 lc .. 4 lc + 3 of row 8 u + lr
  const int a_rc = 8 * (g ^ ((ln >> 1) & 3));                          // read:  halves 8 g .. 8 g + 7 of row 16 t + ln
  const int slot_ = splits > 1 ? (int)blockIdx.x >> 3 : (int)blockIdx.x;
  const int split = splits > 1 ? ((int)blockIdx.x & 7) + 8 * (slot_ / row_tiles) : 0;
  if (split >= splits) return;
  const int r0 = (slot_ % row_tiles) * AXB_BM + wave * 64;
  const bool live = r0 < a_rows;
  const bool full = r0 + 64 <= a_rows;
  const unsigned ldab = (unsigned)lda * 4u;   // row pitch in bytes
  const unsigned lrl = (unsigned)lr * ldab;   // this lane's row inside a group of 8
  // MASK with an active list (`act`: the 64-feature slab pairs that hold at least one unmasked feature, ascending): K and
  // k_per_split count ACTIVE features, the split walks its share of the list and the addresses come from the list
  // entries -- slab pairs made of masked grid points only (land in an ocean field) are never read.
  const bool listed = MASK && act != nullptr;
  const int64_t kb_ = (int64_t)split * k_per_split;
  const int64_t ke = (kb_ + k_per_split < K) ? kb_ + k_per_split : K;
  const int nslab = (int)((ke - kb_) / AXB_KC);
  const int64_t kb = (MASK && listed) ? 0 : kb_;           // address base of the split
  const int* const actp = (MASK && listed) ? act + kb_ / AXB_KG : nullptr;
#define EOFX_PAIR(i) (listed ? actp[(i)] : (i))            /* absolute pair id of the split's i-th pair (uniform) */
  const int bcol0 = col_base + blockIdx.y * 64;
  const float b_scale = f16_scale_for(*b_absmax);
  const float out_scale = 1.f / (a_scale * b_scale);   // exact: both are powers of two

  f32x4 acc[4][NQ];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A: uniform base of the wave's row block + 32-bit byte offsets (the launcher guarantees 64 lda + K < 2^30 elements)
  const char* const Ab = reinterpret_cast<const char*>(A + (int64_t)(live ? r0 : 0) * lda);
  const int arow0 = live ? r0 : 0;     // dead waves stream the first rows (their results are discarded)
  // 32-bit BYTE offsets from uniform bases for the small streams (the launcher guarantees K * ldb < 2^30)
  const float* const aff1 = aff + aff_ld;
  const float* const aff2 = aff + 2 * aff_ld;
  const int fo = (int)kb + 4 * lc;
  // B loader: two adjacent k rows, four columns: k = 2 kp, kp = (lane >> 4) | (wave << 2) -> k-group = wave
  const int bc4 = tid & 15;
  const int bk = 2 * ((lane >> 4) | (wave << 2));
  const int bt = bk & 7;
  const bool b_loader = bc4 < 4 * NQ;
  const int bo = ((int)kb + bk) * ldb + bcol0 + 4 * (bc4 % (4 * NQ));

  f32x4 a0[8], a1[8], fr[3];
  float m1 = -1.f;   // opaque to the optimiser: x - (float)h stays ONE v_fma_mix_f32 instead of a conversion and a subtraction
  asm volatile("" : "+v"(m1));
  f32x4 bn[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#define EOFX_AXB_LD(p_) ((DBG & 8) ? *reinterpret_cast<const f32x4*>(p_) : __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p_)))
  // Every thread issues the same loads on every path (the partial-tile branch issues as many as the full one): the
  // compiler's vmcnt bookkeeping stays exact, and a wait for an older load leaves the younger ones in flight.
#define EOFX_LOAD_BH(pair, hb)   /* B slab = 64 features = TWO A slabs; half hb: rows bk, bk + 1 of that half */ \
  do {                                                                                                 \
    if (!(DBG & 4)) {                                                                                  \
      const unsigned bb_ = (unsigned)(bo + ((pair) * AXB_KG + 32 * (hb)) * ldb) * 4u;   /* K ldb < 2^30 */ \
      bn[0] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(B) + bb_);                 \
      bn[1] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(B) + (bb_ + (unsigned)ldb * 4u)); \
    }                                                                                                  \
  } while (0)
#define EOFX_LOAD_F(freg, chunk)                                                                       \
  do {                                                                                                 \
    if (!(DBG & 4)) {                                                                                  \
      const unsigned fb_ = (unsigned)(fo + (chunk) * AXB_KC) * 4u;   /* K < 2^30: scalar base + 32-bit offset */ \
      freg[0] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(aff) + fb_);             \
      freg[1] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(aff1) + fb_);            \
      freg[2] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(aff2) + fb_);            \
    }                                                                                                  \
  } while (0)
  // row groups u0 .. u0+3 (32 rows) of slab `chunk`
#define EOFX_LOAD_A(areg, chunk, u0)                                                                   \
  do {                                                                                                 \
    const int ko_ = (chunk) * AXB_KC;                                                                  \
    const bool kin_ = fo + ko_ < a_cols;                                                               \
    const unsigned kof_ = kin_ ? (unsigned)(fo + ko_) * 4u : 0u;                                       \
    if (full) {                                                                                        \
      /* per-slab base in ONE register + uniform multiples of the row pitch: nothing per row group for the */ \
      /* optimiser to hoist out of the loop (eight hoisted offsets spilled, and a scratch reload drains vmcnt) */ \
      unsigned base_ = lrl + kof_;                                                                     \
      asm volatile("" : "+v"(base_));                                                                  \
      _Pragma("unroll") for (int u = (u0); u < (u0) + 4; ++u)                                          \
          areg[u] = EOFX_AXB_LD(Ab + (base_ + (unsigned)(8 * u) * ldab));                              \
    } else {                                                                                           \
      int lr_ = lr;   /* the one partial wave per split: recomputed per slab, nothing kept live */      \
      asm volatile("" : "+v"(lr_));                                                                    \
      _Pragma("unroll") for (int u = (u0); u < (u0) + 4; ++u) {                                        \
        const int r_ = arow0 + lr_ + 8 * u < a_rows ? lr_ + 8 * u : a_rows - 1 - arow0;                \
        areg[u] = EOFX_AXB_LD(Ab + ((unsigned)r_ * ldab + kof_));                                      \
      }                                                                                                \
    }                                                                                                  \
  } while (0)
#define EOFX_STORE_BH(buf, hb)                                                                         \
  do {                                                                                                 \
    if (b_loader && !(DBG & 16)) _Pragma("unroll") for (int e = 0; e < 4; ++e) {                       \
      const int col_ = 4 * bc4 + e;                                                                    \
      const int sl_ = col_ ^ ((col_ >> 3) & 7);                                                        \
      const float v0_ = bn[0][e] * b_scale, v1_ = bn[1][e] * b_scale;                                  \
      fp16x2_t h_, l_;                                                                                 \
      if (DBG & 32) {   /* raw bits: the loads, the waits and the LDS stores without the arithmetic */   \
        h_ = __builtin_bit_cast(fp16x2_t, bn[0][e]); l_ = __builtin_bit_cast(fp16x2_t, bn[1][e]);      \
      } else {                                                                                         \
        h_ = cvt_pk_rn(v0_, v1_);                                                     \
        l_[0] = (__fp16)__builtin_fmaf((float)h_[0], m1, v0_);                                         \
        l_[1] = (__fp16)__builtin_fmaf((float)h_[1], m1, v1_);                                         \
      }                                                                                                \
      *reinterpret_cast<unsigned*>(&Bs[buf][0][wave + 4 * (hb)][sl_][bt]) = __builtin_bit_cast(unsigned, h_); \
      *reinterpret_cast<unsigned*>(&Bs[buf][1][wave + 4 * (hb)][sl_][bt]) = __builtin_bit_cast(unsigned, l_); \
    }                                                                                                  \
  } while (0)
  // map + split this lane's 4 features of row groups u0 .. u0+3 into the wave's LDS region
#define EOFX_AXB_CONVERT(areg, u0)                                                                     \
  _Pragma("unroll") for (int u = (u0); u < (u0) + 4; ++u) {                                            \
    u32x2 hi_, lo_;                                                                                    \
    if (DBG & 2) {                                                                                     \
      hi_[0] = __float_as_uint(areg[u][0]); hi_[1] = __float_as_uint(areg[u][1]);                      \
      lo_[0] = __float_as_uint(areg[u][2]); lo_[1] = __float_as_uint(areg[u][3]);                      \
    } else {                                                                                           \
      _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                  \
        /* aff_map on a pair (packed float32 arithmetic: the same three roundings per element) */       \
        const f32x2 x_ = {MASK ? __uint_as_float(__float_as_uint(areg[u][2 * h]) & mk_[2 * h]) : areg[u][2 * h],           \
                          MASK ? __uint_as_float(__float_as_uint(areg[u][2 * h + 1]) & mk_[2 * h + 1]) : areg[u][2 * h + 1]}; \
        f32x2 t_;   /* x - hi as ONE packed add with negated second source (no v_pk_sub_f32 exists) */            \
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t_) : "v"(x_), "v"(fh_[h]));     \
        f32x2 v_;   /* aff_fma: (x - hi) s - lo s, packed (the optimiser splits half of these otherwise) */       \
        asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(v_) : "v"(t_), "v"(fs_[h]), "v"(fl_[h])); \
        const fp16x2_t p_ = cvt_pk_rn(v_[0], v_[1]);                                  \
        const fp16x2_t q_ = cvt_pk_rn(__builtin_fmaf((float)p_[0], m1, v_[0]),        \
                                                       __builtin_fmaf((float)p_[1], m1, v_[1]));        \
        hi_[h] = __builtin_bit_cast(unsigned, p_);                                                     \
        lo_[h] = __builtin_bit_cast(unsigned, q_);                                                     \
      }                                                                                                \
    }                                                                                                  \
    *reinterpret_cast<u32x2*>(&As[wave][0][8 * u + lr][a_wc]) = hi_;                                   \
    *reinterpret_cast<u32x2*>(&As[wave][1][8 * u + lr][a_wc]) = lo_;                                   \
  }
  // 32 rows (two 16-row tiles) x NQ column tiles: the three products ordered so that dependent MFMAs are 2 NQ apart
#define EOFX_AXB_MFMA(jh, buf, hs)     /* hs: which half of the 64-feature B slab this A slab is */      \
  do {                                                                                                 \
    f16x8 af_[2][2], bf_[2][NQ];                                                                       \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int s = 0; s < 2; ++s)        \
        af_[j][s] = *reinterpret_cast<const f16x8*>(&As[wave][s][16 * (2 * (jh) + j) + ln][a_rc]);     \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                   \
      const int col_ = 16 * q + ln;                                                                    \
      const int sl_ = col_ ^ ((col_ >> 3) & 7);                                                        \
      bf_[0][q] = *reinterpret_cast<const f16x8*>(&Bs[buf][0][g + 4 * (hs)][sl_][0]);                  \
      bf_[1][q] = *reinterpret_cast<const f16x8*>(&Bs[buf][1][g + 4 * (hs)][sl_][0]);                  \
    }                                                                                                  \
    if (DBG & 1) {                                                                                     \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) _Pragma("unroll") for (int j = 0; j < 2; ++j)     \
          _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[2 * (jh) + j][q][r] +=                     \
              (float)af_[j][0][r] + (float)af_[j][1][r + 4] + (float)bf_[0][q][r] + (float)bf_[1][q][r]; \
    } else {                                                                                           \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) _Pragma("unroll") for (int j = 0; j < 2; ++j)     \
          acc[2 * (jh) + j][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af_[j][1], bf_[0][q], acc[2 * (jh) + j][q], 0, 0, 0); \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) _Pragma("unroll") for (int j = 0; j < 2; ++j)     \
          acc[2 * (jh) + j][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af_[j][0], bf_[1][q], acc[2 * (jh) + j][q], 0, 0, 0); \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) _Pragma("unroll") for (int j = 0; j < 2; ++j)     \
          acc[2 * (jh) + j][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af_[j][0], bf_[0][q], acc[2 * (jh) + j][q], 0, 0, 0); \
    }                                                                                                  \
  } while (0)
  // One A slab (32 features, half `hs` of the current 64-feature B slab in LDS buffer `buf`): rows 0..31 are converted,
  // their registers immediately take the loads of the slab AFTER the next (two slabs of A stay in flight per wave
  // through the matrix work and the barrier), their MFMAs run under the conversion of rows 32..63.
#define EOFX_SLAB(areg, buf, hs, next_f, next_a)                                                       \
  do {                                                                                                 \
    if (live) {                                                                                        \
      f32x2 fh_[2], fl_[2], fs_[2];                                                                    \
      _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                  \
        fh_[h] = f32x2{fr[0][2 * h], fr[0][2 * h + 1]};                                                \
        fs_[h] = f32x2{fr[2][2 * h], fr[2][2 * h + 1]} * a_scale;   /* exact: a power of two */         \
        fl_[h] = f32x2{fr[1][2 * h], fr[1][2 * h + 1]} * fs_[h];    /* lo * s; aff_fma subtracts it */   \
      }                                                                                                \
      unsigned mk_[4];                                                                                 \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) mk_[e] = (MASK && fr[2][e] == 0.f) ? 0u : 0xffffffffu; \
      (void)mk_;                                                                                       \
      EOFX_LOAD_F(fr, next_f);    /* the triples of the NEXT slab (L2 hits: one slab of lead is enough) */ \
      EOFX_AXB_CONVERT(areg, 0)                                                                        \
      EOFX_LOAD_A(areg, next_a, 0);                                                                    \
      EOFX_AXB_MFMA(0, buf, hs);                                                                       \
      EOFX_AXB_CONVERT(areg, 4)                                                                        \
      EOFX_LOAD_A(areg, next_a, 4);                                                                    \
      EOFX_AXB_MFMA(1, buf, hs);                                                                       \
    } else {                                                                                           \
      EOFX_LOAD_F(fr, next_f);                                                                         \
      EOFX_LOAD_A(areg, next_a, 0);                                                                    \
      EOFX_LOAD_A(areg, next_a, 4);                                                                    \
    }                                                                                                  \
  } while (0)

  // One pair = one 64-feature B slab (LDS buffer pb) = two A slabs.  The B slab of the NEXT pair goes to the other
  // buffer half by half (8 registers in flight, not 16): its first half was requested during the previous pair, its
  // second half is requested now and stored between the two slabs.  ONE workgroup barrier per pair: the B hand-over is
  // the only shared state (with a barrier per slab the kernel ran 3 % slower).
  if constexpr (!MASK) {
    if (nslab > 0) {   // nslab is even (k_per_split and K are multiples of AXB_KG = 64 = one B slab = two A slabs)
      const int npair = nslab / 2;
      EOFX_LOAD_BH(0, 0);
      EOFX_LOAD_F(fr, 0);
      EOFX_LOAD_A(a0, 0, 0);
      EOFX_LOAD_A(a0, 0, 4);
      EOFX_STORE_BH(0, 0);
      EOFX_LOAD_BH(0, 1);
      EOFX_LOAD_A(a1, 1, 0);
      EOFX_LOAD_A(a1, 1, 4);
      EOFX_STORE_BH(0, 1);
      EOFX_LOAD_BH(npair > 1 ? 1 : 0, 0);
      __syncthreads();
      for (int pr = 0; pr < npair; ++pr) {
        const int pb = pr & 1;
        const int c2 = pr + 1 < npair ? 2 * pr + 2 : 2 * pr;          // past the end: harmless re-reads of the last pair
        const int p1 = pr + 1 < npair ? pr + 1 : npair - 1, p2 = pr + 2 < npair ? pr + 2 : npair - 1;
        EOFX_STORE_BH(1 - pb, 0);
        EOFX_LOAD_BH(p1, 1);
        EOFX_SLAB(a0, pb, 0, 2 * pr + 1, c2);
        EOFX_STORE_BH(1 - pb, 1);
        EOFX_LOAD_BH(p2, 0);
        EOFX_SLAB(a1, pb, 1, c2, c2 + 1);
        __syncthreads();
      }
    }
  } else if (nslab > 0) {   // the same schedule with the pair ids taken from the active list (or the identity)
    const int npair = nslab / 2;
    {
      const int q0 = EOFX_PAIR(0), q1 = EOFX_PAIR(npair > 1 ? 1 : 0);
      EOFX_LOAD_BH(q0, 0);
      EOFX_LOAD_F(fr, 2 * q0);
      EOFX_LOAD_A(a0, 2 * q0, 0);
      EOFX_LOAD_A(a0, 2 * q0, 4);
      EOFX_STORE_BH(0, 0);
      EOFX_LOAD_BH(q0, 1);
      EOFX_LOAD_A(a1, 2 * q0 + 1, 0);
      EOFX_LOAD_A(a1, 2 * q0 + 1, 4);
      EOFX_STORE_BH(0, 1);
      EOFX_LOAD_BH(q1, 0);
    }
    __syncthreads();
    for (int pr = 0; pr < npair; ++pr) {
      const int pb = pr & 1;
      // pair ids of this, the next and the next-but-one pair (past the end: harmless re-reads of the last pair)
      const int q0 = EOFX_PAIR(pr), p1 = EOFX_PAIR(pr + 1 < npair ? pr + 1 : npair - 1);
      const int p2 = EOFX_PAIR(pr + 2 < npair ? pr + 2 : npair - 1);
      const int c2 = 2 * p1;
      EOFX_STORE_BH(1 - pb, 0);
      EOFX_LOAD_BH(p1, 1);
      EOFX_SLAB(a0, pb, 0, 2 * q0 + 1, c2);
      EOFX_STORE_BH(1 - pb, 1);
      EOFX_LOAD_BH(p2, 0);
      EOFX_SLAB(a1, pb, 1, c2, c2 + 1);
      __syncthreads();
    }
  }
#undef EOFX_PAIR
#undef EOFX_AXB_LD
#undef EOFX_LOAD_BH
#undef EOFX_LOAD_F
#undef EOFX_LOAD_A
#undef EOFX_STORE_BH
#undef EOFX_AXB_CONVERT
#undef EOFX_AXB_MFMA
#undef EOFX_SLAB

  // D[i = 4 g + r][n = ln] of tile (j, q): row r0 + 16 j + 4 g + r, column 16 q + ln
  float* Cs = C + (int64_t)split * c_rows * ldc;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + 16 * j + 4 * g + r;
      if (row < c_rows) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          Cs[(int64_t)row * ldc + bcol0 + 16 * q + ln] = row < a_rows ? acc[j][q][r] * out_scale : 0.f;
      }
    }
}

// v <- -v (the sharded sign rule all-reduces [max | -min] with ONE max collective)
__global__ void negate_kernel(float* __restrict__ v, int count) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) v[i] = -v[i];
}

// max |v| over a (rows x cols) block with leading dimension ld -> *out (float bits, atomicMax on the
// unsigned view: order-independent, hence deterministic).  *out must be zeroed first.
__global__ __launch_bounds__(256) void panel_absmax_kernel(const float* __restrict__ P, int64_t rows,
                                                            int cols, int64_t ld,
                                                            unsigned* __restrict__ out) {
  __shared__ float red[4];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float m = 0.f;
  if (ld == cols) {   // contiguous block (every panel): plain linear sweep, no index arithmetic
    const int64_t total = rows * (cols / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const f32x4 v = reinterpret_cast<const f32x4*>(P)[i];
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
  } else {
    const int c4n = cols / 4;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x)
      for (int c4 = threadIdx.x; c4 < c4n; c4 += blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(P + r * ld + 4 * c4);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {   // one atomic per workgroup
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > 0.f) atomicMax(out, __float_as_uint(m));
  }
}

// ---------------------------------------------------------------------------------
// atb_f64: the same product with float64 multiply-accumulate on the fp64 matrix cores (v_mfma_f64_16x16x4_f64):
// operands are the float32 data and panel converted exactly, every product is exact (24 + 24 bits) and the sums are
// float64 -- the arithmetic the reference does after promoting the field (xeofs/utils/xarray_utils.py:78-100).
// MFMA-bound: 2 K M L flop at the 78.6 TFLOP/s float64 peak, about 3x the time of the HBM-bound split-fp16 pass.
//   block = 512 threads = 8 waves x 64 columns of A; lane (c = l % 16, kq = l / 16) loads 16 B (columns 4c..4c+3) of
//   row k0 + kq: the four floats are element (i = c, k = kq) of FOUR A fragments (sub-tile t holds columns 4 i + t).
//   Split-K partials are float64 (Cd), reduced by splitk_reduce_f64_kernel; a single split writes float32 directly.
// ---------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NB>
__global__ __launch_bounds__(512) void atb_f64_kernel(const float* __restrict__ A, int64_t lda,
                                                       const float* __restrict__ B, int ldb,
                                                       float* __restrict__ C, double* __restrict__ Cd, int ldc,
                                                       int64_t M, int64_t K, int64_t k_per_split, int col_base) {
  __shared__ __attribute__((aligned(16))) double Bs[2][ATB_KC][32 * NB];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int c16 = lane & 15, kq = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * ATB_BM + wave * 64;
  const int64_t kb = (int64_t)blockIdx.y * k_per_split;
  const int64_t ke = (kb + k_per_split < K) ? kb + k_per_split : K;
  const int nchunks = (int)((ke - kb) / ATB_KC);
  const int bcol0 = col_base + blockIdx.z * 64;
  constexpr int NT = 2 * NB;          // 16-column tiles of the panel

  f64x4 acc[4][NT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[t][q] = f64x4{0.0, 0.0, 0.0, 0.0};

  const float* Ap = A + (kb + kq) * lda + m0 + 4 * c16;
  constexpr int BV = 8 * NB;                      // float4 per panel row
  const bool b_loader = tid < 16 * BV;
  const int brow = tid / BV, bc4 = tid % BV;
  const float* Bp = B + (kb + brow) * (int64_t)ldb + bcol0 + 4 * bc4;

  f32x4 a0[4], a1[4];
  f32x4 bn = {0.f, 0.f, 0.f, 0.f};
#define EOFX_LOAD_SLAB(areg, chunk)                                                              \
  do {                                                                                           \
    if (b_loader) bn = *reinterpret_cast<const f32x4*>(Bp + (int64_t)(chunk) * ATB_KC * ldb);   \
    const float* pa_ = Ap + (int64_t)(chunk) * ATB_KC * lda;                                     \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) areg[s] =                                      \
        __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pa_ + (int64_t)(4 * s) * lda)); \
  } while (0)
#define EOFX_STORE_B(buf)                                                                        \
  do {                                                                                           \
    if (b_loader) {                                                                              \
      double* d_ = &Bs[buf][brow][4 * bc4];                                                      \
      d_[0] = (double)bn[0]; d_[1] = (double)bn[1]; d_[2] = (double)bn[2]; d_[3] = (double)bn[3]; \
    }                                                                                            \
  } while (0)
#define EOFX_COMPUTE_SLAB(areg, buf)                                                             \
  do {                                                                                           \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                              \
      double b_[NT];                                                                             \
      _Pragma("unroll") for (int q = 0; q < NT; ++q) b_[q] = Bs[buf][4 * s + kq][16 * q + c16];  \
      _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                            \
        const double a_ = (double)areg[s][t];                                                    \
        _Pragma("unroll") for (int q = 0; q < NT; ++q)                                           \
            acc[t][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, b_[q], acc[t][q], 0, 0, 0);     \
      }                                                                                          \
    }                                                                                            \
  } while (0)

  if (nchunks > 0) {
    EOFX_LOAD_SLAB(a0, 0);
    EOFX_STORE_B(0);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
      EOFX_LOAD_SLAB(a1, c + 1);
      __builtin_amdgcn_sched_barrier(0);
      EOFX_COMPUTE_SLAB(a0, 0);
      EOFX_STORE_B(1);
      __syncthreads();
      const int c2 = (c + 2 < nchunks) ? c + 2 : c + 1;
      EOFX_LOAD_SLAB(a0, c2);
      __builtin_amdgcn_sched_barrier(0);
      EOFX_COMPUTE_SLAB(a1, 1);
      EOFX_STORE_B(0);
      __syncthreads();
    }
  }
#undef EOFX_LOAD_SLAB
#undef EOFX_COMPUTE_SLAB
#undef EOFX_STORE_B

  // D layout of the f64 16x16 MFMA: col = lane % 16, row = lane / 16 + 4 * reg; A row i of sub-tile t is column 4 i + t
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = kq + 4 * r;
        const int64_t m = m0 + 4 * i + t;
        const int64_t o = m * ldc + bcol0 + 16 * q + c16;
        if (Cd)
          Cd[(int64_t)blockIdx.y * M * ldc + o] = acc[t][q][r];
        else
          C[o] = (float)acc[t][q][r];
      }
}

// out[i] = (float) sum_s part[s][i] over float64 partials, fixed order
__global__ __launch_bounds__(256) void splitk_reduce_f64_kernel(const double* __restrict__ part, float* __restrict__ out,
                                                                int64_t count, int splits) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    double s = 0.0;
    for (int k = 0; k < splits; ++k) s += part[(int64_t)k * count + i];
    out[i] = (float)s;
  }
}

// out[i] = sum_s part[s][i], fixed order, float64 accumulate.  count4 = elements / 4.
// amax_out (may be null): max |out| by atomicMax on the float bits (order independent) -- the panel maximum the next
// split-fp16 pass scales by, taken where the panel is written instead of by one more read of it.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part,
                                                            float* __restrict__ out,
                                                            int64_t count4, int splits,
                                                            unsigned* __restrict__ amax_out = nullptr) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float mx = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += stride) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int s = 0; s < splits; ++s) {
      const f32x4 v = reinterpret_cast<const f32x4*>(part)[(int64_t)s * count4 + i];
      s0 += v[0];
      s1 += v[1];
      s2 += v[2];
      s3 += v[3];
    }
    f32x4 o = {(float)s0, (float)s1, (float)s2, (float)s3};
    reinterpret_cast<f32x4*>(out)[i] = o;
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
  }
  if (amax_out) amax_commit(mx, amax_out);
}

// ---------------------------------------------------------------------------------
// gram_f64: Gpart[bx][L x L] = sum over this block's rows of P[r,:]^T P[r,:]  (float64)
//   grid = (nbx, nb*nb) where nb = ceil(L/64); each block owns one 64x64 sub-block of G
//   and a strided set of 32-row slabs.  HBM-bound on P (rows x L x 4 B), tiny.
// lower triangle of a symmetric product from its upper one: C[m][l] = C[l][m] for l < m  (64 x 64 tiles through LDS)
__global__ __launch_bounds__(256) void symmetrize_lower_kernel(float* __restrict__ C, int64_t n, int64_t ld) {
  __shared__ float T[64][65];
  const int bi = blockIdx.y, bj = blockIdx.x;      // tile row / column of the LOWER triangle being written
  if (bj > bi) return;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {               // read the mirror tile (bj, bi)
    const int64_t gr = (int64_t)bj * 64 + r, gc = (int64_t)bi * 64 + tx;
    T[r][tx] = (gr < n && gc < n) ? C[gr * ld + gc] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int64_t gr = (int64_t)bi * 64 + r, gc = (int64_t)bj * 64 + tx;
    if (gr < n && gc < n && gc < gr) C[gr * ld + gc] = T[tx][r];
  }
}

// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_f64_kernel(const float* __restrict__ P, int64_t rows,
                                                        int L, double* __restrict__ Gpart) {
  __shared__ __attribute__((aligned(16))) float Pa[32][64];
  __shared__ __attribute__((aligned(16))) float Pb[32][64];
  const int nb = (L + 63) / 64;
  const int bi = blockIdx.y / nb, bj = blockIdx.y % nb;
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;
  double acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = 0.0;
  const int lr = tid >> 4;         // 0..15  (two passes -> 32 rows)
  const int lc = (tid & 15) * 4;   // float4 column
  for (int64_t r0 = (int64_t)blockIdx.x * 32; r0 < rows; r0 += (int64_t)gridDim.x * 32) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int rr = lr + 16 * h;
      const int64_t r = r0 + rr;
      f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      if (r < rows) {
        const int ca = bi * 64 + lc, cb = bj * 64 + lc;
        if (ca < L) va = *reinterpret_cast<const f32x4*>(P + r * L + ca);
        if (cb < L) vb = *reinterpret_cast<const f32x4*>(P + r * L + cb);
      }
      *reinterpret_cast<f32x4*>(&Pa[rr][lc]) = va;
      *reinterpret_cast<f32x4*>(&Pb[rr][lc]) = vb;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&Pa[r][4 * ti]);
      const f32x4 b = *reinterpret_cast<const f32x4*>(&Pb[r][4 * tj]);
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] += (double)a[x] * (double)b[y];
    }
    __syncthreads();
  }
  double* G = Gpart + (int64_t)blockIdx.x * L * L;
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int gi = bi * 64 + 4 * ti + x, gj = bj * 64 + 4 * tj + y;
      if (gi < L && gj < L) G[(int64_t)gi * L + gj] = acc[x][y];
    }
}

// ---------------------------------------------------------------------------------
// gram_mfma: the same float64 Gram matrix on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).
//   grid = (nbx, nb*nb); every workgroup writes one partial of its 64x64 sub-block (its four waves meet in LDS in a fixed
//   order); the nbx partials are summed in a fixed order by f64_reduce_kernel.
//   Per k-step a wave reads 4 rows x 64 columns (lane (c = l % 16, k = l / 16) loads the float4 at
//   P[row + k][64 b + 4 c ..]: one full 256 B row per 16 lanes), converts to float64 and issues the 16
//   products a[qa] x b[qb]: lane c of "column group" q stands for column 4 c + q, so tile (qa, qb) holds
//   G[64 bi + 4 i + qa][64 bj + 4 j + qb].  Products and sums are exact float64 (inputs are float32).
// ---------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void gram_mfma_kernel(const float* __restrict__ P, int64_t rows, int L,
                                                         double* __restrict__ Gpart) {
  const int nb = (L + 63) / 64;
  const int bi = blockIdx.y / nb, bj = blockIdx.y % nb;
  if (bi > bj) return;             // G is symmetric: the sub-block (bj, bi) writes this one as well
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lc = lane & 15, lk = lane >> 4;
  const int ca = 64 * bi + 4 * lc, cb = 64 * bj + 4 * lc;
  const bool same = (bi == bj);
  f64x4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int64_t wstride = (int64_t)gridDim.x * 4 * 4;        // rows covered by one sweep of all waves
  // software pipeline: the loads of the next k-step are in flight while the 16 products of this one issue
  auto fetch = [&](int64_t r0, f32x4& va, f32x4& vb) {
    const int64_t r = r0 + lk;
    va = f32x4{0.f, 0.f, 0.f, 0.f};
    vb = va;
    if (r < rows) {
      if (ca < L) va = *reinterpret_cast<const f32x4*>(P + r * L + ca);
      if (!same && cb < L) vb = *reinterpret_cast<const f32x4*>(P + r * L + cb);
    }
    if (same) vb = va;
  };
  // A k-step is 4 rows (1 KB) against 10 or 16 products of 64 cycles each: the loads run GRAM_PF steps ahead (one step
  // ahead leaves every wave waiting on an HBM round trip per step: 300 us instead of ~130 for a 1M x 64 panel)
  constexpr int GRAM_PF = 4;
  int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 4;
  f32x4 pa[GRAM_PF], pb[GRAM_PF];
#pragma unroll
  for (int d = 0; d < GRAM_PF; ++d) fetch(r0 + d * wstride, pa[d], pb[d]);      // rows >= `rows` load zeros
  for (; r0 < rows; r0 += GRAM_PF * wstride) {
#pragma unroll
    for (int d = 0; d < GRAM_PF; ++d) {
      const f32x4 va = pa[d], vb = pb[d];
      fetch(r0 + (GRAM_PF + d) * wstride, pa[d], pb[d]);
      __builtin_amdgcn_sched_barrier(0);
      if (same) {       // diagonal sub-block: tile (x, y) is the transpose of tile (y, x) -- 10 of the 16 products
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = x; y < 4; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)va[x], (double)vb[y], acc[x][y], 0, 0, 0);
      } else {
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)va[x], (double)vb[y], acc[x][y], 0, 0, 0);
      }
    }
  }
  // The four waves add their tiles into ONE 64 x 64 block in LDS, wave after wave (fixed order), and the workgroup writes
  // a single partial: a quarter of the partial traffic (one partial per WAVE made the write + the reduction as long as
  // the products themselves: 67 MB of partials for a 265 MB panel, profiles/r03_small_kernel_probe.txt).
  __shared__ double Gs[64][65];
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          if (same && x > y) continue;                          // (tile (y, x) writes both triangles)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = 4 * (lk + 4 * q) + x, j = 4 * lc + y; // D[lane / 16 + 4 reg][lane % 16] (measured layout)
            if (w == 0) {
              Gs[i][j] = acc[x][y][q];
              if (same && x < y) Gs[j][i] = acc[x][y][q];
            } else {
              Gs[i][j] += acc[x][y][q];
              if (same && x < y) Gs[j][i] += acc[x][y][q];
            }
          }
        }
    }
    __syncthreads();
  }
  double* G = Gpart + (int64_t)blockIdx.x * (int64_t)L * L;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int i = e >> 6, j = e & 63;
    const int gi = 64 * bi + i, gj = 64 * bj + j;
    if (gi < L && gj < L) {
      G[(int64_t)gi * L + gj] = Gs[i][j];
      if (!same) G[(int64_t)gj * L + gi] = Gs[i][j];
    }
  }
}

__global__ __launch_bounds__(256) void f64_reduce_kernel(const double* __restrict__ part,
                                                         double* __restrict__ out, int64_t count,
                                                         int nparts) {
  // 64 outputs per workgroup, four threads per output: thread (i, q) sums the partials b = q, q+4, ...;
  // the four sums are combined in a fixed order (deterministic)
  __shared__ double red[4][64];
  const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + li;
  double s = 0;
  if (i < count) {
#pragma unroll 4
    for (int b = q; b < nparts; b += 4) s += part[(int64_t)b * count + i];
  }
  red[q][li] = s;
  __syncthreads();
  if (q == 0 && i < count) out[i] = ((red[0][li] + red[1][li]) + red[2][li]) + red[3][li];
}

// ---------------------------------------------------------------------------------
// xgram_mfma (round 5): the float64 cross-Gram matrix C = Pa^T Pb of two row-major float32 panels with their own leading
// dimensions (Pa: La columns, Pb: Lb columns; both multiples of 64) on the fp64 matrix cores -- the projection
// coefficients K^H W and the columns of the Rayleigh-Ritz matrix of the complex block-Krylov decomposition.
//   grid = (nbx, (La / 64) * (Lb / 64)); same wave layout, prefetch depth and fixed-order wave merge as gram_mfma_kernel
//   (its bi != bj case); partial b of sub-block (bi, bj) lands in part[b][La x Lb]; f64_reduce_kernel sums the partials.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xgram_mfma_kernel(const float* __restrict__ Pa, int64_t lda, const float* __restrict__ Pb,
                                                          int64_t ldb, int64_t rows, int La, int Lb, double* __restrict__ part) {
  const int nbj = Lb / 64;
  const int bi = blockIdx.y / nbj, bj = blockIdx.y % nbj;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lc = lane & 15, lk = lane >> 4;
  const int ca = 64 * bi + 4 * lc, cb = 64 * bj + 4 * lc;
  f64x4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int64_t wstride = (int64_t)gridDim.x * 4 * 4;
  auto fetch = [&](int64_t r0, f32x4& va, f32x4& vb) {
    const int64_t r = r0 + lk;
    va = f32x4{0.f, 0.f, 0.f, 0.f};
    vb = va;
    if (r < rows) {
      va = *reinterpret_cast<const f32x4*>(Pa + r * lda + ca);
      vb = *reinterpret_cast<const f32x4*>(Pb + r * ldb + cb);
    }
  };
  constexpr int XG_PF = 4;
  int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 4;
  f32x4 pa[XG_PF], pb[XG_PF];
#pragma unroll
  for (int d = 0; d < XG_PF; ++d) fetch(r0 + d * wstride, pa[d], pb[d]);
  for (; r0 < rows; r0 += XG_PF * wstride) {
#pragma unroll
    for (int d = 0; d < XG_PF; ++d) {
      const f32x4 va = pa[d], vb = pb[d];
      fetch(r0 + (XG_PF + d) * wstride, pa[d], pb[d]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)va[x], (double)vb[y], acc[x][y], 0, 0, 0);
    }
  }
  __shared__ double Gs[64][65];
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = 4 * (lk + 4 * q) + x, j = 4 * lc + y;    // D[lane / 16 + 4 reg][lane % 16], as in gram_mfma_kernel
            if (w == 0) Gs[i][j] = acc[x][y][q];
            else Gs[i][j] += acc[x][y][q];
          }
    }
    __syncthreads();
  }
  double* G = part + (int64_t)blockIdx.x * (int64_t)La * Lb;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int i = e >> 6, j = e & 63;
    G[(int64_t)(64 * bi + i) * Lb + 64 * bj + j] = Gs[i][j];
  }
}

// The real cross-Gram matrix C [(nb LP) x LP] of complex panels held as [Re(h) | Im(h)] (LP = 2 h) -> the real matrix
// E [(nb LP) x LP] with [Kr | Ki] E = [Re(K c) | Im(K c)] for the complex coefficients c = K^H W of every block:
//   c = (rr + ii) + i (ri - ir),   E_b = [[Re c, Im c], [-Im c, Re c]]
__global__ __launch_bounds__(256) void cproj_embed_kernel(const double* __restrict__ C, int nb, int LP, double* __restrict__ E) {
  const int h = LP / 2;
  const int64_t total = (int64_t)nb * h * h;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(t / (h * h)), i = (int)(t / h % h), j = (int)(t % h);
    const double* Cb = C + (int64_t)b * LP * LP;
    double* Eb = E + (int64_t)b * LP * LP;
    const double rr = Cb[i * LP + j], ii = Cb[(h + i) * LP + h + j], ri = Cb[i * LP + h + j], ir = Cb[(h + i) * LP + j];
    const double cr = rr + ii, ci = ri - ir;
    Eb[i * LP + j] = cr;
    Eb[(h + i) * LP + j] = -ci;
    Eb[i * LP + h + j] = ci;
    Eb[(h + i) * LP + h + j] = cr;
  }
}

// ---------------------------------------------------------------------------------
// chol_rinv: single workgroup.  G (L x L float64, leading l x l block used) -> Rinv with
// G = R^T R, Rinv = R^-1 (upper triangular), zero outside l x l.  A pivot that falls below
// tol * G[j][j] marks column j as linearly dependent: its Q column becomes exactly zero.
// l <= 64.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double rl_f64(double v, int lane) {   // broadcast of lane `lane` (uniform index)
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// One workgroup, matrix in LDS, plain loops (a fully unrolled register-resident variant is ~300 KB of
// straight-line code and runs out of the instruction cache: 390 us).
//   phase 1: blocked right-looking Cholesky (upper form A = R^T R), 8 rows per block: wave 0 factors the
//            8-row panel in registers (v_readlane broadcasts, no barriers), then all four waves apply the rank-8 update to
//            the trailing rows -- two barriers per block instead of two per column;
//   phase 2: X = R^-1 by 16 x 16 blocks: the four diagonal blocks by back substitution (wave 0, a column per lane in
//            registers), then the blocks above the diagonal as small products, one block diagonal at a time
//            (six barriers).  Round 4: 64 dependent steps took 29 of the kernel's 64 us (tools/probes/rinv_phase_probe.hip).
// Rows / columns >= l are padded with the identity.  All sums run in a fixed order (deterministic).
// d0_ext (optional): the ORIGINAL diagonal the dependency rule refers to, when G is a diagonal block of a blocked
// factorisation (its own diagonal is a Schur complement by then); G / Rinv may then point into a larger matrix of leading
// dimension L.
__global__ __launch_bounds__(256) void chol_rinv_kernel(const double* __restrict__ G, int L, int l,
                                                         double* __restrict__ Rinv, double tol,
                                                         const double* __restrict__ d0_ext = nullptr,
                                                         unsigned long long* __restrict__ prof = nullptr) {
#define EOFX_RINV_T(i) do { if (prof && threadIdx.x == 0) prof[i] = wall_clock64(); } while (0)   /* tools/probes/rinv_phase_probe.hip */
  EOFX_RINV_T(0);
  __shared__ double A[64][65];     // upper triangle: R[r][c], r < c (the diagonal is kept as 1/R[j][j] in pivs)
  __shared__ double X[64][65];     // R^-1 (upper triangle)
  __shared__ double d0s[64], pivs[64];
  __shared__ double Tm[3][16][17];   // phase 2b: the inner sums of one block diagonal
  __shared__ int dead[64];
  const int tid = threadIdx.x;
  for (int i = tid; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    double v = (r == c) ? 1.0 : 0.0;
    if (r <= c && c < l) v = G[(int64_t)r * L + c];
    A[r][c] = v;
  }
  if (tid < 64) {
    d0s[tid] = (tid < l) ? (d0_ext ? d0_ext[tid] : G[(int64_t)tid * L + tid]) : 1.0;
    pivs[tid] = 1.0;
    dead[tid] = 0;
  }
  __syncthreads();
  EOFX_RINV_T(1);
  // rows / columns >= l are identity padding: factoring them is harmless (pivot 1, zero row), so the panels are always
  // full -- 8 rows, no bounds inside: straight-line code the scheduler can overlap across columns
  for (int jb = 0; jb < l; jb += 8) {
    const int je = jb + 8;
    unsigned long long tp0 = prof ? wall_clock64() : 0;
    if (tid < 64) {   // wave 0 factors rows jb .. je-1 of R in registers: lane c holds a[u] = A[jb+u][c];
                      // values cross lanes with v_readlane (uniform lane index), no LDS round trips
      const int c = tid;
      double a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = A[jb + u][c];
      const double d0c = d0s[c];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = jb + u;
        const double d = rl_f64(a[u], j);             // A[j][j] after the updates of all earlier rows
        const double d0 = rl_f64(d0c, j);
        const bool dj = !(d > tol * d0) || !(d0 > 0.0);   // numerically dependent column
        double piv = __builtin_amdgcn_rsq(dj ? 1.0 : d);  // 1/sqrt(d): estimate + two Newton steps
        piv = piv * (1.5 - 0.5 * d * piv * piv);
        piv = piv * (1.5 - 0.5 * d * piv * piv);
        if (dj) piv = 0.0;
        if (c == j) {
          dead[j] = dj;
          pivs[j] = dj ? 1.0 : piv;
        }
        a[u] = (c > j) ? a[u] * piv : 0.0;            // R[j][c]
#pragma unroll
        for (int u2 = u + 1; u2 < 8; ++u2) {
          const double rjr = rl_f64(a[u], jb + u2);   // R[j][r] lives in lane r
          if (c >= jb + u2) a[u2] -= rjr * a[u];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c > jb + u) A[jb + u][c] = a[u];
    }
    __syncthreads();
    if (prof && tid == 0) prof[5] += wall_clock64() - tp0;
    {                                                  // rank-8 update of the trailing rows r >= je: every wave walks the
                                                       // same number of rows (uniform loop, unrolled: the LDS round trips of
                                                       // four rows overlap); entries below the diagonal are not stored
      const int c = tid & 63;
      double rc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) rc[u] = A[jb + u][c];
#pragma unroll 4
      for (int r = je + (tid >> 6); r < 64; r += 4) {
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += A[jb + u][r] * rc[u];
        if (r <= c) A[r][c] -= s;
      }
    }
    __syncthreads();
  }
  EOFX_RINV_T(2);
  if (tid < 64) {   // phase 2a: the four 16 x 16 diagonal blocks of X = R^-1 by back substitution, all in wave 0: lane (b, cc)
                    // keeps column cc of block b in registers (fully unrolled: static indices), R comes from LDS as
                    // broadcasts -- no shuffles, no fences; entries below the diagonal come out as exact zeros (2b reads them)
    const int b16 = (tid >> 4) * 16, cc = tid & 15;
    double x[16];
#pragma unroll
    for (int r = 15; r >= 0; --r) {
      double s = 0.0;
#pragma unroll
      for (int t = r + 1; t < 16; ++t) s += A[b16 + r][b16 + t] * x[t];
      x[r] = (r <= cc) ? (((r == cc) ? 1.0 : 0.0) - s) * pivs[b16 + r] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) X[b16 + r][b16 + cc] = x[r];
  }
  __syncthreads();
  {   // phase 2b: the blocks above the diagonal, by distance d from it: X_ij = -X_ii (sum_{k = i+1 .. j} R_ik X_kj), every
      // block a 16 x 16 x 16 product per term with one output per thread -- 64 dependent steps become 16 + six products
    const int rr = tid >> 4, cc = tid & 15;
    for (int d = 1; d < 4; ++d) {
      for (int i = 0; i + d < 4; ++i) {
        const int j = i + d;
        double t = 0.0;
        for (int k = i + 1; k <= j; ++k)
#pragma unroll 16
          for (int m = 0; m < 16; ++m) t += A[16 * i + rr][16 * k + m] * X[16 * k + m][16 * j + cc];
        Tm[i][rr][cc] = t;
      }
      __syncthreads();
      for (int i = 0; i + d < 4; ++i) {
        const int j = i + d;
        double x = 0.0;
#pragma unroll 16
        for (int m = 0; m < 16; ++m) x += X[16 * i + rr][16 * i + m] * Tm[i][m][cc];   // X_ii is upper triangular, zeros stored
        X[16 * i + rr][16 * j + cc] = -x;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  EOFX_RINV_T(3);
  for (int i = tid; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    if (r < L && c < L) Rinv[(int64_t)r * L + c] = (r <= c && c < l && !dead[c]) ? X[r][c] : 0.0;
  }
  EOFX_RINV_T(4);
#undef EOFX_RINV_T
}

// ---------------------------------------------------------------------------------
// dgemm64: C[M x N] = alpha op(A) B + beta C in float64, row-major, M, N, K multiples of 64 -- the small dense products of
// the blocked Cholesky factorisation / triangular inverse of wide sketches (launch_rinv for l > 64: the PCA pre-reduction's
// 1536-column panels).  One 64 x 64 tile of C per workgroup (16 x 16 threads, 4 x 4 outputs each), K in steps of 16
// through LDS.  TA: op(A) = A^T with A stored [K x M].  upper: tiles strictly below the block diagonal are skipped.
// C may alias B when K == 64 and the tile of B a workgroup reads is the tile of C it writes (the row-panel solve).
// blockIdx.z: batch index, operands advance by sA / sB / sC elements per problem.
// ---------------------------------------------------------------------------------
template <bool TA>
__global__ __launch_bounds__(256) void dgemm64_kernel(const double* __restrict__ A, int lda, const double* B, int ldb,
                                                       double* C, int ldc, int K, double alpha, double beta, int upper,
                                                       int64_t sA = 0, int64_t sB = 0, int64_t sC = 0) {
  __shared__ double As[16][65], Bs[16][65];
  const int bx = blockIdx.x, by = blockIdx.y;
  if (upper && by > bx) return;
  A += (int64_t)blockIdx.z * sA;      // batched form (the level-wise triangular inverse): element strides between the problems
  B += (int64_t)blockIdx.z * sB;
  C += (int64_t)blockIdx.z * sC;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  for (int k0 = 0; k0 < K; k0 += 16) {
    // stage op(A)[64 by .. +64][k0 .. +16] as As[k][m] and B[k0 .. +16][64 bx .. +64] as Bs[k][n]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = threadIdx.x + 256 * e;     // 0 .. 1023
      if (TA) {
        const int k = idx >> 6, m = idx & 63;
        As[k][m] = A[(int64_t)(k0 + k) * lda + 64 * by + m];
      } else {
        const int m = idx >> 4, k = idx & 15;
        As[k][m] = A[(int64_t)(64 * by + m) * lda + k0 + k];
      }
      const int kb = idx >> 6, n = idx & 63;
      Bs[kb][n] = B[(int64_t)(k0 + kb) * ldb + 64 * bx + n];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[k][4 * ty + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[k][4 * tx + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fma(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double* c = C + (int64_t)(64 * by + 4 * ty + i) * ldc + 64 * bx + 4 * tx + j;
      *c = beta == 0.0 ? alpha * acc[i][j] : alpha * acc[i][j] + beta * *c;
    }
}

// S <- [leading l x l block of G, identity beyond] as an Lb x Lb matrix; d0 <- its diagonal (the dependency rule's reference)
__global__ void chol_blocked_init_kernel(const double* __restrict__ G, int L, int l, double* __restrict__ S, int Lb,
                                          double* __restrict__ d0) {
  const int64_t total = (int64_t)Lb * Lb;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / Lb), c = (int)(i % Lb);
    double v = (r == c) ? 1.0 : 0.0;
    if (r < l && c < l) v = G[(int64_t)r * L + c];
    S[i] = v;
    if (r == c) d0[r] = v;
  }
}
// Rinv (L x L) <- leading l x l block of X (Lb x Lb, upper triangular), zero elsewhere
__global__ void chol_blocked_export_kernel(const double* __restrict__ X, int Lb, int l, double* __restrict__ Rinv, int L) {
  const int64_t total = (int64_t)L * L;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / L), c = (int)(i % L);
    Rinv[i] = (r <= c && c < l) ? X[(int64_t)r * Lb + c] : 0.0;
  }
}

// ---------------------------------------------------------------------------------
// panel_matmul: out[rows x Lo] = P[rows x L] * Mx[L x Lo]  (Mx float64; exact float64 products of the float32 panel
// entries, float64 sums) on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).  out must not alias P.
//   grid = (G, ceil(Lo / 64)); block = 4 waves.  L <= KW (256): the 64-column slice of Mx sits in LDS for the whole
//   launch (dynamic LDS: KW x 66 doubles); every wave walks 32-row groups g = wave id, wave id + 4 G, ... and, inside a
//   group, the K axis in chunks of 64, with the NEXT chunk's panel entries already in flight (register double buffer).
//   Wider inner dimensions (the PCA route's Rayleigh-Ritz panels): one group per wave, Mx passes through LDS in windows.
//   A: lane (i = l % 16, kq = l / 16) loads the float4 P[row i][kc + 16 tt + 4 kq ..] straight into registers;
//      element c of it is simply NAMED k index kq of MFMA step (tt, c), i.e. actual k = kc + 16 tt + 4 kq + c, and the
//      B fragment is read from LDS under the same naming: Ms[kc + 16 tt + 4 kq + c][16 q + i].  No LDS trip for P.
//   D[kq + 4 r][i] of tile (t, q) (measured layout) = out[row0 + 16 t + kq + 4 r][c0 + 16 q + i].
// ---------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int PMM_LD = 66;   // doubles per staged row of Mx

// GEN (round 5, the block-Krylov steps of eofx_rsvd_c64): the inner dimension is a list of 64-column chunks, chunk c at
// P + (c / cpb) * slab + (c % cpb) * 64 with row stride ldp (a wide row-major panel: cpb = all its chunks, ldp = its width; a
// stack of separate panels of cpb chunks each: slab = rows_pad * 64 cpb, ldp = 64 cpb), and with `sub` the kernel writes
// sub - P Mx (a projection step).
template <bool GEN = false>
__global__ __launch_bounds__(256) void panel_matmul_kernel(const float* __restrict__ P,
                                                           int64_t rows, int L,
                                                           const double* __restrict__ Mx, int Lo,
                                                           float* __restrict__ out, int KW,
                                                           unsigned* __restrict__ amax_out = nullptr,
                                                           int64_t ldp = 0, int64_t slab = 0, int cpb = 1,
                                                           const float* __restrict__ sub = nullptr, int upper = 0) {
  // upper != 0: Mx is upper triangular (the R^-1 of a Cholesky-QR): output column block b needs the K chunks 0 .. b only --
  // half the products of a wide panel; the skipped ones are exact zeros, the result is the same bits
  extern __shared__ __attribute__((aligned(16))) double Ms[];
  float amx = 0.f;   // [KW][PMM_LD]: KW = multiple of 64, the K window in LDS
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int by = upper ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;    // (triangular: the long column blocks start first)
  const int c0 = by * 64;
  const int Lc_all = (L + 63) / 64;
  const int Lc = upper ? (Lc_all < by + 1 ? Lc_all : by + 1) : Lc_all;   // K chunks of 64 that matter
  const int Le = L < 64 * Lc ? L : 64 * Lc;                                                         // ... = inner dimension used
  const int64_t ngroups = (rows + 31) / 32;
  const int64_t g0 = (int64_t)blockIdx.x * 4 + wave;
  auto stage = [&](int k0) {             // rows [k0, k0 + KW) of the 64-column slice of Mx
    for (int i = tid; i < KW * 64; i += 256) {
      const int r = i >> 6, c = i & 63;
      Ms[r * PMM_LD + c] = (k0 + r < L && c0 + c < Lo) ? Mx[(int64_t)(k0 + r) * Lo + c0 + c] : 0.0;
    }
  };
  auto fetch = [&](bool valid, int64_t g, int kc, f32x4 (&a)[2][4]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t r = 32 * g + 16 * t + li;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int k = kc + 16 * tt + 4 * lk;
        if constexpr (GEN)
          a[t][tt] = (valid && r < rows && k < L) ? *reinterpret_cast<const f32x4*>(P + (int64_t)((k >> 6) / cpb) * slab + ((k >> 6) % cpb) * 64 + r * ldp + (k & 63))
                                                 : f32x4{0.f, 0.f, 0.f, 0.f};
        else
          a[t][tt] = (valid && r < rows && k < L) ? *reinterpret_cast<const f32x4*>(P + r * L + k)
                                                 : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  f64x4 acc[2][4];
  // one K chunk of one 32-row group: 32 MFMA steps x 8 tiles; ms0 = first row of Mx held in LDS
  auto compute = [&](int64_t g, int kc, int ms0, const f32x4 (&a)[2][4]) {
    if (kc == 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = f64x4{0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = Ms[(kc - ms0 + 16 * tt + 4 * lk + c) * PMM_LD + 16 * q + li];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[t][q] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[t][tt][c], b[q], acc[t][q], 0, 0, 0);
      }
    if (kc + 64 >= Le) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = 32 * g + 16 * t + lk + 4 * r;
          if (row >= rows) continue;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = c0 + 16 * q + li;
            if (col < Lo) {
              float o_ = (float)acc[t][q][r];
              if constexpr (GEN) {
                if (sub) o_ = (float)((double)sub[row * Lo + col] - acc[t][q][r]);
              }
              out[row * Lo + col] = o_;
              amx = fmaxf(amx, fabsf(o_));
            }
          }
        }
    }
  };
  f32x4 a0[2][4], a1[2][4];
  if (Lc * 64 <= KW) {
    // the whole slice fits: staged once, every wave walks groups g0, g0 + 4 G, ... with the next (group, chunk) in flight
    stage(0);
    __syncthreads();
    const int64_t gstride = (int64_t)gridDim.x * 4;
    const int64_t nitems = g0 < ngroups ? ((ngroups - g0 + gstride - 1) / gstride) * Lc : 0;
    fetch(nitems > 0, g0, 0, a0);
    for (int64_t item = 0; item < nitems; item += 2) {
      fetch(item + 1 < nitems, g0 + ((item + 1) / Lc) * gstride, (int)((item + 1) % Lc) * 64, a1);
      __builtin_amdgcn_sched_barrier(0);
      compute(g0 + (item / Lc) * gstride, (int)(item % Lc) * 64, 0, a0);
      if (item + 1 < nitems) {
        fetch(item + 2 < nitems, g0 + ((item + 2) / Lc) * gstride, (int)((item + 2) % Lc) * 64, a0);
        __builtin_amdgcn_sched_barrier(0);
        compute(g0 + ((item + 1) / Lc) * gstride, (int)((item + 1) % Lc) * 64, 0, a1);
      }
    }
  } else {
    // wide inner dimension (the launcher gives every wave at most ONE group): windows of KW rows of Mx pass through LDS
    const bool active = g0 < ngroups;
    const int wc = KW / 64;
    fetch(active, g0, 0, a0);
    for (int w0 = 0; w0 < Lc; w0 += wc) {
      __syncthreads();
      stage(64 * w0);
      __syncthreads();
      const int we = w0 + wc < Lc ? w0 + wc : Lc;
      for (int ch = w0; ch < we; ch += 2) {        // wc is even
        fetch(active && ch + 1 < Lc, g0, 64 * (ch + 1), a1);
        __builtin_amdgcn_sched_barrier(0);
        if (active) compute(g0, 64 * ch, 64 * w0, a0);
        fetch(active && ch + 2 < Lc, g0, 64 * (ch + 2), a0);
        __builtin_amdgcn_sched_barrier(0);
        if (active && ch + 1 < we) compute(g0, 64 * (ch + 1), 64 * w0, a1);
      }
    }
  }
  if (amax_out) amax_commit(amx, amax_out);
}

// per-column max/min over rows [0, rows): partial per block, then a second tiny pass.
// thread (cq, rl): float4 of columns 64 by + 4 cq .., rows rl, rl + 16, ... (one 256 B row per 16 lanes);
// eight independent loads in flight per thread
__global__ __launch_bounds__(256) void colminmax_part_kernel(const float* __restrict__ P,
                                                             int64_t rows, int L,
                                                             float* __restrict__ pmx,
                                                             float* __restrict__ pmn) {
  __shared__ float smx[16][64], smn[16][64];
  const int tid = threadIdx.x;
  const int cq = tid & 15, rl = tid >> 4;
  const int c = blockIdx.y * 64 + 4 * cq;
  f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, mn = {INFINITY, INFINITY, INFINITY, INFINITY};
  if (c < L) {
    const int64_t step = (int64_t)gridDim.x * 16;
    int64_t r = (int64_t)blockIdx.x * 16 + rl;
    for (; r + 7 * step < rows; r += 8 * step) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(P + (r + u * step) * L + c);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          mx[e] = fmaxf(mx[e], v[u][e]);
          mn[e] = fminf(mn[e], v[u][e]);
        }
    }
    for (; r < rows; r += step) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(P + r * L + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mx[e] = fmaxf(mx[e], v[e]);
        mn[e] = fminf(mn[e], v[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    smx[rl][4 * cq + e] = mx[e];
    smn[rl][4 * cq + e] = mn[e];
  }
  __syncthreads();
  if (tid < 64 && blockIdx.y * 64 + tid < L) {
    float a = smx[0][tid], b = smn[0][tid];
    for (int q = 1; q < 16; ++q) {
      a = fmaxf(a, smx[q][tid]);
      b = fminf(b, smn[q][tid]);
    }
    pmx[(int64_t)blockIdx.x * L + blockIdx.y * 64 + tid] = a;
    pmn[(int64_t)blockIdx.x * L + blockIdx.y * 64 + tid] = b;
  }
}
// 64 columns per workgroup, four threads per column (max / min are order independent)
__global__ __launch_bounds__(256) void colminmax_final_kernel(const float* __restrict__ pmx,
                                                              const float* __restrict__ pmn, int nparts, int L,
                                                              float* __restrict__ mx, float* __restrict__ mn) {
  __shared__ float sa[4][64], sb[4][64];
  const int li = threadIdx.x & 63, qq = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + li;
  float a = -INFINITY, b = INFINITY;
  if (c < L) {
#pragma unroll 8
    for (int q = qq; q < nparts; q += 4) {
      a = fmaxf(a, pmx[(int64_t)q * L + c]);
      b = fminf(b, pmn[(int64_t)q * L + c]);
    }
  }
  sa[qq][li] = a;
  sb[qq][li] = b;
  __syncthreads();
  if (qq == 0 && c < L) {
    mx[c] = fmaxf(fmaxf(sa[0][li], sa[1][li]), fmaxf(sa[2][li], sa[3][li]));
    mn[c] = fminf(fminf(sb[0][li], sb[1][li]), fminf(sb[2][li], sb[3][li]));
  }
}

// dst[r*k + c] = P[r*L + c] * sign[c]   (dense export, drops padding)
__global__ __launch_bounds__(256) void panel_export_kernel(const float* __restrict__ P,
                                                           int64_t rows, int L, int k,
                                                           const double* __restrict__ sign,
                                                           float* __restrict__ dst) {
  const int64_t total = rows * k;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / k;
    const int c = (int)(i - r * k);
    float v = P[r * L + c];
    if (sign) v *= (float)sign[c];
    dst[i] = v;
  }
}
// P[rows_pad x L] <- src[rows x l] zero padded.  One thread per four adjacent columns (one 16-byte store); 32-bit
// index arithmetic whenever the panel has fewer than 2^31 quads (a 64-bit division per element cost 100 us on the
// 10000 x 60 sketch).
__global__ __launch_bounds__(256) void panel_import_kernel(const float* __restrict__ src,
                                                           int64_t rows, int l,
                                                           float* __restrict__ P, int64_t rows_pad,
                                                           int L, unsigned* __restrict__ amax_out = nullptr) {
  const int l4 = L / 4;
  const int64_t total = rows_pad * l4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool small = total < ((int64_t)1 << 31);
  float mx = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int64_t r;
    int c;
    if (small) {
      const unsigned iu = (unsigned)i, ru = iu / (unsigned)l4;
      r = ru;
      c = (int)(iu - ru * (unsigned)l4) * 4;
    } else {
      r = i / l4;
      c = (int)(i - r * l4) * 4;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      const float* sp = src + r * l + c;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < l) v[e] = sp[e];
    }
    reinterpret_cast<f32x4*>(P)[i] = v;
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
  if (amax_out) amax_commit(mx, amax_out);
}

// ---------------------------------------------------------------------------------
// Fused preprocessor (Scaler + Sanitizer), HBM-bound.
//   colstats : NaN-aware count / sum / sum of squares per feature in float64,
//              one thread per feature column, rows split over gridDim.y.
//   finalize : combine row-splits in fixed order -> mean, std, shift, scale, variance term.
//   rowcount : valid-feature count per sample (only launched when samples are missing).
//   apply    : gather valid rows/cols, (x - shift) * scale, write X and X^T zero padded.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ X, int64_t n,
                                                        int64_t P, int64_t ld,
                                                        const int64_t* __restrict__ row_map,
                                                        int64_t rows_per_split,
                                                        int* __restrict__ cnt,
                                                        double* __restrict__ sum,
                                                        double* __restrict__ sumsq,
                                                        float* __restrict__ vmin,
                                                        float* __restrict__ vmax) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= P) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
  const int64_t r1 = (r0 + rows_per_split < n) ? r0 + rows_per_split : n;
  int k = 0;
  double s = 0.0, q = 0.0;
  float lo = INFINITY, hi = -INFINITY;   // fminf/fmaxf skip NaN
  // row_map (bootstrap resampling: source row of every logical row) is wave-uniform -> scalar loads
  int64_t r = r0;
  for (; r + 8 <= r1; r += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t rr = row_map ? row_map[r + u] : r + u;
      v[u] = __builtin_nontemporal_load(X + rr * ld + c);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      lo = fminf(lo, v[u]);
      hi = fmaxf(hi, v[u]);
      if (v[u] == v[u]) {
        const double d = (double)v[u];
        ++k;
        s += d;
        q += d * d;
      }
    }
  }
  for (; r < r1; ++r) {
    const int64_t rr = row_map ? row_map[r] : r;
    const float v = X[rr * ld + c];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
    if (v == v) {
      const double d = (double)v;
      ++k;
      s += d;
      q += d * d;
    }
  }
  const int64_t o = (int64_t)blockIdx.y * P + c;
  cnt[o] = k;
  sum[o] = s;
  sumsq[o] = q;
  vmin[o] = lo;
  vmax[o] = hi;
}

// The same statistics with FOUR adjacent features per thread (16-byte loads: one instruction of a wave covers 1 KiB of a
// row instead of 256 B).  P % 4 == 0, ld % 4 == 0, X 16-byte aligned.  Rows are summed in the same order per split.
__global__ __launch_bounds__(256) void colstats4_kernel(const float* __restrict__ X, int64_t n, int64_t P, int64_t ld,
                                                         const int64_t* __restrict__ row_map, int64_t rows_per_split,
                                                         int* __restrict__ cnt, double* __restrict__ sum,
                                                         double* __restrict__ sumsq, float* __restrict__ vmin,
                                                         float* __restrict__ vmax) {
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= P) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
  const int64_t r1 = (r0 + rows_per_split < n) ? r0 + rows_per_split : n;
  int k[4] = {0, 0, 0, 0};
  double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
  float lo[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, hi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  auto take = [&](const f32x4& v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lo[e] = fminf(lo[e], v[e]);      // fminf / fmaxf skip NaN
      hi[e] = fmaxf(hi[e], v[e]);
      if (v[e] == v[e]) {
        const double d = (double)v[e];
        ++k[e];
        s[e] += d;
        q[e] += d * d;
      }
    }
  };
  int64_t r = r0;
  for (; r + 8 <= r1; r += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t rr = row_map ? row_map[r + u] : r + u;      // wave-uniform -> scalar loads
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(X + rr * ld + c));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) take(v[u]);
  }
  for (; r < r1; ++r) {
    const int64_t rr = row_map ? row_map[r] : r;
    take(*reinterpret_cast<const f32x4*>(X + rr * ld + c));
  }
  const int64_t o = (int64_t)blockIdx.y * P + c;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    cnt[o + e] = k[e];
    sum[o + e] = s[e];
    sumsq[o + e] = q[e];
    vmin[o + e] = lo[e];
    vmax[o + e] = hi[e];
  }
}

// The column statistics of colstats4_kernel + the sample-contiguous RAW field on the way (round 5: the Hilbert stage of an
// in-place matrix needs whole series per feature and paid a 14 ms transposing copy behind the 5.5 ms statistics pass).
// The tiling of apply_kernel<true>: a workgroup walks a strip of 64 features down its row split in 64 x 64 tiles -- thread
// (tq = t % 16, tr = t / 16) loads the float4 of features 4 tq .. of rows tr + 16 q, takes them into its statistics, stages
// them in LDS and writes 4 consecutive samples of one feature (256 contiguous bytes per feature and wave instruction) into
// Xt [P][n_pad].  At the end the 16 row groups of a feature meet in LDS in a fixed order: one partial per (split, feature),
// as colstats4_kernel leaves them.  P % 4 == 0, rows_per_split % 64 == 0.  The values written are the RAW ones.
__global__ __launch_bounds__(256) void colstats_tr_kernel(const float* __restrict__ X, int64_t n, int64_t P, int64_t ld,
                                                           int64_t rows_per_split, int* __restrict__ cnt, double* __restrict__ sum,
                                                           double* __restrict__ sumsq, float* __restrict__ vmin,
                                                           float* __restrict__ vmax, float* __restrict__ Xt, int64_t n_pad) {
  __shared__ float T[64][65];
  __shared__ double Sd[16][64];
  __shared__ float Sf[16][64];
  const int tid = threadIdx.x, tq = tid & 15, tr = tid >> 4;
  const int64_t c0 = (int64_t)blockIdx.x * 64, cb = c0 + 4 * tq;
  const bool live = cb < P;
  const int64_t rs0 = (int64_t)blockIdx.y * rows_per_split;
  const int64_t rs1 = (rs0 + rows_per_split < n) ? rs0 + rows_per_split : n;
  int k[4] = {0, 0, 0, 0};
  double s[4] = {0.0, 0.0, 0.0, 0.0}, qq[4] = {0.0, 0.0, 0.0, 0.0};
  float lo[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, hi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  for (int64_t r0 = rs0; r0 < rs1; r0 += 64) {
    f32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t r = r0 + tr + 16 * q;
      v[q] = (live && r < rs1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(X + r * ld + cb)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool in = live && r0 + tr + 16 * q < rs1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = v[q][e];
        T[tr + 16 * q][4 * tq + e] = x;
        if (in) {
          lo[e] = fminf(lo[e], x);      // fminf / fmaxf skip NaN
          hi[e] = fmaxf(hi[e], x);
          if (x == x) {
            const double d = (double)x;
            ++k[e];
            s[e] += d;
            qq[e] += d * d;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = tr + 16 * q;
      if (c0 + cc < P) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = T[4 * tq + e][cc];
        *reinterpret_cast<f32x4*>(Xt + (c0 + cc) * n_pad + r0 + 4 * tq) = o;
      }
    }
    __syncthreads();
  }
  // the 16 row groups of every feature, added in the order tr = 0 .. 15
  const int64_t o = (int64_t)blockIdx.y * P + c0 + tid;
  const bool writer = tid < 64 && c0 + tid < P;
#define EOFX_CTR_RED_D(arr, dst)                                         \
  {                                                                      \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) Sd[tr][4 * tq + e] = arr[e]; \
    __syncthreads();                                                     \
    if (writer) {                                                        \
      double a = Sd[0][tid];                                             \
      for (int g = 1; g < 16; ++g) a += Sd[g][tid];                      \
      dst[o] = a;                                                        \
    }                                                                    \
    __syncthreads();                                                     \
  }
  EOFX_CTR_RED_D(s, sum)
  EOFX_CTR_RED_D(qq, sumsq)
#undef EOFX_CTR_RED_D
#pragma unroll
  for (int e = 0; e < 4; ++e) Sf[tr][4 * tq + e] = lo[e];
  __syncthreads();
  if (writer) {
    float a = Sf[0][tid];
    for (int g = 1; g < 16; ++g) a = fminf(a, Sf[g][tid]);
    vmin[o] = a;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) Sf[tr][4 * tq + e] = hi[e];
  __syncthreads();
  if (writer) {
    float a = Sf[0][tid];
    for (int g = 1; g < 16; ++g) a = fmaxf(a, Sf[g][tid]);
    vmax[o] = a;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) Sf[tr][4 * tq + e] = __int_as_float(k[e]);
  __syncthreads();
  if (writer) {
    int a = 0;
    for (int g = 0; g < 16; ++g) a += __float_as_int(Sf[g][tid]);
    cnt[o] = a;
  }
}

// One thread per feature.  weights may be null (ones).  Outputs per (uncompacted) feature.
__global__ __launch_bounds__(256) void colstats_finalize_kernel(
    const int* __restrict__ cnt_p, const double* __restrict__ sum_p,
    const double* __restrict__ sumsq_p, const float* __restrict__ vmin_p,
    const float* __restrict__ vmax_p, int splits, int64_t P, int center, int standardize,
    const double* __restrict__ weights, double eps, int* __restrict__ cnt, double* __restrict__ mean,
    double* __restrict__ stdv, double* __restrict__ shift, double* __restrict__ scale,
    double* __restrict__ m2, unsigned* __restrict__ absmax, float* __restrict__ vmin_out = nullptr,
    float* __restrict__ vmax_out = nullptr) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float amax = 0.f;
  if (c < P) {
  int k = 0;
  double s = 0.0, q = 0.0;
  float lo = INFINITY, hi = -INFINITY;
  for (int sp = 0; sp < splits; ++sp) {
    k += cnt_p[(int64_t)sp * P + c];
    s += sum_p[(int64_t)sp * P + c];
    q += sumsq_p[(int64_t)sp * P + c];
    lo = fminf(lo, vmin_p[(int64_t)sp * P + c]);
    hi = fmaxf(hi, vmax_p[(int64_t)sp * P + c]);
  }
  cnt[c] = k;
  if (vmin_out) {       // kept for eofx_apply_f32: the maximum of the FITTED map is taken from them
    vmin_out[c] = lo;
    vmax_out[c] = hi;
  }
  double mu = NAN, sd = NAN, M2 = 0.0;
  if (k > 0) {
    mu = s / k;
    M2 = q - s * mu;
    if (M2 < 0.0) M2 = 0.0;
    sd = sqrt(M2 / k);
    if (sd < eps) sd = eps;
  }
  mean[c] = mu;
  stdv[c] = sd;
  const double w = weights ? weights[c] : 1.0;
  const double sh = center ? mu : 0.0, sc = (standardize ? 1.0 / sd : 1.0) * w;
  shift[c] = sh;
  scale[c] = sc;
  m2[c] = M2;
  // max |(x - shift) * scale| of this feature after the transform (1 ulp headroom for the float cast)
  if (k > 0) amax = (float)(fmax(fabs((double)hi - sh), fabs((double)lo - sh)) * fabs(sc) * 1.000001);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  if ((threadIdx.x & 63) == 0 && amax > 0.f && amax < INFINITY) atomicMax(absmax, __float_as_uint(amax));
}

// max over the features of |(x - shift) * scale| from the per-feature extremes of the data and a GIVEN (fitted) shift /
// scale: the transform path of new data (eofx_apply_f32) then needs no extra read of the written matrix.  The map is
// monotone in x, so the extreme values decide; same 1-ulp headroom as colstats_finalize_kernel.
__global__ __launch_bounds__(256) void fitted_absmax_kernel(const int* __restrict__ cnt, const float* __restrict__ vmin,
                                                            const float* __restrict__ vmax, const double* __restrict__ shift,
                                                            const double* __restrict__ scale, int64_t P,
                                                            unsigned* __restrict__ absmax) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float amax = 0.f;
  if (c < P && cnt[c] > 0)
    amax = (float)(fmax(fabs((double)vmax[c] - shift[c]), fabs((double)vmin[c] - shift[c])) * fabs(scale[c]) * 1.000001);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  if ((threadIdx.x & 63) == 0 && amax > 0.f && amax < INFINITY) atomicMax(absmax, __float_as_uint(amax));
}

// Per-workgroup summary of the column statistics (fixed-order tree): number of valid features, min / max of
// their non-NaN counts, and the total variance sum_c scale_c^2 M2_c / (cnt_c - 1) -- so that the common
// NaN-free case needs a few hundred bytes from the device instead of P-sized arrays.
__global__ __launch_bounds__(256) void feature_summary_kernel(const int* __restrict__ cnt,
                                                              const double* __restrict__ m2,
                                                              const double* __restrict__ scale, int64_t P,
                                                              double* __restrict__ tv_part,
                                                              int* __restrict__ ipart) {
  __shared__ double st[256];
  __shared__ int sp[256], smin[256], smax[256];
  const int tid = threadIdx.x;
  double tv = 0.0;
  int pv = 0, cmin = INT32_MAX, cmax = 0;
  const int64_t per = (P + gridDim.x - 1) / gridDim.x;
  const int64_t c0 = (int64_t)blockIdx.x * per, c1 = (c0 + per < P) ? c0 + per : P;
  for (int64_t c = c0 + tid; c < c1; c += 256) {
    const int k = cnt[c];
    if (k > 0) {
      ++pv;
      cmin = k < cmin ? k : cmin;
      cmax = k > cmax ? k : cmax;
      tv += scale[c] * scale[c] * m2[c] / (double)(k - 1);
    }
  }
  st[tid] = tv;
  sp[tid] = pv;
  smin[tid] = cmin;
  smax[tid] = cmax;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      st[tid] += st[tid + s];
      sp[tid] += sp[tid + s];
      smin[tid] = smin[tid + s] < smin[tid] ? smin[tid + s] : smin[tid];
      smax[tid] = smax[tid + s] > smax[tid] ? smax[tid + s] : smax[tid];
    }
    __syncthreads();
  }
  if (tid == 0) {
    tv_part[blockIdx.x] = st[0];
    ipart[3 * blockIdx.x] = sp[0];
    ipart[3 * blockIdx.x + 1] = smin[0];
    ipart[3 * blockIdx.x + 2] = smax[0];
  }
}

// count of non-NaN entries per row restricted to valid feature columns
__global__ __launch_bounds__(256) void rowcount_kernel(const float* __restrict__ X, int64_t n,
                                                        int64_t P, const int* __restrict__ colcnt,
                                                        int* __restrict__ rowcnt) {
  __shared__ int red[256];
  const int64_t r = blockIdx.x;
  int k = 0;
  for (int64_t c = threadIdx.x; c < P; c += 256) {
    const float v = X[r * P + c];
    if (colcnt[c] > 0 && v == v) ++k;
  }
  red[threadIdx.x] = k;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) rowcnt[r] = red[0];
}

// shift / scale (float64, per source column) -> {hi, lo, scale} float triples of the raw view; columns >= p: zeros.
// cnt (may be null): features without a single value (all-NaN grid points kept as zero columns of a masked in-place
// matrix) get the zero triple too -- scale 0 is what the MASK kernels key on.
__global__ __launch_bounds__(256) void aff_pack_kernel(const double* __restrict__ shift, const double* __restrict__ scale,
                                                        int64_t p, int64_t p_pad, float* __restrict__ out,
                                                        const int* __restrict__ cnt = nullptr) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p_pad) return;
  float hi = 0.f, lo = 0.f, sl = 0.f;
  if (c < p && (!cnt || cnt[c] > 0)) {
    aff_split(shift ? shift[c] : 0.0, hi, lo);
    sl = scale ? (float)scale[c] : 1.f;
  }
  out[c] = hi;
  out[p_pad + c] = lo;
  out[2 * p_pad + c] = sl;
}

// grid = (p_pad/64, n_pad/64): 64x64 tile of the compacted matrix, 16 B per lane everywhere:
// each thread owns 4 adjacent compact columns; 16 threads cover a 256 B row segment.
// col_map/row_map: compact index -> source index (null = identity).  shift/scale indexed by
// SOURCE column (null = 0 / 1).  nan_flag set (atomicOr) when a NaN lands on a kept entry.
// VEC: the source can be read with aligned 16 B loads (identity column map, ld % 4 == 0).
template <bool VEC>
__global__ __launch_bounds__(256) void apply_kernel(const float* __restrict__ X, int64_t ldx_src,
                                                     const int64_t* __restrict__ row_map,
                                                     const int64_t* __restrict__ col_map,
                                                     const double* __restrict__ shift,
                                                     const double* __restrict__ scale, int64_t n,
                                                     int64_t p, float* __restrict__ Xc,
                                                     int64_t p_pad, float* __restrict__ Xt,
                                                     int64_t n_pad, int* __restrict__ nan_flag,
                                                     const float* __restrict__ aff, int64_t aff_ld) {
  __shared__ float T[64][65];
  const int tid = threadIdx.x;
  const int tq = tid & 15, tr = tid >> 4;  // column quad 0..15, row 0..15 (+16 per pass)
  float vmax = 0.f;
  const int64_t c0 = (int64_t)blockIdx.x * 64, r0 = (int64_t)blockIdx.y * 64;
  const int64_t cb = c0 + 4 * tq;
  int64_t sc[4];
  float shh[4], shl[4], slf[4];   // shift as a float pair (hi + lo carries the float64 mean), scale in float
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t c = cb + e;
    sc[e] = -1;
    shh[e] = shl[e] = 0.f;
    slf[e] = 1.f;
    if (c < p) {
      sc[e] = col_map ? col_map[c] : c;
      if (aff) {   // the packed float triples of a raw / in-place matrix (aff_pack_kernel): the same map
        shh[e] = aff[sc[e]];
        shl[e] = aff[aff_ld + sc[e]];
        slf[e] = aff[2 * aff_ld + sc[e]];
      } else {
        if (shift) aff_split(shift[sc[e]], shh[e], shl[e]);
        if (scale) slf[e] = (float)scale[sc[e]];
      }
    }
  }
  bool bad = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rr = tr + 16 * q;
    const int64_t r = r0 + rr;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < n) {
      const int64_t sr = row_map ? row_map[r] : r;
      const float* src = X + sr * ldx_src;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (VEC && cb + 3 < p) {
        x = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + cb));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (sc[e] >= 0) x[e] = src[sc[e]];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (sc[e] >= 0) {
          // scale 0 in the packed triples of an in-place matrix = an all-NaN grid point kept as a zero column
          const bool masked_ = aff && slf[e] == 0.f;
          if (x[e] != x[e] && !masked_) bad = true;
          v[e] = masked_ ? 0.f : aff_map(x[e], shh[e], shl[e], slf[e]);
        }
    }
    if (Xc) *reinterpret_cast<f32x4*>(Xc + r * p_pad + cb) = v;   // absent in raw mode (eofx_ctx_set_layout)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      T[rr][4 * tq + e] = v[e];
      vmax = fmaxf(vmax, fabsf(v[e]));
    }
  }
  if (bad && nan_flag) atomicOr(nan_flag, 1);
  (void)vmax;    // the maximum comes from the column statistics (or panel_absmax_kernel): one atomic per
                 // wave here would serialise 10^7 updates of a single word
  __syncthreads();
  // transposed write: thread owns 4 consecutive samples of one feature column per pass
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int cc = tr + 16 * q;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = T[4 * tq + e][cc];
    *reinterpret_cast<f32x4*>(Xt + (c0 + cc) * n_pad + r0 + 4 * tq) = o;
  }
}

// dense download of the resident matrix: dst[n x p]
__global__ __launch_bounds__(256) void mat_download_kernel(const float* __restrict__ Xc,
                                                           int64_t p_pad, int64_t n, int64_t p,
                                                           float* __restrict__ dst) {
  const int64_t total = n * p;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / p, c = i - r * p;
    dst[i] = Xc[r * p_pad + c];
  }
}

// out[n x p] = S[n x k] V[p x k]^T  (k <= 64 per pass; HBM-write bound)
__global__ __launch_bounds__(256) void reconstruct_kernel(const float* __restrict__ S,
                                                          const float* __restrict__ V, int64_t n,
                                                          int64_t p, int k,
                                                          float* __restrict__ out) {
  __shared__ float Ss[64][65];
  __shared__ float Vs[64][65];
  const int tid = threadIdx.x;
  const int tc = tid & 63, tr = tid >> 6;
  const int64_t c0 = (int64_t)blockIdx.x * 64, r0 = (int64_t)blockIdx.y * 64;
  float acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (int k0 = 0; k0 < k; k0 += 64) {
    for (int i = tid; i < 64 * 64; i += 256) {
      const int r = i >> 6, c = i & 63;
      Ss[r][c] = (r0 + r < n && k0 + c < k) ? S[(r0 + r) * k + k0 + c] : 0.f;
      Vs[r][c] = (c0 + r < p && k0 + c < k) ? V[(c0 + r) * k + k0 + c] : 0.f;
    }
    __syncthreads();
    const int kk = (k - k0 < 64) ? k - k0 : 64;
    for (int j = 0; j < kk; ++j) {
      const float v = Vs[tc][j];
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] += Ss[tr + 4 * q][j] * v;
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int64_t r = r0 + tr + 4 * q, c = c0 + tc;
    if (r < n && c < p) out[r * p + c] = acc[q];
  }
}

// sum_i a[i]*b[i] in float64 over float32 inputs, fixed tree: partial per block
__global__ __launch_bounds__(256) void dotprod_part_kernel(const float* __restrict__ a,
                                                           const float* __restrict__ b,
                                                           int64_t count,
                                                           double* __restrict__ part) {
  __shared__ double red[256];
  double s = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
    s += (double)a[i] * (double)b[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// ---------------------------------------------------------------------------------
// Hilbert transform along the sample axis (xeofs/utils/hilbert_transform.py:40-114), per feature
// row of the sample-contiguous layout Xt.  The FFTs themselves are batched hipFFT C2C plans; the
// kernels here build the exponentially padded series, apply the analytic-signal filter and
// extract / re-centre the middle segment.
// ---------------------------------------------------------------------------------
struct cfloat {
  float x, y;
};

__device__ __forceinline__ double block_sum_256(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  const double out = red[0];
  __syncthreads();
  return out;
}

// The padded transform is linear in the series y (the linear fit, the pad amplitudes and the pads are
// all linear functionals of y), so for the middle n samples
//     Im(analytic(y_ext))[n:2n] = T y + amp_pre u1 + amp_pos u2 + c0 u3 + c1 u4
// with T the n x n Toeplitz block of the period-3n Hilbert kernel and u1..u4 fixed vectors (host,
// once per (n, decay)).  T y is a linear convolution with lags |d| < n: one circular convolution of
// POWER-OF-TWO length P >= 2n (rocFFT's single-kernel territory) instead of a length-3n transform.

// one workgroup per feature: zero-padded series -> work row; fit/pad coefficients -> coef[f] (4 floats)
__global__ __launch_bounds__(256) void hilbert_pack_kernel(const float* __restrict__ Xt, int64_t n_pad,
                                                            int64_t n, int64_t f0, int padding,
                                                            float* __restrict__ work, int64_t ldw, int64_t P,
                                                            float* __restrict__ coef) {
  __shared__ double red[256];
  const int64_t f = f0 + blockIdx.x;
  const float* y = Xt + f * n_pad;
  float* out = work + (int64_t)blockIdx.x * ldw;
  double sy = 0.0, sty = 0.0;
  const double tbar = 0.5 * (double)(n - 1);
  for (int64_t i = threadIdx.x; i < P; i += 256) {
    float v = 0.f;
    if (i < n) {
      v = y[i];
      sy += (double)v;
      sty += ((double)i - tbar) * (double)v;
    }
    out[i] = v;
  }
  if (!padding) return;
  sy = block_sum_256(sy, red);
  sty = block_sum_256(sty, red);
  if (threadIdx.x == 0) {
    const double stt = (double)n * ((double)n * (double)n - 1.0) / 12.0;
    const double c1 = (n > 1) ? sty / stt : 0.0;
    const double c0 = sy / (double)n - c1 * tbar;  // fit(t) = c0 + c1 t  (numpy polyfit deg 1)
    float* cf = coef + (int64_t)blockIdx.x * 4;
    cf[0] = (float)((double)y[0] - c0);                                  // amp_pre
    cf[1] = (float)((double)y[n - 1] - (c0 + c1 * (double)(n - 1)));     // amp_pos
    cf[2] = (float)c0;
    cf[3] = (float)c1;
  }
}

// spec[f][k] *= chat[k]   (chat = rfft of the Toeplitz kernel, 1/P of the inverse folded in)
__global__ __launch_bounds__(256) void hilbert_filter_kernel(cfloat* __restrict__ spec,
                                                              const cfloat* __restrict__ chat, int64_t nh,
                                                              int64_t total) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const cfloat c = chat[i % nh];
    const cfloat v = spec[i];
    spec[i] = cfloat{v.x * c.x - v.y * c.y, v.x * c.y + v.y * c.x};
  }
}

// first n samples of the convolution + the four corrections, minus the mean -> Bt row; optionally the
// re-centred input (Re of the analytic signal) -> At row
__global__ __launch_bounds__(256) void hilbert_unpack_kernel(const float* __restrict__ work, int64_t ldw,
                                                              int64_t n, int64_t n_pad, int64_t f0,
                                                              int padding, const float* __restrict__ coef,
                                                              const float* __restrict__ u,
                                                              const float* __restrict__ Xt,
                                                              float* __restrict__ Bt,
                                                              float* __restrict__ At,
                                                              unsigned* __restrict__ bmax,
                                                              unsigned* __restrict__ amax) {
  __shared__ double red[256];
  const float* in = work + (int64_t)blockIdx.x * ldw;
  const int64_t f = f0 + blockIdx.x;
  const float* y = Xt + f * n_pad;
  float a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
  if (padding) {
    const float* cf = coef + (int64_t)blockIdx.x * 4;
    a1 = cf[0]; a2 = cf[1]; a3 = cf[2]; a4 = cf[3];
  }
  double si = 0.0, sr = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    float v = in[i];
    if (padding) v += a1 * u[i] + a2 * u[n + i] + a3 * u[2 * n + i] + a4 * u[3 * n + i];
    si += (double)v;
    if (At) sr += (double)y[i];
  }
  si = block_sum_256(si, red) / (double)n;
  sr = block_sum_256(sr, red) / (double)n;
  float mb = 0.f, ma = 0.f;
  for (int64_t i = threadIdx.x; i < n_pad; i += 256) {
    const bool ok = i < n;
    float v = 0.f;
    if (ok) {
      v = in[i];
      if (padding) v += a1 * u[i] + a2 * u[n + i] + a3 * u[2 * n + i] + a4 * u[3 * n + i];
      v = (float)((double)v - si);
    }
    Bt[f * n_pad + i] = v;
    mb = fmaxf(mb, fabsf(v));
    if (At) {
      const float a = ok ? (float)((double)y[i] - sr) : 0.f;
      At[f * n_pad + i] = a;
      ma = fmaxf(ma, fabsf(a));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mb = fmaxf(mb, __shfl_xor(mb, o));
    ma = fmaxf(ma, __shfl_xor(ma, o));
  }
  if ((threadIdx.x & 63) == 0) {
    if (mb > 0.f) atomicMax(bmax, __float_as_uint(mb));
    if (At && ma > 0.f) atomicMax(amax, __float_as_uint(ma));
  }
}

// dst[c x r] = src[r x c]^T, both dense with the given leading dimensions (multiples of 64)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int64_t ld_src,
                                                         float* __restrict__ dst, int64_t ld_dst) {
  __shared__ float T[64][65];
  const int tid = threadIdx.x;
  const int tq = tid & 15, tr = tid >> 4;
  const int64_t c0 = (int64_t)blockIdx.x * 64, r0 = (int64_t)blockIdx.y * 64;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rr = tr + 16 * q;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (r0 + rr) * ld_src + c0 + 4 * tq);
#pragma unroll
    for (int e = 0; e < 4; ++e) T[rr][4 * tq + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int cc = tr + 16 * q;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = T[4 * tq + e][cc];
    *reinterpret_cast<f32x4*>(dst + (c0 + cc) * ld_dst + r0 + 4 * tq) = o;
  }
}

// complex panel algebra on [Re | Im] panels (Re in columns [0,h), Im in [h, 2h), h = L/2):
//   out.re = P1.re + sgn * P2.im ;  out.im = P1.im - sgn * P2.re
// sgn = +1:  Z^H W = (A^T - i B^T)(Wr + i Wi)   with P1 = A^T [Wr|Wi], P2 = B^T [Wr|Wi]
// sgn = -1:  Z   Y = (A + i B)(Yr + i Yi)       with P1 = A [Yr|Yi],   P2 = B [Yr|Yi]
__global__ __launch_bounds__(256) void cpanel_combine_kernel(const float* __restrict__ P1,
                                                              const float* __restrict__ P2, float sgn,
                                                              int64_t rows, int L,
                                                              float* __restrict__ out) {
  const int h = L / 2;
  const int64_t total = rows * h;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / h;
    const int c = (int)(i - r * h);
    const float p1r = P1[r * L + c], p1i = P1[r * L + h + c];
    const float p2r = P2[r * L + c], p2i = P2[r * L + h + c];
    out[r * L + c] = p1r + sgn * p2i;
    out[r * L + h + c] = p1i - sgn * p2r;
  }
}

// companion panel of a complex pass: out = sgn * [Pi | -Pr]  (so that A^T [Pr|Pi] + B^T out = [Re | Im] of Z^H P for
// sgn = +1, and A [Pr|Pi] + B out = Z P for sgn = -1)
__global__ __launch_bounds__(256) void cpanel_rot_kernel(const float* __restrict__ P, float sgn, int64_t rows, int L,
                                                          float* __restrict__ out) {
  const int h = L / 2;
  const int64_t total = rows * h;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / h;
    const int c = (int)(i - r * h);
    out[r * L + c] = sgn * P[r * L + h + c];
    out[r * L + h + c] = -sgn * P[r * L + c];
  }
}

// (re, im) of column j at the rows amax[j] and amin[j]: out[4 j .. 4 j + 3]  (complex sign rule)
__global__ void cpanel_pick_kernel(const float* __restrict__ P, int L, int k, const int64_t* __restrict__ amax,
                                   const int64_t* __restrict__ amin, float* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  const int h = L / 2;
  out[4 * j + 0] = P[amax[j] * L + j];
  out[4 * j + 1] = P[amax[j] * L + h + j];
  out[4 * j + 2] = P[amin[j] * L + j];
  out[4 * j + 3] = P[amin[j] * L + h + j];
}

// [Re(ko) | Im(ko)] panel -> dense [rows x k] interleaved complex64, optional column signs
__global__ __launch_bounds__(256) void cpanel_export_kernel(const float* __restrict__ P, int64_t rows, int L, int k,
                                                             const double* __restrict__ sign, float* __restrict__ dst) {
  const int h = L / 2;
  const int64_t total = rows * k;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / k;
    const int c = (int)(i - r * k);
    const float sg = sign ? (float)sign[c] : 1.f;
    dst[2 * i] = sg * P[r * L + c];
    dst[2 * i + 1] = sg * P[r * L + h + c];
  }
}

// Replacement columns for a factor whose trailing modes are numerically null (fix_null_columns in eofx_abi.hip): column c with
// colflag[c] != 0 becomes a fixed pseudo-random vector in (-1, 1) on the rows that carry anything in the columns before
// `first` (rows of masked features / padding stay zero), zero elsewhere.
__global__ __launch_bounds__(256) void null_fill_kernel(float* __restrict__ P, int64_t rows, int Lo, int first, int k,
                                                         const int* __restrict__ colflag) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    float* row = P + r * Lo;
    bool live = first == 0;
    for (int c = 0; c < first; ++c) live = live || row[c] != 0.f;
    for (int c = first; c < k; ++c) {
      if (!colflag[c]) continue;
      unsigned h = (unsigned)(r * 0x9E3779B1ull) ^ ((unsigned)c * 0x85EBCA6Bu + 0xC2B2AE35u);
      h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
      row[c] = live ? (float)(h >> 8) * (1.f / 8388608.f) - 1.f : 0.f;
    }
  }
}

// per-column arg max / arg min over rows [0, rows): (value, row) partials, ties -> lowest row
__global__ __launch_bounds__(256) void colargminmax_part_kernel(const float* __restrict__ P, int64_t rows,
                                                                 int L, float* __restrict__ pmx,
                                                                 int64_t* __restrict__ imx,
                                                                 float* __restrict__ pmn,
                                                                 int64_t* __restrict__ imn,
                                                                 const float* __restrict__ rowscale = nullptr) {
  // rowscale (may be null): rows r with rowscale[r] == 0 are skipped -- the masked features of an in-place matrix, whose
  // rows of V are exact zeros and must not take part in the sign rule of the compacted matrix
  __shared__ float smx[4][64], smn[4][64];
  __shared__ int64_t sax[4][64], san[4][64];
  const int tid = threadIdx.x;
  const int c = blockIdx.y * 64 + (tid & 63);
  const int rl = tid >> 6;
  float mx = -INFINITY, mn = INFINITY;
  int64_t ax = -1, an = -1;
  if (c < L) {
    // four rows in flight per thread (independent loads; the compares run in increasing row order: ties keep the lowest row)
    const int64_t step = (int64_t)gridDim.x * 4;
    int64_t r = (int64_t)blockIdx.x * 4 + rl;
    for (; r + 3 * step < rows; r += 4 * step) {
      float v[4];
      bool live[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        live[u] = !(rowscale && rowscale[r + u * step] == 0.f);
        v[u] = P[(r + u * step) * L + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!live[u]) continue;
        if (v[u] > mx) { mx = v[u]; ax = r + u * step; }
        if (v[u] < mn) { mn = v[u]; an = r + u * step; }
      }
    }
    for (; r < rows; r += step) {
      if (rowscale && rowscale[r] == 0.f) continue;
      const float v = P[r * L + c];
      if (v > mx) { mx = v; ax = r; }
      if (v < mn) { mn = v; an = r; }
    }
  }
  smx[rl][tid & 63] = mx; sax[rl][tid & 63] = ax;
  smn[rl][tid & 63] = mn; san[rl][tid & 63] = an;
  __syncthreads();
  if (rl == 0 && c < L) {
    for (int q = 1; q < 4; ++q) {
      const float a = smx[q][tid], b = smn[q][tid];
      const int64_t ia = sax[q][tid], ib = san[q][tid];
      if (a > mx || (a == mx && ia >= 0 && (ax < 0 || ia < ax))) { mx = a; ax = ia; }
      if (b < mn || (b == mn && ib >= 0 && (an < 0 || ib < an))) { mn = b; an = ib; }
    }
    const int64_t o = (int64_t)blockIdx.x * L + c;
    pmx[o] = mx; imx[o] = ax; pmn[o] = mn; imn[o] = an;
  }
}
__global__ void colargminmax_final_kernel(const float* __restrict__ pmx, const int64_t* __restrict__ imx,
                                          const float* __restrict__ pmn, const int64_t* __restrict__ imn,
                                          int nparts, int L, int64_t* __restrict__ amax,
                                          int64_t* __restrict__ amin) {
  // block = 64 columns x 16 part lanes; the tie rule names the row explicitly, so any reduction order gives the same answer
  __shared__ float sa[16][64], sb[16][64];
  __shared__ int64_t sia[16][64], sib[16][64];
  const int cl = threadIdx.x & 63, ql = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a = -INFINITY, b = INFINITY;
  int64_t ia = -1, ib = -1;
  auto take = [&](float va, int64_t ja, float vb, int64_t jb) {
    if (ja >= 0 && (ia < 0 || va > a || (va == a && ja < ia))) { a = va; ia = ja; }
    if (jb >= 0 && (ib < 0 || vb < b || (vb == b && jb < ib))) { b = vb; ib = jb; }
  };
  if (c < L)
    for (int q = ql; q < nparts; q += 16)
      take(pmx[(int64_t)q * L + c], imx[(int64_t)q * L + c], pmn[(int64_t)q * L + c], imn[(int64_t)q * L + c]);
  sa[ql][cl] = a; sia[ql][cl] = ia; sb[ql][cl] = b; sib[ql][cl] = ib;
  __syncthreads();
  if (ql == 0 && c < L) {
    for (int q = 1; q < 16; ++q) take(sa[q][cl], sia[q][cl], sb[q][cl], sib[q][cl]);
    amax[c] = ia;
    amin[c] = ib;
  }
}

// max over the rows of |P[r, c] + i P[r, c + L/2]| for the L/2 complex columns of a [Re | Im] panel -> out[L/2]
// (float bits through atomicMax on the unsigned view: non-negative values, order-independent; out zeroed first)
__global__ __launch_bounds__(256) void cpanel_colabsmax_kernel(const float* __restrict__ P, int64_t rows, int L,
                                                               unsigned* __restrict__ out) {
  const int h = L / 2;
  const int c = threadIdx.x % h;
  const int rstep = 256 / h;
  float m = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * rstep + threadIdx.x / h; r < rows; r += (int64_t)gridDim.x * rstep) {
    const float a = P[r * L + c], b = P[r * L + h + c];
    m = fmaxf(m, sqrtf(a * a + b * b));
  }
  if (threadIdx.x < rstep * h && m > 0.f) atomicMax(out + c, __float_as_uint(m));
}

// ---------------------------------------------------------------------------------
// Varimax / Promax rotation of loadings (xeofs/linalg/_numpy/_rotation.py:6-187), modes <= 64.
// The loadings panel X (rows = features, L = 64 columns) stays resident; one iteration of the
// reference loop  basis = X R;  T = basis * (|basis|^2 - alpha W);  G = X^T T  is ONE pass over X:
// per row  b = x R (float64),  t = f(b),  G += left^T t  with everything but the final m x m SVD on
// the device.  mode 0 (varimax step):  left = x, t_j = b_j (b_j^2 - aw_j)
//              mode 1 (promax fit)  :  left = b, t_j = (b_j / mx_j) |b_j / mx_j|^(power-1)
// Complex loadings (modes 2 / 3 = complex versions of 0 / 1; "This implementation also works for complex numbers",
// _rotation.py:16,105): the panel holds [Re (32 columns) | Im (32 columns)], R arrives as the real 64 x 64 embedding
// [[Rr, Ri], [-Ri, Rr]] so that b = x R is the same row product, |b_j|^2 pairs column j with column j + 32, and the
// 64 x 64 real result [Xr | Xi]^T [Tr | Ti] carries the four blocks of X^H T = (Xr^T Tr + Xi^T Ti) + i (Xr^T Ti - Xi^T Tr).
// grid = (nblocks); each workgroup owns a strided set of 32-row tiles; partials reduced in fixed order.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rot_step_kernel(const float* __restrict__ X, int64_t rows, int L,
                                                       const double* __restrict__ R,
                                                       const double* __restrict__ aux, int mode,
                                                       double power, double* __restrict__ Gpart) {
  // Both matrix products of a 32-row tile run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64; operand / result lane
  // maps as in gram_mfma_kernel): b = x R as 2 x 4 tiles over 16 k-steps (wave w: row tile w & 1, column tiles
  // 2 (w >> 1) + {0, 1}), then G += left^T t as 4 x 4 tiles over 8 k-steps (wave w: tile row w, accumulators kept
  // across the whole launch).  Only the elementwise transform in between is vector-ALU work.  LDS rows are padded by
  // two doubles (float: four) so that the four k-rows a wave reads at once fall into different banks.
  __shared__ float Xs[32][68];
  __shared__ double Rs[64][66];
  __shared__ double Ts[32][66];
  __shared__ double Ls[32][66];   // b = x R (the left factor in modes 1 / 3, the partner parts in modes 2 / 3)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  for (int i = tid; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    Rs[r][c] = (r < L && c < L) ? R[(int64_t)r * L + c] : 0.0;
  }
  const int brow = tid >> 3, bc0 = (tid & 7) * 8;   // elementwise step: row brow, columns bc0..bc0+7
  f64x4 acc[4];
#pragma unroll
  for (int y = 0; y < 4; ++y) acc[y] = f64x4{0.0, 0.0, 0.0, 0.0};
  double auxr[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) auxr[e] = (bc0 + e < L) ? aux[bc0 + e] : ((mode & 1) ? 1.0 : 0.0);
  const int rt = wave & 1, ct0 = 2 * (wave >> 1);
  __syncthreads();
  for (int64_t r0 = (int64_t)blockIdx.x * 32; r0 < rows; r0 += (int64_t)gridDim.x * 32) {
    for (int i = tid; i < 32 * 16; i += 256) {
      const int rr = i >> 4, c4 = (i & 15) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r0 + rr < rows && c4 < L) v = *reinterpret_cast<const f32x4*>(X + (r0 + rr) * L + c4);
      *reinterpret_cast<f32x4*>(&Xs[rr][c4]) = v;
    }
    __syncthreads();
    // b = x R
    f64x4 bt[2] = {f64x4{0.0, 0.0, 0.0, 0.0}, f64x4{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const double a = (double)Xs[16 * rt + li][4 * s + lk];
#pragma unroll
      for (int c = 0; c < 2; ++c)
        bt[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Rs[4 * s + lk][16 * (ct0 + c) + li], bt[c], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ls[16 * rt + lk + 4 * r][16 * (ct0 + c) + li] = bt[c][r];   // D[lk + 4 r][li]
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double b = Ls[brow][bc0 + e];
      double t;
      if (mode >= 2) {       // complex: column j pairs with column j + 32
        const int cr = (bc0 + e) & 31;
        const double br_ = Ls[brow][cr], bi_ = Ls[brow][cr + 32];
        const double a2 = br_ * br_ + bi_ * bi_;
        if (mode == 2) {
          t = b * (a2 - auxr[e]);
        } else {
          const double za = sqrt(a2) / auxr[e];
          t = (power == 1.0 || !(za > 0.0)) ? b / auxr[e] : (b / auxr[e]) * pow(za, power - 1.0);
        }
      } else if (mode == 0) {
        t = b * (b * b - auxr[e]);
      } else {
        const double z = b / auxr[e];
        t = (power == 1.0) ? z : z * pow(fabs(z), power - 1.0);
      }
      Ts[brow][bc0 + e] = t;
    }
    __syncthreads();
    // G += left^T t  (left = x in modes 0 / 2, b in modes 1 / 3): k runs over the 32 rows of the tile
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const double a = (mode & 1) ? Ls[4 * s + lk][16 * wave + li] : (double)Xs[4 * s + lk][16 * wave + li];
#pragma unroll
      for (int y = 0; y < 4; ++y)
        acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Ts[4 * s + lk][16 * y + li], acc[y], 0, 0, 0);
    }
    __syncthreads();
  }
  double* G = Gpart + (int64_t)blockIdx.x * L * L;
#pragma unroll
  for (int y = 0; y < 4; ++y)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = 16 * wave + lk + 4 * r, gj = 16 * y + li;      // D[lk + 4 r][li] of tile (wave, y)
      if (gi < L && gj < L) G[(int64_t)gi * L + gj] = acc[y][r];
    }
}

// The same step for 65 .. 256 modes (panels 128 / 256 wide; complex: 64 / 128 columns per half).  An LW x LW float64
// accumulator does not fit one workgroup's registers, so G is cut into column blocks of 64: workgroup (x, cb) walks a
// strided set of 32-row tiles like rot_step_kernel and owns G[:, 64 cb .. 64 cb + 63].  It needs t only for its own
// columns, hence b = x R only there (plus the block holding the partner parts in the complex Varimax mode; the whole
// width when the left factor is b, modes 1 / 3 -- the Promax regression, one step per rotation): the float64 matrix
// work of an iteration stays 2 x rows x LW^2 multiply-adds over the launch.  R (up to 512 KB) is read from L2 in the
// operand layout of v_mfma_f64_16x16x4_f64 (a k-row of 16 columns = one 128-byte line); 8 waves: b tile (w >> 2,
// w & 3) of the block, G tile rows LW/128 per wave x 4 column tiles.
template <int LW>
__global__ __launch_bounds__(512) void rot_step_wide_kernel(const float* __restrict__ X, int64_t rows,
                                                            const double* __restrict__ R,
                                                            const double* __restrict__ aux, int mode, double power,
                                                            double* __restrict__ Gpart) {
  static_assert(LW == 128 || LW == 256, "128 / 256-wide panels");
  constexpr int TR = 32, NCB = LW / 64, XP = LW + 4, DP = LW + 2, HALF = LW / 2, TRW = LW / 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char rot_smem[];
  float (*Xs)[XP] = reinterpret_cast<float (*)[XP]>(rot_smem);
  double (*Ls)[DP] = reinterpret_cast<double (*)[DP]>(rot_smem + sizeof(float) * TR * XP);
  double (*Ts)[66] = reinterpret_cast<double (*)[66]>(rot_smem + sizeof(float) * TR * XP + sizeof(double) * TR * DP);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int cb = blockIdx.y;             // this workgroup's 64 columns of t and of G
  const int pb = cb ^ (NCB / 2);         // complex modes: the block with the partner parts (column j pairs with j +- LW/2)
  const int brow = tid >> 4, bq = 4 * (tid & 15);   // elementwise step: row brow, columns bq .. bq + 3 of the block
  double auxr[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) auxr[e] = aux[64 * cb + bq + e];
  const int rt = wave >> 2, ct = wave & 3;
  f64x4 acc[TRW][4];
#pragma unroll
  for (int q = 0; q < TRW; ++q)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[q][y] = f64x4{0.0, 0.0, 0.0, 0.0};
  for (int64_t r0 = (int64_t)blockIdx.x * TR; r0 < rows; r0 += (int64_t)gridDim.x * TR) {
    for (int i = tid; i < TR * (LW / 4); i += 512) {
      const int rr = i / (LW / 4), c4 = (i % (LW / 4)) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r0 + rr < rows) v = *reinterpret_cast<const f32x4*>(X + (r0 + rr) * LW + c4);
      *reinterpret_cast<f32x4*>(&Xs[rr][c4]) = v;
    }
    __syncthreads();
    // b = x R on the column blocks this workgroup needs
    for (int blk = 0; blk < NCB; ++blk) {
      if (!((mode & 1) || blk == cb || (mode == 2 && blk == pb))) continue;
      const double* Rc = R + 64 * blk + 16 * ct + li;
      f64x4 bt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
      for (int s = 0; s < LW / 4; ++s)
        bt = __builtin_amdgcn_mfma_f64_16x16x4f64((double)Xs[16 * rt + li][4 * s + lk], Rc[(4 * s + lk) * LW], bt, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) Ls[16 * rt + lk + 4 * r][64 * blk + 16 * ct + li] = bt[r];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = 64 * cb + bq + e;
      const double b = Ls[brow][col];
      double t;
      if (mode >= 2) {
        const int cr = col & (HALF - 1);
        const double br_ = Ls[brow][cr], bi_ = Ls[brow][cr + HALF];
        const double a2 = br_ * br_ + bi_ * bi_;
        if (mode == 2) {
          t = b * (a2 - auxr[e]);
        } else {
          const double za = sqrt(a2) / auxr[e];
          t = (power == 1.0 || !(za > 0.0)) ? b / auxr[e] : (b / auxr[e]) * pow(za, power - 1.0);
        }
      } else if (mode == 0) {
        t = b * (b * b - auxr[e]);
      } else {
        const double z = b / auxr[e];
        t = (power == 1.0) ? z : z * pow(fabs(z), power - 1.0);
      }
      Ts[brow][bq + e] = t;
    }
    __syncthreads();
    // G[:, block] += left^T t
#pragma unroll
    for (int s = 0; s < TR / 4; ++s) {
#pragma unroll
      for (int q = 0; q < TRW; ++q) {
        const int lc_ = 16 * (TRW * wave + q) + li;
        const double a = (mode & 1) ? Ls[4 * s + lk][lc_] : (double)Xs[4 * s + lk][lc_];
#pragma unroll
        for (int y = 0; y < 4; ++y)
          acc[q][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Ts[4 * s + lk][16 * y + li], acc[q][y], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  double* G = Gpart + (int64_t)blockIdx.x * LW * LW;
#pragma unroll
  for (int q = 0; q < TRW; ++q)
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        G[(int64_t)(16 * (TRW * wave + q) + lk + 4 * r) * LW + 64 * cb + 16 * y + li] = acc[q][y][r];
}

// Kaiser normalisation: out[r,:] = P[r,:] / (||P[r,:]|| + eps)   (one wave per row quad)
__global__ __launch_bounds__(256) void row_normalize_kernel(const float* __restrict__ P, int64_t rows, int L,
                                                            double eps, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  double s = 0.0;
  for (int c = lane; c < L; c += 64) {
    const double v = (double)P[r * L + c];
    s += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const double inv = 1.0 / (sqrt(s) + eps);
  for (int c = lane; c < L; c += 64) out[r * L + c] = (float)((double)P[r * L + c] * inv);
}

// Euclidean norm of every row of a row-major [rows x L] array with leading dimension ld (one wave per
// row, float64 accumulation).  Used for the per-feature standard deviations of the correlation patterns.
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ P, int64_t rows, int64_t L,
                                                      int64_t ld, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  double s = 0.0;
  for (int64_t c = lane; c < L; c += 64) {
    const double v = (double)P[r * ld + c];
    s += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[r] = sqrt(s);
}

// the same over the RAW field of an in-place matrix: rows of aff_map(raw) (see atb_f16_kernel<NB, true>)
__global__ __launch_bounds__(256) void rownorm_aff_kernel(const float* __restrict__ P, int64_t rows, int64_t L, int64_t ld,
                                                          const float* __restrict__ aff, int64_t aff_ld,
                                                          double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  double s = 0.0;
  for (int64_t c = lane; c < L; c += 64) {
    const float sl = aff[2 * aff_ld + c];      // 0: an all-NaN grid point kept as a zero column
    const double v = sl == 0.f ? 0.0 : (double)aff_map(P[r * ld + c], aff[c], aff[aff_ld + c], sl);
    s += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[r] = sqrt(s);
}

}  // namespace eofx
