// The Hilbert stage of HilbertEOF / HilbertMCA (xeofs/utils/hilbert_transform.py:40-114) as ONE kernel per field.
//
// For the middle n samples of the exponentially padded series the transform is  Im = T y + (four rank-one corrections)
// (see eofx_kernels.hpp, "Hilbert transform along the sample axis"), T the n x n Toeplitz block of the Hilbert kernel:
// a circular convolution of power-of-two length P >= 2n with a REAL, ODD kernel c.  So
//   * two features share one complex transform, z = y_a + i y_b: c * z = (c * y_a) + i (c * y_b);
//   * the kernel spectrum is purely imaginary, FFT(c)[k] = i h[k]: the filter is one real table;
//   * a decimation-in-frequency forward transform leaves the spectrum digit-reversed, the mirrored
//     decimation-in-time inverse takes it from there: no reordering pass, the table is stored in that order.
// One workgroup of P/16 threads owns a pair of features for the whole trip: the series come from HBM straight into
// the registers of the first butterfly (thread t holds samples t + c P/16; the zero padding is never materialised),
// the transform lives in LDS (P complex values, skewed against bank conflicts), and the last inverse butterfly
// hands thread t the samples it started with -- corrections, centring and the store happen in registers.  HBM
// traffic is the algorithmic minimum (read y once, write Im once); the next pair's loads are in flight during the
// LDS stages.  Radix 16 butterflies (4 x 4 in registers), one optional leading radix 2 / 4 / 8 stage, twiddle factors
// per thread computed once in float64 and kept in registers (a thread meets the same butterfly for every pair).
//
// MODE 1 (series of 8193 .. 16384 samples, circular length 2^15): one feature per workgroup through the half-length
// complex transform of its even / odd samples, z[j] = y[2j] + i y[2j+1].  The real-input split, the filter and the merge
// back collapse into one step on the pair (Z[k], Z[M-k]):  W[k] = i hm[k] Z[k] + hp2[k] (w^k D - conj(w^k) S),
// D = Z[k] - conj Z[M-k], S = Z[k] + conj Z[M-k], hm = h[k] - h[M-k], hp2 = (h[k] + h[M-k]) / 2, w = exp(-2 pi i / 2M).
// In the digit-reversed layout the 16 frequencies of a middle-stage row have their partners in ONE other row, in
// reverse order, so the step costs one extra trip through LDS in the middle stage.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef HFFT_WAVES
#define HFFT_WAVES 4
#endif
namespace hfft {

typedef float cf __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// Complex products and quarter turns as packed-f32 instructions with explicit operand selection: a product is two
// instructions (the compiler's rendering of the same arithmetic is four or five: it forms both sign variants and
// merges halves with moves), x +- i y folds the turn into the add.
__device__ __forceinline__ cf cmul(cf a, cf b) {    // a b
  cf r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ cf cmulc(cf a, cf b) {   // a conj(b)
  cf r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ cf cmul_k(cf a, cf b) {    // a b, b uniform (scalar registers)
  cf r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "s"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(r) : "v"(a), "s"(b));
  return r;
}
__device__ __forceinline__ cf cmulc_k(cf a, cf b) {   // a conj(b), b uniform
  cf r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "s"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(r) : "v"(a), "s"(b));
  return r;
}
// DIR = +1: forward (factors exp(-i ...)), DIR = -1: inverse (their conjugates)
template <int DIR> __device__ __forceinline__ cf twm(cf a, cf w) { return DIR > 0 ? cmul(a, w) : cmulc(a, w); }
template <int DIR> __device__ __forceinline__ cf rot90(cf a) { return DIR > 0 ? cf{a.y, -a.x} : cf{-a.y, a.x}; }  // * (-+ i)
// a + rot90<DIR>(t): forward (a.x + t.y, a.y - t.x), inverse (a.x - t.y, a.y + t.x)
template <int DIR> __device__ __forceinline__ cf add_rot(cf a, cf t) {
  cf r;
  if constexpr (DIR > 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(t));
  else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(t));
  return r;
}

// a * w16^K, w16 = exp(-+ 2 pi i / 16)
template <int DIR, int K> __device__ __forceinline__ cf mulw16(cf a) {
  constexpr int k = K & 15;
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
  if constexpr (k == 0) return a;
  else if constexpr (k == 4) return rot90<DIR>(a);
  else if constexpr (k == 8) return cf{-a.x, -a.y};
  else if constexpr (k == 12) return rot90<-DIR>(a);
  else {
    constexpr float c = (k == 1 || k == 15) ? C1 : (k == 3 || k == 13) ? S1 : (k == 5 || k == 11) ? -S1 : (k == 7 || k == 9) ? -C1
                        : (k == 2 || k == 14) ? H : -H;                                                   // cos(pi k / 8)
    constexpr float sn = (k == 1 || k == 7) ? S1 : (k == 3 || k == 5) ? C1 : (k == 9 || k == 15) ? -S1 : (k == 11 || k == 13) ? -C1
                         : (k == 2 || k == 6) ? H : -H;                                                   // sin(pi k / 8)
    return DIR > 0 ? cmul_k(a, cf{c, -sn}) : cmulc_k(a, cf{c, -sn});
  }
}

template <int DIR> __device__ __forceinline__ void dft4(cf& a0, cf& a1, cf& a2, cf& a3) {
  const cf s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
  a0 = s02 + s13; a1 = add_rot<DIR>(d02, d13); a2 = s02 - s13; a3 = add_rot<-DIR>(d02, d13);
}
template <int DIR> __device__ __forceinline__ void dft4_lo(cf& a0, cf& a1, cf& a2, cf& a3) {   // a2 = a3 = 0 on entry
  const cf x0 = a0, x1 = a1;
  a0 = x0 + x1; a1 = add_rot<DIR>(x0, x1); a2 = x0 - x1; a3 = add_rot<-DIR>(x0, x1);
}

// 16-point transform in registers, natural order in; result q sits in v[slot16(q)].  HALF: v[8..15] are zero on entry.
__host__ __device__ constexpr int slot16(int q) { return 4 * (q & 3) + (q >> 2); }
template <int DIR, bool HALF> __device__ __forceinline__ void dft16(cf* v) {
#pragma unroll
  for (int j1 = 0; j1 < 4; ++j1) {
    if constexpr (HALF) dft4_lo<DIR>(v[j1], v[j1 + 4], v[j1 + 8], v[j1 + 12]);
    else dft4<DIR>(v[j1], v[j1 + 4], v[j1 + 8], v[j1 + 12]);
  }
  v[5] = mulw16<DIR, 1>(v[5]);   v[9] = mulw16<DIR, 2>(v[9]);    v[13] = mulw16<DIR, 3>(v[13]);
  v[6] = mulw16<DIR, 2>(v[6]);   v[10] = mulw16<DIR, 4>(v[10]);  v[14] = mulw16<DIR, 6>(v[14]);
  v[7] = mulw16<DIR, 3>(v[7]);   v[11] = mulw16<DIR, 6>(v[11]);  v[15] = mulw16<DIR, 9>(v[15]);
#pragma unroll
  for (int q2 = 0; q2 < 4; ++q2) dft4<DIR>(v[4 * q2], v[4 * q2 + 1], v[4 * q2 + 2], v[4 * q2 + 3]);
}
// 8-point transform on a0..a7 (natural order in); result q sits in slot8(q) = 2 (q & 3) + (q >> 2)
__host__ __device__ constexpr int slot8(int q) { return 2 * (q & 3) + (q >> 2); }
template <int DIR, bool HALF> __device__ __forceinline__ void dft8(cf& a0, cf& a1, cf& a2, cf& a3, cf& a4, cf& a5,
                                                                   cf& a6, cf& a7) {
  if constexpr (HALF) { dft4_lo<DIR>(a0, a2, a4, a6); dft4_lo<DIR>(a1, a3, a5, a7); }
  else { dft4<DIR>(a0, a2, a4, a6); dft4<DIR>(a1, a3, a5, a7); }
  a3 = mulw16<DIR, 2>(a3); a5 = mulw16<DIR, 4>(a5); a7 = mulw16<DIR, 6>(a7);
  cf t;
  t = a0; a0 = t + a1; a1 = t - a1;
  t = a2; a2 = t + a3; a3 = t - a3;
  t = a4; a4 = t + a5; a5 = t - a5;
  t = a6; a6 = t + a7; a7 = t - a7;
}

// LDS position of element idx: 16 B of padding per 16 elements (the stride-1 stage reads 128 B per lane) and
// 128 B more per 256 elements (the stride-16 stage)
__host__ __device__ constexpr int phys(int idx) { return idx + 2 * (idx >> 4) + 16 * (idx >> 8); }

__device__ __forceinline__ cf unit(double num, double den) {   // exp(-2 pi i num / den)
  double s, c;
  sincospi(-2.0 * num / den, &s, &c);
  return cf{(float)c, (float)s};
}

// stored powers w^1 w^2 w^4 w^8 of a butterfly's base factor (each rounded from float64); w^3 = w^1 w^2 and
// w^12 = w^4 w^8 are formed per stage, and w^q = w^(4a) w^b -- at most three float32 roundings away from exact
struct tw4 {
  cf w1, w2, w4, w8;
};
struct tw6 {
  cf w1, w2, w3, w4, w8, w12;
};
__device__ __forceinline__ tw4 make_tw4(int lo, int span) {
  return tw4{unit(lo, span), unit(2.0 * lo, span), unit(4.0 * lo, span), unit(8.0 * lo, span)};
}
__device__ __forceinline__ tw6 expand(const tw4& s) { return tw6{s.w1, s.w2, cmul(s.w1, s.w2), s.w4, s.w8, cmul(s.w4, s.w8)}; }
template <int DIR, int Q> __device__ __forceinline__ cf apply_tw(cf a, const tw6& t) {
  constexpr int hi = Q >> 2, lo = Q & 3;
  if constexpr (lo == 1) a = twm<DIR>(a, t.w1);
  if constexpr (lo == 2) a = twm<DIR>(a, t.w2);
  if constexpr (lo == 3) a = twm<DIR>(a, t.w3);
  if constexpr (hi == 1) a = twm<DIR>(a, t.w4);
  if constexpr (hi == 2) a = twm<DIR>(a, t.w8);
  if constexpr (hi == 3) a = twm<DIR>(a, t.w12);
  return a;
}

// one radix-16 stage through LDS: elements base + j S.  Forward: transform, then the stage's factors; inverse: mirrored.
// (S is 16 or 256 and base has no bits at S .. 16 S, so phys(base + j S) = phys(base) + phys(j S): `at` = data + phys(base))
template <int DIR, int S> __device__ __forceinline__ void stage16(cf* at, const tw4& ts) {
  static_assert(S == 16 || S == 256, "stride");
  const tw6 t = expand(ts);
  cf v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = at[phys(j * S)];
  if constexpr (DIR > 0) {
    dft16<1, false>(v);
#define HFFT_ST(Q) at[phys(Q * S)] = apply_tw<1, Q>(v[slot16(Q)], t);
    HFFT_ST(0) HFFT_ST(1) HFFT_ST(2) HFFT_ST(3) HFFT_ST(4) HFFT_ST(5) HFFT_ST(6) HFFT_ST(7)
    HFFT_ST(8) HFFT_ST(9) HFFT_ST(10) HFFT_ST(11) HFFT_ST(12) HFFT_ST(13) HFFT_ST(14) HFFT_ST(15)
#undef HFFT_ST
  } else {
#define HFFT_LD(Q) v[Q] = apply_tw<-1, Q>(v[Q], t);
    HFFT_LD(1) HFFT_LD(2) HFFT_LD(3) HFFT_LD(4) HFFT_LD(5) HFFT_LD(6) HFFT_LD(7)
    HFFT_LD(8) HFFT_LD(9) HFFT_LD(10) HFFT_LD(11) HFFT_LD(12) HFFT_LD(13) HFFT_LD(14) HFFT_LD(15)
#undef HFFT_LD
    dft16<-1, false>(v);
#pragma unroll
    for (int j = 0; j < 16; ++j) at[phys(j * S)] = v[slot16(j)];
  }
}

// The register-resident outer stage on e[c] <-> element t + c NT (NT = P/16): the leading radix-RL stage, or the first
// radix-16 stage when there is none.  Forward: e[8..15] are zero on entry.  w holds the powers of exp(-2 pi i t / P).
template <int RL, int DIR> __device__ __forceinline__ void outer_stage(cf* e, const tw4& ws) {
  const tw6 w = expand(ws);
  if constexpr (RL == 1) {
    if constexpr (DIR > 0) {
      dft16<1, true>(e);
      cf o[16];
#define HFFT_O(Q) o[Q] = apply_tw<1, Q>(e[slot16(Q)], w);
      HFFT_O(0) HFFT_O(1) HFFT_O(2) HFFT_O(3) HFFT_O(4) HFFT_O(5) HFFT_O(6) HFFT_O(7)
      HFFT_O(8) HFFT_O(9) HFFT_O(10) HFFT_O(11) HFFT_O(12) HFFT_O(13) HFFT_O(14) HFFT_O(15)
#undef HFFT_O
#pragma unroll
      for (int q = 0; q < 16; ++q) e[q] = o[q];
    } else {
#define HFFT_I(Q) e[Q] = apply_tw<-1, Q>(e[Q], w);
      HFFT_I(1) HFFT_I(2) HFFT_I(3) HFFT_I(4) HFFT_I(5) HFFT_I(6) HFFT_I(7)
      HFFT_I(8) HFFT_I(9) HFFT_I(10) HFFT_I(11) HFFT_I(12) HFFT_I(13) HFFT_I(14) HFFT_I(15)
#undef HFFT_I
      dft16<-1, false>(e);
      cf o[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) o[j] = e[slot16(j)];
#pragma unroll
      for (int j = 0; j < 16; ++j) e[j] = o[j];
    }
  } else if constexpr (RL == 2) {
    // butterflies m = 0..7 on (e[m], e[m + 8]); factor of output 1: w^1 w16^m
#define HFFT_B2(M)                                                        \
    if constexpr (DIR > 0) {                                              \
      e[M + 8] = mulw16<1, M>(twm<1>(e[M], w.w1));                        \
    } else {                                                              \
      const cf y1 = mulw16<-1, M>(twm<-1>(e[M + 8], w.w1));               \
      const cf y0 = e[M];                                                 \
      e[M] = y0 + y1; e[M + 8] = y0 - y1;                                 \
    }
    HFFT_B2(0) HFFT_B2(1) HFFT_B2(2) HFFT_B2(3) HFFT_B2(4) HFFT_B2(5) HFFT_B2(6) HFFT_B2(7)
#undef HFFT_B2
  } else if constexpr (RL == 4) {
    // butterflies m = 0..3 on e[m + 4 j]; factor of output q: w^q w16^(m q)
#define HFFT_B4(M)                                                                         \
    if constexpr (DIR > 0) {                                                               \
      dft4_lo<1>(e[M], e[M + 4], e[M + 8], e[M + 12]);                                     \
      e[M + 4] = mulw16<1, M>(twm<1>(e[M + 4], w.w1));                                     \
      e[M + 8] = mulw16<1, 2 * M>(twm<1>(e[M + 8], w.w2));                                 \
      e[M + 12] = mulw16<1, 3 * M>(twm<1>(e[M + 12], w.w3));                               \
    } else {                                                                               \
      e[M + 4] = mulw16<-1, M>(twm<-1>(e[M + 4], w.w1));                                   \
      e[M + 8] = mulw16<-1, 2 * M>(twm<-1>(e[M + 8], w.w2));                               \
      e[M + 12] = mulw16<-1, 3 * M>(twm<-1>(e[M + 12], w.w3));                             \
      dft4<-1>(e[M], e[M + 4], e[M + 8], e[M + 12]);                                       \
    }
    HFFT_B4(0) HFFT_B4(1) HFFT_B4(2) HFFT_B4(3)
#undef HFFT_B4
  } else {
    // RL == 8: butterflies m = 0, 1 on e[m + 2 j]; factor of output q: w^q w16^(m q)
    const cf w5 = cmul(w.w4, w.w1), w6 = cmul(w.w4, w.w2), w7 = cmul(w.w4, w.w3);
#define HFFT_B8(M)                                                                                        \
    { if constexpr (DIR > 0) {                                                                              \
      dft8<1, true>(e[M], e[M + 2], e[M + 4], e[M + 6], e[M + 8], e[M + 10], e[M + 12], e[M + 14]);       \
      cf o[8];                                                                                            \
      o[0] = e[M + 2 * slot8(0)];                                                                         \
      o[1] = mulw16<1, M>(twm<1>(e[M + 2 * slot8(1)], w.w1));                                             \
      o[2] = mulw16<1, 2 * M>(twm<1>(e[M + 2 * slot8(2)], w.w2));                                         \
      o[3] = mulw16<1, 3 * M>(twm<1>(e[M + 2 * slot8(3)], w.w3));                                         \
      o[4] = mulw16<1, 4 * M>(twm<1>(e[M + 2 * slot8(4)], w.w4));                                         \
      o[5] = mulw16<1, 5 * M>(twm<1>(e[M + 2 * slot8(5)], w5));                                           \
      o[6] = mulw16<1, 6 * M>(twm<1>(e[M + 2 * slot8(6)], w6));                                           \
      o[7] = mulw16<1, 7 * M>(twm<1>(e[M + 2 * slot8(7)], w7));                                           \
      _Pragma("unroll") for (int q = 0; q < 8; ++q) e[M + 2 * q] = o[q];                                  \
    } else {                                                                                              \
      e[M + 2] = mulw16<-1, M>(twm<-1>(e[M + 2], w.w1));                                                  \
      e[M + 4] = mulw16<-1, 2 * M>(twm<-1>(e[M + 4], w.w2));                                              \
      e[M + 6] = mulw16<-1, 3 * M>(twm<-1>(e[M + 6], w.w3));                                              \
      e[M + 8] = mulw16<-1, 4 * M>(twm<-1>(e[M + 8], w.w4));                                              \
      e[M + 10] = mulw16<-1, 5 * M>(twm<-1>(e[M + 10], w5));                                              \
      e[M + 12] = mulw16<-1, 6 * M>(twm<-1>(e[M + 12], w6));                                              \
      e[M + 14] = mulw16<-1, 7 * M>(twm<-1>(e[M + 14], w7));                                              \
      dft8<-1, false>(e[M], e[M + 2], e[M + 4], e[M + 6], e[M + 8], e[M + 10], e[M + 12], e[M + 14]);     \
      cf o[8];                                                                                            \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) o[j] = e[M + 2 * slot8(j)];                           \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) e[M + 2 * j] = o[j];                                  \
    } }
    HFFT_B8(0) HFFT_B8(1)
#undef HFFT_B8
  }
}

// Frequency held by LDS position pos after the forward stages (host side: order of the filter table)
inline int64_t position_frequency(int L, int64_t pos) {
  const int rl_bits = L & 3;
  int64_t k = 0, weight = 1, span = (int64_t)1 << L;
  if (rl_bits) {
    span >>= rl_bits;
    k += (pos / span) * weight;
    pos %= span;
    weight <<= rl_bits;
  }
  while (span > 1) {
    span >>= 4;
    k += (pos / span) * weight;
    pos %= span;
    weight <<= 4;
  }
  return k;
}

// L = log2 P in [10, 14].  One workgroup of NT = P/16 threads owns a pair of features (persistent over pairs).
//   Xt [p][n_pad] sample-contiguous input; Bt same shape: Im of the analytic signal minus its mean over the samples;
//   the kernel writes samples [0, P/2) of a row, so it needs n_pad <= P/2 (true for n_pad = round_up(n, 512), P >= 1024);
//   At (optional): the input minus its mean; hperm [P]: filter table in LDS-position order (1/P folded in);
//   u [n][4]: the four correction vectors, interleaved per sample, followed by their four means over the samples
//   (padding only; without padding the table loads return zeros); bmax / amax: running absmax of the outputs (float bits).
template <int L> struct plan {
  static_assert(L >= 10 && L <= 14, "circular length 2^10 .. 2^14");
  static constexpr int P = 1 << L, NT = P / 16, RL = 1 << (L & 3), N16 = L / 4;
  static constexpr int WG = NT;                        // threads per workgroup
  static constexpr int NW = NT / 64;                   // waves per workgroup
  static constexpr int S0 = P / RL;                    // span entering the radix-16 stages
  static constexpr int FIRST16 = (RL == 1) ? 1 : 0;    // radix-16 stages through LDS: FIRST16 .. N16 - 2, then the middle
  static constexpr int NIN = N16 - 1 - FIRST16;        // twiddled LDS stages: 1 or 2
  static constexpr int SA = (S0 >> 4) >> (4 * FIRST16), SB = SA >> 4;   // their strides
  static constexpr size_t lds = (size_t)phys(P) * 8 + 1536;
};

// Several sums at once over a full wave: every exchange step also halves the number of values a lane carries, so
// four sums cost 7 exchanges instead of 24.  reduce4: lanes of group g = lane >> 4 end with the total of value g
// (a, b, c, d); reduce2: lanes 0-31 the total of a, lanes 32-63 the total of b.
__device__ __forceinline__ double tail16(double k) {
  k += __shfl_xor(k, 8); k += __shfl_xor(k, 4); k += __shfl_xor(k, 2); k += __shfl_xor(k, 1);
  return k;
}
__device__ __forceinline__ double reduce4(double a, double b, double c, double d, int lane) {
  const bool h32 = lane & 32, h16 = lane & 16;
  double k0 = h32 ? c : a, k1 = h32 ? d : b;
  k0 += __shfl_xor(h32 ? a : c, 32);
  k1 += __shfl_xor(h32 ? b : d, 32);
  double k = h16 ? k1 : k0;
  k += __shfl_xor(h16 ? k0 : k1, 16);
  return tail16(k);
}
__device__ __forceinline__ double reduce2(double a, double b, int lane) {
  const bool h32 = lane & 32;
  double k = h32 ? b : a;
  k += __shfl_xor(h32 ? a : b, 32);
  k += __shfl_xor(k, 16);
  return tail16(k);
}
__device__ __forceinline__ double lane_value(double v, int src) {   // v of lane src (compile-time), uniform
  union { double d; int i[2]; } x;
  x.d = v;
  x.i[0] = __builtin_amdgcn_readlane(x.i[0], src);
  x.i[1] = __builtin_amdgcn_readlane(x.i[1], src);
  return x.d;
}

__device__ __forceinline__ double wave_total(double v) {   // every lane ends with the sum over the wave
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// keeps the optimiser from carrying values derived from x across iterations of the pair loop (it would hoist eight
// 64-bit addresses per array, the float64 sample indices, negated copies of every factor, ... and spill them)
template <typename T> __device__ __forceinline__ void fresh(T& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void fresh(cf& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void fresh(tw4& w) { fresh(w.w1); fresh(w.w2); fresh(w.w4); fresh(w.w8); }
// Rows are addressed through buffer resources: a lane beyond the row's byte count reads 0 / stores nothing, so the
// ragged ends (n is not a multiple of the group size, the odd last feature, idle groups) need no branches, and a
// lane's address is one 32-bit offset shared by every row.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x2v __attribute__((__vector_size__(2 * sizeof(unsigned int))));
__device__ __forceinline__ rsrc_t row_rsrc(const float* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float ld_nt(rsrc_t r, unsigned off, unsigned soff = 0) {   // streaming (nt) load
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 2));
}
__device__ __forceinline__ float ld(rsrc_t r, unsigned off, unsigned soff = 0) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0));
}
__device__ __forceinline__ void st_nt(float v, rsrc_t r, unsigned off) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 2);
}
__device__ __forceinline__ unsigned absbits(float x) { return __builtin_bit_cast(unsigned, x) & 0x7fffffffu; }

// Middle-stage row t holds the frequencies klow(t) + q 2^(L-4), q = 0..15 (digit reversal of the stages above it)
template <int L> __device__ __forceinline__ int klow_of_row(int t) {
  constexpr int r = L & 3, ND = L / 4 - 1;
  int k = 0, w = 0, shift = 4 * ND;
  if constexpr (r > 0) { k = (t >> shift) & ((1 << r) - 1); w = r; }
#pragma unroll
  for (int d = 0; d < ND; ++d) { shift -= 4; k |= ((t >> shift) & 15) << w; w += 4; }
  return k;
}
template <int L> __device__ __forceinline__ int row_of_klow(int k) {
  constexpr int r = L & 3, ND = L / 4 - 1;
  int t = 0, w = 0, shift = 4 * ND;
  if constexpr (r > 0) { t = (k & ((1 << r) - 1)) << shift; w = r; }
#pragma unroll
  for (int d = 0; d < ND; ++d) { shift -= 4; t |= ((k >> w) & 15) << shift; w += 4; }
  return t;
}
// a * exp(-2 pi i Q / 32)
template <int Q> __device__ __forceinline__ cf mulw32(cf a) {
  constexpr float C[16] = {1.f, 0.98078528040323044f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                           0.55557023301960222f, 0.38268343236508977f, 0.19509032201612827f, 0.f, -0.19509032201612827f,
                           -0.38268343236508977f, -0.55557023301960222f, -0.70710678118654752f, -0.83146961230254524f,
                           -0.92387953251128674f, -0.98078528040323044f};
  constexpr float S[16] = {0.f, 0.19509032201612827f, 0.38268343236508977f, 0.55557023301960222f, 0.70710678118654752f,
                           0.83146961230254524f, 0.92387953251128674f, 0.98078528040323044f, 1.f, 0.98078528040323044f,
                           0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960222f,
                           0.38268343236508977f, 0.19509032201612827f};
  if constexpr (Q == 0) return a;
  else if constexpr (Q == 8) return rot90<1>(a);
  else return cmul_k(a, cf{C[Q], -S[Q]});
}

template <int L, int MODE>
__global__ __launch_bounds__(plan<L>::WG, HFFT_WAVES) void hilbert_fft_kernel(const float* __restrict__ Xt, int64_t n_pad,
                                                                              int n, int64_t p, int padding,
                                                                              const float* __restrict__ hperm,
                                                                              const float* __restrict__ u,
                                                                              float* __restrict__ Bt, float* __restrict__ At,
                                                                              unsigned* __restrict__ bmax,
                                                                              unsigned* __restrict__ amax,
                                                                              const float* __restrict__ aff = nullptr,
                                                                              int64_t aff_ld = 0,
                                                                              const float* __restrict__ oscale = nullptr,
                                                                              double* __restrict__ sqpart = nullptr) {
  using PL = plan<L>;
  constexpr int P = PL::P, NT = PL::NT, RL = PL::RL, NW = PL::NW, NIN = PL::NIN, SA = PL::SA, SB = PL::SB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t0 = tid;
  cf* const data = reinterpret_cast<cf*>(smem);
  double* const red1 = reinterpret_cast<double*>(smem + (size_t)phys(P) * 8);   // [16][4]  (NW > 1 only)
  double* const red2 = red1 + 64;                                                    // [16][2]
  float* const coef = reinterpret_cast<float*>(red2 + 32);                           // [2][6]: amp_pre amp_pos c0 c1 mean_y -
  float* const edge = coef + 12;                                                     // y_a[0] y_a[n-1] y_b[0] y_b[n-1]

  // per-thread factors, once
  tw4 wout = make_tw4(t0, P);
  tw4 wa = NIN >= 1 ? make_tw4(t0 & (SA - 1), 16 * SA) : tw4{};
  tw4 wb = NIN >= 2 ? make_tw4(t0 & (SB - 1), 16 * SB) : tw4{};
  // MODE 0: thread t holds samples t + c NT of two features; MODE 1: samples 2 j, 2 j + 1 (j = t + c NT) of one feature
  const int jl = MODE ? (n - 1) >> 1 : n - 1;
  const int cl = jl / NT, tl = jl % NT;             // owner of the last sample
  const double tbar = 0.5 * (double)(n - 1);
  const int64_t npairs = MODE ? p : (p + 1) / 2;   // work items: features (MODE 1) or pairs of features
  cf wmid = cf{1.f, 0.f};
  int trow = 0;                                     // MODE 1: the row holding the partner frequencies M - k
  if constexpr (MODE == 1) {
    const int kl = klow_of_row<L>(t0);
    wmid = unit(kl, 2.0 * P);
    trow = row_of_klow<L>((NT - kl) & (NT - 1));
  }
  const bool sums = padding || At;
  const unsigned nb = (unsigned)n * 4u, npb = (unsigned)n_pad * 4u;

  unsigned run_mx = 0u, run_my = 0u;   // running absmax of this thread's outputs (one atomic per wave at the end)
  // sqpart (MODE 0): the sum of squares of the outputs, one float64 partial per wave in a fixed order (total variance of
  // the imaginary part -- eofx_hilbert_sumsq_f64); with Bt == nullptr the rows are not written at all (zero-sized descriptors)
  double run_sq = 0.0;
  float ya[8], yb[8];
  auto load_pair = [&](int64_t pr, unsigned tb4) {
    if constexpr (MODE == 1) {   // (the padded tail of a row is zero: whole 8-byte pairs up to n_pad)
      const rsrc_t ra = row_rsrc(Xt + pr * n_pad, pr < npairs ? npb : 0u);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const cf v = __builtin_bit_cast(cf, __builtin_amdgcn_raw_buffer_load_b64(ra, 2u * (tb4 + (unsigned)(c * NT * 4)), 0, 2));
        ya[c] = v.x; yb[c] = v.y;
      }
    } else {
      const int64_t fa = 2 * pr, fb = fa + 1;
      const rsrc_t ra = row_rsrc(Xt + fa * n_pad, pr < npairs ? nb : 0u);
      const rsrc_t rb = row_rsrc(Xt + fb * n_pad, fb < p ? nb : 0u);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const unsigned o = tb4 + (unsigned)(c * NT * 4);
        ya[c] = ld_nt(ra, o);
        yb[c] = ld_nt(rb, o);
      }
    }
  };
#pragma unroll
  for (int c = 0; c < 8; ++c) ya[c] = yb[c] = 0.f;
  const int64_t stride = gridDim.x;
  int64_t pair = blockIdx.x;
  load_pair(pair, (unsigned)t0 * 4u);
  for (; pair < npairs; pair += stride) {
    int t = t0;
    fresh(t);
    fresh(wout);
    if constexpr (NIN >= 1) fresh(wa);
    if constexpr (NIN >= 2) fresh(wb);
    const unsigned tb4 = (unsigned)t * 4u;
    cf* const at_o = data + phys(t);                                                       // element t + c NT at at_o[phys(c NT)]
    cf* const at_a = data + phys((t / SA) * 16 * SA + (t & (SA - 1)));
    cf* const at_b = data + phys((t / SB) * 16 * SB + (t & (SB - 1)));
    cf* const row = data + phys(16 * t);
    const int64_t fa = MODE ? pair : 2 * pair, fb = fa + 1;
    const bool hb = MODE ? false : fb < p;
    if constexpr (MODE == 0) {
      if (aff) {   // rows of the RAW field (eofx_hilbert_f32, from_rawT): the Scaler map ((x - hi) - lo) * scale of apply_kernel
        const float ha = aff[fa], la = aff[aff_ld + fa], sa_ = aff[2 * aff_ld + fa];
        const float hb_ = hb ? aff[fb] : 0.f, lb = hb ? aff[aff_ld + fb] : 0.f, sb_ = hb ? aff[2 * aff_ld + fb] : 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const bool in = tb4 + (unsigned)(c * NT * 4) < nb;      // (samples beyond n were read as zeros and stay zeros)
          ya[c] = (in && sa_ != 0.f) ? ((ya[c] - ha) - la) * sa_ : 0.f;      // (scale 0: an all-NaN grid point kept as a zero column)
          yb[c] = (in && sb_ != 0.f) ? ((yb[c] - hb_) - lb) * sb_ : 0.f;
        }
      }
    }
    // ---- linear fit and pad amplitudes (float64 sums over the group)
    double sa = 0.0, ta = 0.0, sb = 0.0, tb = 0.0;
    if (sums) {
      double s1a = 0.0, s1b = 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) {   // sample index t + c NT; zero samples beyond n add nothing
        sa += (double)ya[c]; s1a += (double)c * (double)ya[c];
        sb += (double)yb[c]; s1b += (double)c * (double)yb[c];
      }
      if constexpr (MODE == 1) {   // samples 2 j and 2 j + 1
        const double tc = 2.0 * (double)t - tbar;
        ta = tc * sa + 2.0 * (double)NT * s1a;
        tb = (tc + 1.0) * sb + 2.0 * (double)NT * s1b;
      } else {
        const double tc = (double)t - tbar;
        ta = tc * sa + (double)NT * s1a;
        tb = tc * sb + (double)NT * s1b;
      }
      if constexpr (NW > 1) {
        const double k = reduce4(sa, ta, sb, tb, lane);
        if ((lane & 15) == 0) red1[wave * 4 + (lane >> 4)] = k;
      } else {
        sa = wave_total(sa); ta = wave_total(ta); sb = wave_total(sb); tb = wave_total(tb);
      }
      if (t == 0) { edge[0] = ya[0]; edge[2] = yb[0]; }
      if (t == tl) {
        float la = 0.f, lb = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) if (c == cl) { la = ya[c]; lb = yb[c]; }
        if constexpr (MODE == 1) la = ((n - 1) & 1) ? lb : la;   // sample n - 1 is the odd or the even one of its pair
        edge[1] = la; edge[3] = lb;
      }
    }
    // ---- outer forward stage in registers -> LDS
    {
      cf e[16];
#pragma unroll
      for (int c = 0; c < 8; ++c) { e[c] = cf{ya[c], yb[c]}; e[c + 8] = cf{0.f, 0.f}; }
      outer_stage<RL, 1>(e, wout);
#pragma unroll
      for (int c = 0; c < 16; ++c) at_o[phys(c * NT)] = e[c];
    }
    __syncthreads();
    if (sums && (NW == 1 || wave == 0)) {
      int ft = t;              // feature of the pair this lane finishes (0 / 1), if any
      bool fin = t < 2;
      if constexpr (NW > 1) {   // lane l: value l >> 4 of wave l & 15
        double k = ((lane & 15) < NW) ? red1[(lane & 15) * 4 + (lane >> 4)] : 0.0;
        k = tail16(k);
        const double o = __shfl_xor(k, 16);
        sa = sb = k; ta = tb = o;      // lanes 0 / 32 hold (sum y, sum t y) of features a / b
        ft = lane >> 5;
        fin = (lane & 31) == 0;
        if constexpr (MODE == 1) {     // one series: its even and odd halves add up
          sa += __shfl_xor(k, 32); ta += __shfl_xor(o, 32);
          fin = lane == 0;
        }
      } else if constexpr (MODE == 1) {
        sa += sb; ta += tb;
        fin = t == 0;
      }
      if (fin) {
        const double sy = ft ? sb : sa, sty = ft ? tb : ta;
        const double stt = (double)n * ((double)n * (double)n - 1.0) / 12.0;
        const double c1 = (n > 1) ? sty / stt : 0.0;
        const double c0 = sy / (double)n - c1 * tbar;       // fit(t) = c0 + c1 t  (numpy polyfit deg 1)
        float* cfo = coef + ft * 6;
        cfo[0] = (float)((double)edge[2 * ft] - c0);                                   // amp_pre
        cfo[1] = (float)((double)edge[2 * ft + 1] - (c0 + c1 * (double)(n - 1)));      // amp_pos
        cfo[2] = (float)c0;
        cfo[3] = (float)c1;
        cfo[4] = (float)(sy / (double)n);
      }
    }
    // ---- forward stages through LDS
    if constexpr (NIN >= 1) { stage16<1, SA>(at_a, wa); __syncthreads(); }
    if constexpr (NIN >= 2) { stage16<1, SB>(at_b, wb); __syncthreads(); }
    // ---- middle: last forward stage, the filter, first inverse stage
    {
      cf v[16], w[16];
#pragma unroll
      for (int j = 0; j < 16; j += 2) {
        const f4 q = *reinterpret_cast<const f4*>(row + j);
        v[j] = cf{q.x, q.y}; v[j + 1] = cf{q.z, q.w};
      }
      dft16<1, false>(v);
      if constexpr (MODE == 1) {
        // the spectrum of this row, in frequency order, for the partner row; then the partner's, reversed
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
          const cf a = v[slot16(j)], b = v[slot16(j + 1)];
          *reinterpret_cast<f4*>(row + j) = f4{a.x, a.y, b.x, b.y};
        }
        __syncthreads();
        int tr = trow;
        fresh(tr);
        const cf* const rowp = data + phys(16 * tr);
        const bool self0 = t == 0;      // row 0 pairs k = q M/16 with (16 - q) M/16 inside itself
        cf wm = wmid;
        fresh(wm);
#define HFFT_MB(Q)                                                                                        \
        {                                                                                                 \
          const cf zq = rowp[self0 ? ((16 - Q) & 15) : (15 - Q)];                                         \
          const cf a = v[slot16(Q)], bc = cf{zq.x, -zq.y};                                                \
          const cf wq = mulw32<Q>(wm);                                                                    \
          const cf r = cmul(a - bc, wq) - cmulc(a + bc, wq);                                              \
          const float hm = hperm[16 * t + Q], hp2 = hperm[P + 16 * t + Q];                                \
          w[Q] = cf{hp2 * r.x - hm * a.y, hp2 * r.y + hm * a.x};                                          \
        }
        HFFT_MB(0) HFFT_MB(1) HFFT_MB(2) HFFT_MB(3) HFFT_MB(4) HFFT_MB(5) HFFT_MB(6) HFFT_MB(7)
        HFFT_MB(8) HFFT_MB(9) HFFT_MB(10) HFFT_MB(11) HFFT_MB(12) HFFT_MB(13) HFFT_MB(14) HFFT_MB(15)
#undef HFFT_MB
        __syncthreads();                // every row has been read before any is overwritten
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f4 h = *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(hperm) + (tb4 * 16u + 16u * g));
          const cf z0 = v[slot16(4 * g)], z1 = v[slot16(4 * g + 1)], z2 = v[slot16(4 * g + 2)], z3 = v[slot16(4 * g + 3)];
          w[4 * g] = cf{-h.x * z0.y, h.x * z0.x};      // * i h
          w[4 * g + 1] = cf{-h.y * z1.y, h.y * z1.x};
          w[4 * g + 2] = cf{-h.z * z2.y, h.z * z2.x};
          w[4 * g + 3] = cf{-h.w * z3.y, h.w * z3.x};
        }
      }
      dft16<-1, false>(w);
#pragma unroll
      for (int j = 0; j < 16; j += 2) {
        const cf a = w[slot16(j)], b = w[slot16(j + 1)];
        *reinterpret_cast<f4*>(row + j) = f4{a.x, a.y, b.x, b.y};
      }
    }
    __syncthreads();
    // the next pair's samples travel during the inverse stages (their registers are free until the next outer stage)
    load_pair(pair + stride, tb4);
    // ---- inverse stages through LDS
    if constexpr (NIN >= 2) { stage16<-1, SB>(at_b, wb); __syncthreads(); }
    if constexpr (NIN >= 1) { stage16<-1, SA>(at_a, wa); __syncthreads(); }
    // ---- outer inverse stage LDS -> registers; corrections, centring, store
    {
      cf e[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) e[c] = at_o[phys(c * NT)];
      outer_stage<RL, -1>(e, wout);
      float a1a = 0.f, a2a = 0.f, a3a = 0.f, a4a = 0.f, a1b = 0.f, a2b = 0.f, a3b = 0.f, a4b = 0.f;
      if (padding) {
        a1a = coef[0]; a2a = coef[1]; a3a = coef[2]; a4a = coef[3];
        a1b = coef[MODE ? 0 : 6]; a2b = coef[MODE ? 1 : 7]; a3b = coef[MODE ? 2 : 8]; a4b = coef[MODE ? 3 : 9];
      }
      // The mean over the samples of (convolution + corrections) is the mean of the convolution plus the coefficients
      // times the means of the correction vectors (four per-setup constants stored behind the table): the reduction needs
      // no table access, and the corrections are applied in the store loop, where registers are free to keep several
      // 16-byte table loads in flight (one load at a time is an exposed L2 round trip per sample group).
      float va[8], vb[8];
      double ua = 0.0, ub = 0.0;
      if constexpr (MODE == 0) {
        // (round 5: the kernel is bound by instruction issue, DESIGN.md section 6 -- the pair (feature a, feature b) of a sample
        // is one packed float2 from here on: a float32 partial sum of the thread's 8 samples per feature, then float64 across threads)
        cf acc2 = cf{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const unsigned o = tb4 + (unsigned)(c * NT * 4);
          e[c] = (o < nb) ? e[c] : cf{0.f, 0.f};
          acc2 += e[c];
        }
        ua = (double)acc2.x;
        ub = (double)acc2.y;
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const unsigned o = tb4 + (unsigned)(c * NT * 4);
          float xa = e[c].x, xb = e[c].y;    // samples 2 j, 2 j + 1: byte offsets 2 o, 2 o + 4
          xa = (2u * o < nb) ? xa : 0.f;
          xb = (2u * o + 4u < nb) ? xb : 0.f;
          ua += (double)xa; ub += (double)xb;
          va[c] = xa; vb[c] = xb;
        }
      }
      if constexpr (NW > 1) {
        const double k = reduce2(ua, ub, lane);
        if ((lane & 31) == 0) red2[wave * 2 + (lane >> 5)] = k;
        __syncthreads();
        double m = ((lane & 15) < NW) ? red2[(lane & 15) * 2 + ((lane >> 4) & 1)] : 0.0;   // value (l >> 4) & 1 of wave l & 15
        m = tail16(m);
        ua = lane_value(m, 0);
        ub = lane_value(m, 16);
      } else {
        ua = wave_total(ua); ub = wave_total(ub);
      }
      if constexpr (MODE == 1) ua = ub = ua + ub;      // one series: one mean
      float m0 = (float)(ua / (double)n), m1 = (float)(ub / (double)n);
      if (padding) {
        const float* ubar = u + 4 * (size_t)n;         // means of the four correction vectors
        const float b1 = ubar[0], b2 = ubar[1], b3 = ubar[2], b4 = ubar[3];
        m0 += a1a * b1 + a2a * b2 + a3a * b3 + a4a * b4;
        m1 += a1b * b1 + a2b * b2 + a3b * b3 + a4b * b4;
      }
      const rsrc_t ru = row_rsrc(u, padding ? 4u * nb : 0u);   // [n][4]: 16 bytes per sample
      const rsrc_t wa_ = row_rsrc(Bt + fa * n_pad, Bt ? npb : 0u);
      const rsrc_t wb_ = row_rsrc(Bt + fb * n_pad, (Bt && hb) ? npb : 0u);
      cf sq2 = cf{0.f, 0.f};
      // oscale (masked in-place input): a feature whose Scaler scale is 0 is an all-NaN grid point kept as a zero column -- its
      // output row is written as exact zeros (sharing a complex transform with a live feature leaves rounding noise in it)
      const bool za = oscale && oscale[fa] == 0.f, zb = oscale && hb && oscale[fb] == 0.f;
      constexpr int UB = 2;                                    // sample groups per batch of table loads
#pragma unroll
      for (int c0 = 0; c0 < 8; c0 += UB) {
        f4 ue[UB], uo[UB];
#pragma unroll
        for (int q = 0; q < UB; ++q) {
          const unsigned o = tb4 + (unsigned)((c0 + q) * NT * 4);
          if constexpr (MODE == 1) {                           // vectors of samples 2 j and 2 j + 1
            ue[q] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ru, 8u * o, 0, 0));
            uo[q] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ru, 8u * o + 16u, 0, 0));
          } else {
            ue[q] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ru, 4u * o, 0, 0));
          }
        }
#pragma unroll
        for (int q = 0; q < UB; ++q) {
          const int c = c0 + q;
          const unsigned o = tb4 + (unsigned)(c * NT * 4);
          if constexpr (MODE == 1) {
            float xa = va[c] + (a1a * ue[q].x + a2a * ue[q].y + a3a * ue[q].z + a4a * ue[q].w) - m0;
            float xb = vb[c] + (a1a * uo[q].x + a2a * uo[q].y + a3a * uo[q].z + a4a * uo[q].w) - m0;
            xa = (2u * o < nb) ? xa : 0.f;
            xb = (2u * o + 4u < nb) ? xb : 0.f;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, cf{xa, xb}), wa_, 2u * o, 0, 2);
            run_mx = max(run_mx, max(absbits(xa), absbits(xb)));
          } else {
            // both features of the pair at once: x = (e - mean) + a1 u1 + a2 u2 + a3 u3 + a4 u4 (packed float2 arithmetic)
            cf x = e[c] - cf{m0, m1};
            x = __builtin_elementwise_fma(cf{a1a, a1b}, cf{ue[q].x, ue[q].x}, x);
            x = __builtin_elementwise_fma(cf{a2a, a2b}, cf{ue[q].y, ue[q].y}, x);
            x = __builtin_elementwise_fma(cf{a3a, a3b}, cf{ue[q].z, ue[q].z}, x);
            x = __builtin_elementwise_fma(cf{a4a, a4b}, cf{ue[q].w, ue[q].w}, x);
            const float xa = (o < nb && !za) ? x.x : 0.f;
            const float xb = (o < nb && !zb) ? x.y : 0.f;
            st_nt(xa, wa_, o);
            st_nt(xb, wb_, o);
            if (sqpart) sq2 = __builtin_elementwise_fma(cf{xa, xb}, cf{xa, xb}, sq2);
            run_mx = max(run_mx, max(absbits(xa), absbits(xb)));   // (idle rows carry zeros)
          }
        }
      }
      run_sq += (double)sq2.x + (double)sq2.y;
      if (At) {   // the re-centred input (only asked for when the field was not centred before): second read of y
        const float ma = coef[4], mb = coef[10];
        const rsrc_t ra = row_rsrc(Xt + fa * n_pad, nb);
        const rsrc_t rb = row_rsrc(Xt + fb * n_pad, hb ? nb : 0u);
        const rsrc_t qa = row_rsrc(At + fa * n_pad, npb);
        const rsrc_t qb = row_rsrc(At + fb * n_pad, hb ? npb : 0u);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const unsigned o = tb4 + (unsigned)(c * NT * 4);
          if constexpr (MODE == 1) {
            const cf y2 = __builtin_bit_cast(cf, __builtin_amdgcn_raw_buffer_load_b64(row_rsrc(Xt + fa * n_pad, npb), 2u * o, 0, 0));
            const float xa = (2u * o < nb) ? y2.x - ma : 0.f;
            const float xb = (2u * o + 4u < nb) ? y2.y - ma : 0.f;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, cf{xa, xb}), qa, 2u * o, 0, 2);
            run_my = max(run_my, max(absbits(xa), absbits(xb)));
          } else {
            const float xa = (o < nb) ? ld(ra, o) - ma : 0.f;
            const float xb = (hb && o < nb) ? ld(rb, o) - mb : 0.f;
            st_nt(xa, qa, o);
            st_nt(xb, qb, o);
            run_my = max(run_my, max(absbits(xa), absbits(xb)));
          }
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { run_mx = max(run_mx, (unsigned)__shfl_xor((int)run_mx, o)); run_my = max(run_my, (unsigned)__shfl_xor((int)run_my, o)); }
  if (lane == 0) {   // bit patterns of non-negative floats order like the floats
    if (run_mx) atomicMax(bmax, run_mx);
    if (At && run_my) atomicMax(amax, run_my);
  }
  if (sqpart) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) run_sq += __shfl_xor(run_sq, o);
    if (lane == 0) sqpart[(size_t)blockIdx.x * NW + wave] = run_sq;
  }
}

}  // namespace hfft
