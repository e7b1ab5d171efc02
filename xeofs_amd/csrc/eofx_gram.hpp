// eofx_gram.hpp -- the MFMA-bound products of the engine: sample-space Gram matrices G = X' X'^T (n x n, the contraction
// runs over the ~10^5..10^6 features) on the fp16 matrix cores of gfx950, in the same scaled split-fp16 arithmetic as the
// streaming passes (x = hi + lo, products hh + hl + lh, float32 accumulation: ~2^-22 per product).
//
// Who needs them (reference lines): the total squared covariance of the cross models, ||X^T Y||_F^2 = <X X^T, Y Y^T>
// (xeofs/cross/cpcca.py:197,991-1000), the power iterations of the cross-covariance rSVD carried out in sample space
// (eofx_abi.hip, crosscov driver) and the PCA pre-reduction of the cross models (xeofs/preprocessing/pca.py:94-120).
//
// Two steps, because a GEMM re-reads every operand element ~20 times and the map + split costs ~4 VALU per element:
//   planes_split_kernel   one pass over the field (raw field through the Scaler map, or a written layout): every 32
//                         features of a row become ONE 128-byte line {hi[32], lo[32]} of fp16 -- the same bytes as the
//                         float32 field, so a stage of the GEMM is one full cache line per row and plane pair.
//   gram_nt_kernel        C = A B^T on 256 x 256 tiles, 8 waves (2 x 4, 128 x 64 each), both operands K-contiguous:
//                         LDS-DMA (global_load_lds_dwordx4, no VGPRs, no VALU) of 8 rows x 128 B per instruction into a
//                         lane-linear image whose 16-byte chunks are XOR-swizzled on the SOURCE side, conflict-free
//                         ds_read_b128 fragments for v_mfma_f32_32x32x16_f16, two 64 KiB stages, one barrier per stage
//                         (= 48 MFMAs per wave: the three products triple the matrix work per staged byte), fragment
//                         reads a quarter-stage ahead of their MFMAs with hand-counted waits.
//                         Work items = (tile, K split) dealt so that each XCD owns a compact patch of the tile triangle
//                         and walks the splits in order: a row block's lines are fetched once per XCD and shared in L2.
//   gram_finish_kernel    fixed-order sum of the split partials, mirrored into the full symmetric matrix.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "eofx_kernels.hpp"

namespace eofx {

constexpr int GR_BM = 256;                 // tile rows = tile columns
constexpr int GR_BK = 32;                  // features per stage: one 128-byte line {hi[32], lo[32]} per row
constexpr int GR_OP_BYTES = GR_BM * 128;   // one operand tile of one stage: 32 KiB
constexpr int GR_STAGE_BYTES = 2 * GR_OP_BYTES;

// One pass over the field: (x - mean) * scale of 8 consecutive features of one row -> 8 hi + 8 lo halves at their places
// in the row's 128-byte line.  src: the raw field (aff != nullptr: {shift hi, shift lo, scale}[aff_ld], scale 0 marks
// masked / padding features; the map is the streaming kernels' aff_fma) or a written layout (aff == nullptr).  Rows >= rows
// and features >= cols become zeros.  a_scale: exact power of two.
// planes: [rows_pad][kpad / 32][2][32] fp16, rows_pad % 256 == 0, kpad % 32 == 0.
// grid = (ceil(kpad / 2048), ceil(rows_pad / rows_per_wg)): a thread keeps its 8 features (their map in registers) for all
// rows of the workgroup; a wave reads 2 KiB contiguous per row and writes full 64-byte halves of the lines.
__global__ __launch_bounds__(256) void planes_split_kernel(const float* __restrict__ src, int64_t ld, int64_t rows,
                                                            int64_t cols, const float* __restrict__ aff, int64_t aff_ld,
                                                            float a_scale, _Float16* __restrict__ planes,
                                                            int64_t rows_pad, int64_t kpad, int rows_per_wg) {
  const int64_t k = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (k >= kpad) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_wg;
  const int64_t r1 = r0 + rows_per_wg < rows_pad ? r0 + rows_per_wg : rows_pad;
  float sh[8], sc[8], nls[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool in = k + e < cols;
    sh[e] = (in && aff) ? aff[k + e] : 0.f;
    sc[e] = in ? (aff ? aff[2 * aff_ld + k + e] * a_scale : a_scale) : 0.f;
    nls[e] = (in && aff) ? -(aff[aff_ld + k + e] * sc[e]) : 0.f;
  }
  const bool vec = k + 8 <= cols && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
  float m1 = -1.f;
  asm volatile("" : "+v"(m1));
  char* out = reinterpret_cast<char*>(planes) + r0 * (4 * kpad) + (k >> 5) * 128 + (k & 31) * 2;
  for (int64_t r = r0; r < r1; ++r, out += 4 * kpad) {
    float v[8];
    if (r < rows) {
      const float* p = src + r * ld + k;
      if (vec) {
        const f32x4 x0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        const f32x4 x1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = x0[e];
          v[4 + e] = x1[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = k + e < cols ? p[e] : 0.f;
      }
      // scale 0 = masked grid point (possibly NaN in the field) or padding: a clean zero, as the MASK streaming kernels give
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = sc[e] != 0.f ? aff_fma(v[e], sh[e], sc[e], nls[e]) : 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    u32x4 hi, lo;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const fp16x2_t a = cvt_pk_rn(v[2 * h], v[2 * h + 1]);
      const fp16x2_t b = cvt_pk_rn(__builtin_fmaf((float)a[0], m1, v[2 * h]), __builtin_fmaf((float)a[1], m1, v[2 * h + 1]));
      hi[h] = __builtin_bit_cast(unsigned, a);
      lo[h] = __builtin_bit_cast(unsigned, b);
    }
    *reinterpret_cast<u32x4*>(out) = hi;
    *reinterpret_cast<u32x4*>(out + 64) = lo;
  }
}

// Planes of the TRANSPOSED matrix: row f of the planes = feature (column) f of src, the K axis = the rows of src (samples).
// For the wide products X'^T Z (hundreds of columns: the PCA pre-reduction's p x 1500 panel) through gram_nt_kernel, whose
// operands are both K-contiguous.  src [rows x cols] (ld) through the map as above (aff may be null); planes
// [cols_pad][kpad / 32][2][32], cols_pad % 64 == 0, kpad = round_up(rows, 64).  grid = (cols_pad / 64, kpad / 64): a workgroup
// reads a 64 x 64 tile with 256-byte row segments, turns it in LDS and writes 32-byte pieces of the features' lines.
__global__ __launch_bounds__(256) void planes_split_t_kernel(const float* __restrict__ src, int64_t ld, int64_t rows,
                                                              int64_t cols, const float* __restrict__ aff, int64_t aff_ld,
                                                              float a_scale, _Float16* __restrict__ planes, int64_t kpad) {
  __shared__ float tile[64][65];      // [sample][feature]
  const int64_t f0 = (int64_t)blockIdx.x * 64, s0 = (int64_t)blockIdx.y * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t f = f0 + 4 * tx;
  float sh[4], sc[4], nls[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool in = f + e < cols;
    sh[e] = (in && aff) ? aff[f + e] : 0.f;
    sc[e] = in ? (aff ? aff[2 * aff_ld + f + e] * a_scale : a_scale) : 0.f;
    nls[e] = (in && aff) ? -(aff[aff_ld + f + e] * sc[e]) : 0.f;
  }
  const bool vec = f + 4 <= cols && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
#pragma unroll
  for (int sweep = 0; sweep < 4; ++sweep) {
    const int64_t r = s0 + ty + 16 * sweep;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      const float* p = src + r * ld + f;
      if (vec) {
        const f32x4 x = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = f + e < cols ? p[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = sc[e] != 0.f ? aff_fma(v[e], sh[e], sc[e], nls[e]) : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[ty + 16 * sweep][4 * tx + e] = v[e];
  }
  __syncthreads();
  float m1 = -1.f;
  asm volatile("" : "+v"(m1));
  const int fl = threadIdx.x >> 2, q = threadIdx.x & 3;      // feature of the tile, quarter of its 64 samples
  u32x4 hi[2], lo[2];
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    const float a0 = tile[16 * q + 2 * h][fl], a1 = tile[16 * q + 2 * h + 1][fl];
    const fp16x2_t a = cvt_pk_rn(a0, a1);
    const fp16x2_t b = cvt_pk_rn(__builtin_fmaf((float)a[0], m1, a0), __builtin_fmaf((float)a[1], m1, a1));
    hi[h >> 2][h & 3] = __builtin_bit_cast(unsigned, a);
    lo[h >> 2][h & 3] = __builtin_bit_cast(unsigned, b);
  }
  char* line = reinterpret_cast<char*>(planes) + (f0 + fl) * (4 * kpad) + ((s0 >> 5) + (q >> 1)) * 128 + 32 * (q & 1);
  *reinterpret_cast<u32x4*>(line) = hi[0];
  *reinterpret_cast<u32x4*>(line + 16) = hi[1];
  *reinterpret_cast<u32x4*>(line + 64) = lo[0];
  *reinterpret_cast<u32x4*>(line + 80) = lo[1];
}

// work item of gram_nt_kernel: tile (bi, bj), first stage, number of stages (EVEN), output slot (tile-major: slot = tile * S + split)
struct GramItem {
  int bi, bj, st0, nst, slot, pad0, pad1, pad2;
};

#define EOFX_GLDS16(gp, lp)                                                                               \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned*)(gp), \
                                   (__attribute__((address_space(3))) unsigned*)(lp), 16, 0, 0)

// C_slot[256 x 256] = A_tile (256 rows of PA from row 256 bi) . B_tile^T (256 rows of PB from row 256 bj) over the item's
// stages.  pitch: bytes per row of the planes (= 4 kpad).  Cp: [slots][256][256] float32.
// out_scale: exact inverse of the two operands' power-of-two scales.
__global__ __launch_bounds__(512, 2) void gram_nt_kernel(const _Float16* __restrict__ PA, const _Float16* __restrict__ PB,
                                                          int64_t pitch, const GramItem* __restrict__ items,
                                                          float* __restrict__ Cp, float out_scale) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * GR_STAGE_BYTES];   // the ONLY LDS object (hipcc: a second one drains vmcnt)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const GramItem it = items[blockIdx.x];
  if (it.nst <= 0) return;
  const int nst = it.nst;

  // ---- LDS-DMA side: wave w stages rows [32 w, 32 w + 32) of both operand tiles, 4 instructions of 8 rows x 128 B each.
  // Lane (rr = lane / 8, c = lane % 8) lands at row-major position [row][chunk c]; it FETCHES chunk c ^ f(row),
  // f(row) = (row / 2) % 8, so that LDS chunk position c of row `row` holds source chunk c ^ f(row).
  const int rr = lane >> 3, cc = lane & 7;
  unsigned goff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 32 * wave + 8 * i + rr;
    goff[i] = (unsigned)row * (unsigned)pitch + 16u * (unsigned)(cc ^ ((row >> 1) & 7));
  }
  const char* Ag = reinterpret_cast<const char*>(PA) + (int64_t)it.bi * GR_BM * pitch + (int64_t)it.st0 * 128;
  const char* Bg = reinterpret_cast<const char*>(PB) + (int64_t)it.bj * GR_BM * pitch + (int64_t)it.st0 * 128;
  char* const lw = lds + (32 * wave) * 128;   // wave-uniform
#define GR_ISSUE(t, buf)                                                                \
  do {                                                                                  \
    const char* ag_ = Ag + (int64_t)(t) * 128;                                          \
    const char* bg_ = Bg + (int64_t)(t) * 128;                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                     \
      EOFX_GLDS16(ag_ + goff[i], lw + (buf) * GR_STAGE_BYTES + i * 1024);               \
      EOFX_GLDS16(bg_ + goff[i], lw + (buf) * GR_STAGE_BYTES + GR_OP_BYTES + i * 1024); \
    }                                                                                   \
  } while (0)

  // ---- MFMA side: wave (wr, wc) owns rows [128 wr, +128) x columns [64 wc, +64) of the tile
  const int wr = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, g = lane >> 5;
  const int fx = (l31 >> 1) & 7;
  // fragment of (plane pl, k-step ks): source chunk q = 4 pl + 2 ks + g of row l31 (+ 32 rt), at position q ^ fx:
  // byte address = lane part ^ (64 pl + 32 ks)  (the lane part carries g ^ fx in bit 4 and fx's upper bits in bits 5-6)
  const int lane_c = 16 * (((g ^ fx) & 1) + (fx & 6));
  const int a_lane = (128 * wr + l31) * 128 + lane_c;
  const int b_lane = GR_OP_BYTES + (64 * wc + l31) * 128 + lane_c;

  f32x16 acc[4][2];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;

  {
    // Every fragment set is read one quarter-stage AHEAD of its 12 MFMAs (a quarter = one k-step x two of the four row
    // tiles), the first quarter of the next stage right after the barrier, under the last quarter's MFMAs.  The reads and
    // their waits are written by hand: hipcc waits lgkmcnt(0) in front of every MFMA group -- it loses count of the reads
    // across the loop edge -- and so waits for the reads it has JUST issued; here the wait in front of a group leaves exactly
    // the younger reads in flight (LDS returns in order).  [Measured, config-3 Gram, same box: compiler-scheduled reads
    // 8.87 ms, quarter-stage prefetch with the compiler's waits 8.49 ms, this 8.15 ms; s_setprio around the MFMA groups: no
    // difference -- profiles/r04_gram_probe.txt.]
    f16x8 Aa[2][2], Ab[2][2], Ba[2][2], Bb[2][2];
    int ax[2][2], bx[2][2];     // [plane][k-step] byte addresses of this lane's fragments in buffer 0
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        ax[pl][ks] = a_lane ^ (64 * pl + 32 * ks);
        bx[pl][ks] = b_lane ^ (64 * pl + 32 * ks);
      }
#define GR_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(off) : "memory")
#define GR_RA2(D, boff, ks, half)                                                 \
  _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                              \
    const int ad_ = ax[pl][ks] + (boff);                                          \
    GR_DSR(D[0][pl], ad_, (2 * (half)) * 4096);                                   \
    GR_DSR(D[1][pl], ad_, (2 * (half) + 1) * 4096);                               \
  }
#define GR_RB2(D, boff, ks)                                                       \
  _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                              \
    const int ad_ = bx[pl][ks] + (boff);                                          \
    GR_DSR(D[0][pl], ad_, 0);                                                     \
    GR_DSR(D[1][pl], ad_, 4096);                                                  \
  }
#define GR_MM(A_, B_, half)                                                                                       \
  do {                                                                                                            \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                \
        acc[2 * (half) + r][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[r][1], B_[ct][0], acc[2 * (half) + r][ct], 0, 0, 0); \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                \
        acc[2 * (half) + r][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[r][0], B_[ct][1], acc[2 * (half) + r][ct], 0, 0, 0); \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                \
        acc[2 * (half) + r][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[r][0], B_[ct][0], acc[2 * (half) + r][ct], 0, 0, 0); \
  } while (0)
#define GR_FENCE() __builtin_amdgcn_sched_barrier(0)
#define GR_LGKM(n)                                             \
  do {                                                         \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory");    \
    GR_FENCE();                                                \
  } while (0)
#define GR_STAGE2(boff, noff, tnext)      /* boff / noff: byte offsets of this / the other buffer */ \
  do {                                                               \
    GR_ISSUE(tnext, (noff) / GR_STAGE_BYTES);                        \
    GR_RA2(Ab, boff, 0, 1);              /* 4 reads */               \
    GR_LGKM(4);                          /* Aa, Ba of this stage */  \
    GR_MM(Aa, Ba, 0);                                                \
    GR_FENCE();                                                      \
    GR_RA2(Aa, boff, 1, 0);              /* + 8 */                   \
    GR_RB2(Bb, boff, 1);                                             \
    GR_LGKM(8);                          /* Ab */                    \
    GR_MM(Ab, Ba, 1);                                                \
    GR_FENCE();                                                      \
    GR_RA2(Ab, boff, 1, 1);              /* + 4 */                   \
    GR_LGKM(4);                          /* Aa, Bb */                \
    GR_MM(Aa, Bb, 0);                                                \
    GR_FENCE();                                                      \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_s_barrier();                                    \
    GR_FENCE();                                                      \
    GR_RA2(Aa, noff, 0, 0);              /* 8 reads of the next stage */ \
    GR_RB2(Ba, noff, 0);                                             \
    GR_FENCE();                                                      \
    GR_MM(Ab, Bb, 1);                                                \
    GR_FENCE();                                                      \
  } while (0)
    GR_ISSUE(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    GR_FENCE();
    GR_RA2(Aa, 0, 0, 0);
    GR_RB2(Ba, 0, 0);
    for (int t = 0; t < nst; t += 2) {
      GR_STAGE2(0, GR_STAGE_BYTES, t + 1);
      GR_STAGE2(GR_STAGE_BYTES, 0, (t + 2 < nst ? t + 2 : nst - 1));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef GR_STAGE2
#undef GR_RA2
#undef GR_RB2
#undef GR_DSR
#undef GR_MM
#undef GR_FENCE
#undef GR_LGKM
  }
#undef GR_ISSUE

  // D of the 32x32 MFMA: column = lane % 32, row = (r % 4) + 8 (r / 4) + 4 (lane / 32)
  float* out = Cp + (int64_t)it.slot * (GR_BM * GR_BM);
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 128 * wr + 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * g;
        out[row * GR_BM + 64 * wc + 32 * ct + l31] = acc[rt][ct][r] * out_scale;
      }
}

// G[n_pad x n_pad] (ld) from the split partials of the upper-triangle tiles: fixed-order sum over the S slots of a tile,
// written to block (bi, bj) and, transposed, to block (bj, bi) -- G is symmetric bit for bit.  Inside a diagonal tile the
// kernel's two triangles differ in the last bit (hl and lh meet the accumulator in another order), so only its upper
// triangle is used there too.  grid = (tiles, 16): one 64 x 64 sub-block per workgroup.  tiles: {bi, bj} in slot order.
__global__ __launch_bounds__(256) void gram_finish_kernel(const float* __restrict__ Cp, const int2* __restrict__ tiles, int S,
                                                           float* __restrict__ G, int64_t ld) {
  __shared__ float tr[64][65];
  const int tile = blockIdx.x, sb = blockIdx.y;
  const int bi = tiles[tile].x, bj = tiles[tile].y;
  const int sr = sb >> 2, sc = sb & 3;
  if (bi == bj && sr > sc) return;      // written as the mirror image of sub-block (sc, sr)   (uniform)
  const int r0 = 64 * sr, c0 = 64 * sc;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;     // float4 column, row (16 rows per sweep)
  const float* base = Cp + (int64_t)tile * S * (GR_BM * GR_BM);
#pragma unroll
  for (int sweep = 0; sweep < 4; ++sweep) {
    const int r = r0 + 16 * sweep + ty, c = c0 + 4 * tx;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) v += *reinterpret_cast<const f32x4*>(base + (int64_t)s * (GR_BM * GR_BM) + r * GR_BM + c);
    tr[16 * sweep + ty][4 * tx + 0] = v[0];
    tr[16 * sweep + ty][4 * tx + 1] = v[1];
    tr[16 * sweep + ty][4 * tx + 2] = v[2];
    tr[16 * sweep + ty][4 * tx + 3] = v[3];
  }
  __syncthreads();
  const bool diag = bi == bj && sr == sc;     // the sub-block straddles the diagonal: element (r, c) with r > c := (c, r)
  const int qx = threadIdx.x & 63, qy = threadIdx.x >> 6;
#pragma unroll
  for (int sweep = 0; sweep < 16; ++sweep) {
    const int rr = 4 * sweep + qy;
    G[((int64_t)bi * GR_BM + r0 + rr) * ld + (int64_t)bj * GR_BM + c0 + qx] = (diag && rr > qx) ? tr[qx][rr] : tr[rr][qx];
  }
  if (diag) return;
#pragma unroll
  for (int sweep = 0; sweep < 16; ++sweep) {
    const int cc = 4 * sweep + qy;      // column of the sub-block = row of the mirrored block
    G[((int64_t)bj * GR_BM + c0 + cc) * ld + (int64_t)bi * GR_BM + r0 + qx] = tr[qx][cc];
  }
}

// C[nrows x ncols] (ldc) from the tile partials of a general (non-symmetric) NT product: fixed-order sum over the S slots of a
// tile.  grid = (tiles, 16): one 64 x 64 sub-block per workgroup.
__global__ __launch_bounds__(256) void nt_finish_kernel(const float* __restrict__ Cp, const int2* __restrict__ tiles, int S,
                                                         float* __restrict__ C, int64_t ldc, int64_t nrows, int ncols) {
  const int tile = blockIdx.x, sb = blockIdx.y;
  const int bi = tiles[tile].x, bj = tiles[tile].y;
  const int r0 = 64 * (sb >> 2), c0 = 64 * (sb & 3);
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const float* base = Cp + (int64_t)tile * S * (GR_BM * GR_BM);
#pragma unroll
  for (int sweep = 0; sweep < 4; ++sweep) {
    const int r = r0 + 16 * sweep + ty, c = c0 + 4 * tx;
    const int64_t gr = (int64_t)bi * GR_BM + r;
    const int gc = bj * GR_BM + c;
    if (gr >= nrows || gc >= ncols) continue;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) v += *reinterpret_cast<const f32x4*>(base + (int64_t)s * (GR_BM * GR_BM) + r * GR_BM + c);
    if (gc + 4 <= ncols) {
      *reinterpret_cast<f32x4*>(C + gr * ldc + gc) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (gc + e < ncols) C[gr * ldc + gc + e] = v[e];
    }
  }
}

}  // namespace eofx

// ---- host side: the work list of gram_nt_kernel -----------------------------------------------------------------
#include <algorithm>
#include <vector>

namespace eofx {

struct GramPlan {
  int nti = 0, ntj = 0, T = 0, S = 1, st_per_split = 0, grid = 0;
  std::vector<GramItem> items;   // [grid]: workgroup b -> item (XCD b % 8 walks its own patch of tiles split by split)
  std::vector<int2> tiles;       // [T]: slot order
};

static inline unsigned gram_morton(unsigned i, unsigned j) {
  unsigned m = 0;
  for (int b = 0; b < 12; ++b) m |= ((i >> b) & 1u) << (2 * b + 1) | ((j >> b) & 1u) << (2 * b);
  return m;
}

// sym: only tiles bi <= bj (C = A A^T).  nst_total: stages (32 features each) of the contraction.
// S <= 0: chosen here (rounds of 32 workgroups per XCD x stages per split, ~40 stages of fixed cost per item).
static inline void gram_plan_build(int nti, int ntj, bool sym, int nst_total, int S, GramPlan& pl) {
  pl.nti = nti;
  pl.ntj = ntj;
  pl.tiles.clear();
  for (int i = 0; i < nti; ++i)
    for (int j = sym ? i : 0; j < ntj; ++j) pl.tiles.push_back(int2{i, j});
  std::sort(pl.tiles.begin(), pl.tiles.end(), [](const int2& a, const int2& b) {
    return gram_morton((unsigned)a.x, (unsigned)a.y) < gram_morton((unsigned)b.x, (unsigned)b.y);
  });
  const int T = (int)pl.tiles.size();
  pl.T = T;
  int lo[9];
  for (int x = 0; x <= 8; ++x) lo[x] = (int)((int64_t)T * x / 8);
  int cmax = 0;
  for (int x = 0; x < 8; ++x) cmax = std::max(cmax, lo[x + 1] - lo[x]);
  if (S <= 0) {
    double best = 1e300;
    for (int s = 1; s <= 16 && s <= nst_total; ++s) {
      const int sps = ((nst_total + s - 1) / s + 1) / 2 * 2;
      const int s_eff = (nst_total + sps - 1) / sps;
      if (s_eff != s) continue;
      const double rounds = (double)((cmax * s + 31) / 32);
      const double cost = rounds * (sps + 40.0) + 2.0 * s;
      if (cost < best * 0.97) {
        best = cost;
        S = s;
      }
    }
  }
  const int sps = ((nst_total + S - 1) / S + 1) / 2 * 2;     // even (nst_total is: kpad % 64 == 0)
  S = (nst_total + sps - 1) / sps;
  pl.S = S;
  pl.st_per_split = sps;
  const int W = cmax * S;
  pl.grid = 8 * W;
  pl.items.assign((size_t)pl.grid, GramItem{0, 0, 0, 0, 0, 0, 0, 0});
  for (int x = 0; x < 8; ++x) {
    const int cnt = lo[x + 1] - lo[x];
    for (int s = 0; s < S; ++s)
      for (int t = 0; t < cnt; ++t) {
        const int w = s * cnt + t, tile = lo[x] + t;
        const int st0 = s * sps, nst = std::min(sps, nst_total - st0);
        pl.items[(size_t)8 * w + x] = GramItem{pl.tiles[tile].x, pl.tiles[tile].y, st0, nst, tile * S + s, 0, 0, 0};
      }
  }
}

}  // namespace eofx
