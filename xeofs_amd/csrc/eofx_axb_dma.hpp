// eofx_axb_dma.hpp -- the in-place sample-side product with the B slab moved by LDS-DMA (round 4).
//
// How an asynchronous pipeline is written in HIP C++ here, and why it looks the way it does:
//  * hipcc (ROCm 7.2) files an LDS-DMA load and an ordinary global load under different event types of the same counter and
//    then waits vmcnt(0) in front of every use of a loaded register while both are pending; and it makes every LDS access it
//    can see wait for an LDS-DMA in flight.  So no load and no LDS access of the pair loop is visible to it: they are inline
//    assembly, and every wait is counted by hand (the issue order is the same on every path, see the loop).
//  * A register the compiler manages cannot be the target of such a load: it copies "defined" values wherever it likes, also
//    between the load and its wait (seen in the assembly of the first attempt).  The kernel is therefore compiled with
//    amdgpu_num_vgpr(176) -- the allocator stays below v176 -- and the loads name v176 .. v252 in their text.
//  tools/probes/axb_probe.hip compares it bit for bit with axb_f16_kernel; the library's tests run over it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "eofx_kernels.hpp"

namespace eofx {

// ---------------------------------------------------------------------------------
// axb_f16 with the B slab taken out of the waves' instruction streams (round 4).  profiles/r04_axb_bcost.txt priced the
// hand-over of the B slab in axb_f16_kernel (loads, conversion, LDS stores, their waits) at 5.8 % of the kernel; here
//   * axb_bsplit_kernel converts the panel ONCE into two fp16 planes in the exact order the LDS buffer holds them
//     ([column block][feature pair][plane][k-group of 8][column slot, XOR-swizzled][8 halves]: 16 KiB per 64 features), and
//   * axb_f16_dma_kernel moves a pair's 16 KiB with 16 LDS-DMA instructions per workgroup (global_load_lds_dwordx4: no
//     VGPRs, no VALU, no LDS store instructions), one pair ahead.
// Everything else is axb_f16_kernel: same A stream, same map, same split, same MFMA order -> the SAME BITS in C.
// ONE LDS object, touched only from inline assembly (stores of the converted A, fragment reads, their lgkmcnt waits); the A
// and map-triple loads are inline assembly into v176 .. v251, the pair id of a masked matrix into v252; the DMA itself is
// the builtin (with nothing else of the compiler's in flight it adds no waits of its own).  See the file header for why.
// ---------------------------------------------------------------------------------
constexpr int AXB_PAIR_BYTES = 2 * 8 * 64 * 16;   // one feature pair (64 features) x 64 columns, two fp16 planes

// planes[cb][P][plane][g][slot][8]: element (k = 64 P + 8 g + t, column 64 cb + c) at slot c ^ ((c >> 3) & 7), half t.
// Columns >= L are zero.  grid = (K_all / 32, column blocks), block = 256 = 4 k-groups x 64 columns.
__global__ __launch_bounds__(256) void axb_bsplit_kernel(const float* __restrict__ B, int ldb, int L, int64_t K_all,
                                                          const float* __restrict__ b_absmax, _Float16* __restrict__ planes) {
  const int c = threadIdx.x & 63, cb = blockIdx.y;
  const int64_t kg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (kg * 8 >= K_all) return;
  const float b_scale = f16_scale_for(*b_absmax);
  float m1 = -1.f;
  asm volatile("" : "+v"(m1));
  const int col = 64 * cb + c;
  u32x4 hi, lo;
#pragma unroll
  for (int t = 0; t < 8; t += 2) {
    const float v0 = (col < L ? B[(kg * 8 + t) * ldb + col] : 0.f) * b_scale;
    const float v1 = (col < L ? B[(kg * 8 + t + 1) * ldb + col] : 0.f) * b_scale;
    const fp16x2_t h = cvt_pk_rn(v0, v1);
    fp16x2_t l;
    l[0] = (__fp16)__builtin_fmaf((float)h[0], m1, v0);
    l[1] = (__fp16)__builtin_fmaf((float)h[1], m1, v1);
    hi[t >> 1] = __builtin_bit_cast(unsigned, h);
    lo[t >> 1] = __builtin_bit_cast(unsigned, l);
  }
  const int64_t P = kg >> 3;
  const int g = (int)(kg & 7), slot = c ^ ((c >> 3) & 7);
  char* base = reinterpret_cast<char*>(planes) + ((int64_t)cb * (K_all / AXB_KG) + P) * AXB_PAIR_BYTES + (g * 64 + slot) * 16;
  *reinterpret_cast<u32x4*>(base) = hi;
  *reinterpret_cast<u32x4*>(base + 8192) = lo;
}

template <int NQ, int DBG = 0, bool MASK = false>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(176))) void axb_f16_dma_kernel(const float* __restrict__ A, int64_t lda, int a_rows,
                                                              int64_t a_cols, const float* __restrict__ aff, int64_t aff_ld,
                                                              const _Float16* __restrict__ planes, int64_t pairs_all,
                                                              float* __restrict__ C, int ldc, int64_t c_rows, int64_t K,
                                                              int64_t k_per_split, int splits, int row_tiles, int col_base,
                                                              float a_scale, const float* __restrict__ b_absmax,
                                                              const int* __restrict__ act = nullptr) {
  // the ONLY LDS object: [0, 32 KiB) A staging, 4 waves x [plane][row 64][64 bytes]; [32 KiB, 64 KiB) two B buffers
  __shared__ __attribute__((aligned(1024))) char lds[65536];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, g = lane >> 4;
  const int lr = lane >> 3, lc = lane & 7;
  const int a_wc = 8 * ((lc >> 1) ^ ((lr >> 1) & 3)) + 4 * (lc & 1);   // as in axb_f16_kernel (halves)
  const int a_rc = 8 * (g ^ ((ln >> 1) & 3));
  const int slot_ = splits > 1 ? (int)blockIdx.x >> 3 : (int)blockIdx.x;
  const int split = splits > 1 ? ((int)blockIdx.x & 7) + 8 * (slot_ / row_tiles) : 0;
  if (split >= splits) return;
  const int r0 = (slot_ % row_tiles) * AXB_BM + wave * 64;
  const bool live = r0 < a_rows;
  const bool full = r0 + 64 <= a_rows;
  const unsigned ldab = (unsigned)lda * 4u;
  const unsigned lrl = (unsigned)lr * ldab;
  const bool listed = MASK && act != nullptr;
  const int64_t kb_ = (int64_t)split * k_per_split;
  const int64_t ke = (kb_ + k_per_split < K) ? kb_ + k_per_split : K;
  const int nslab = (int)((ke - kb_) / AXB_KC);
  const int64_t kb = (MASK && listed) ? 0 : kb_;
  const int* const actp = (MASK && listed) ? act + kb_ / AXB_KG : nullptr;
#define EOFX_PAIR(i) (listed ? actp[(i)] : (i))
  const int bcol0 = col_base + blockIdx.y * 64;
  const float b_scale = f16_scale_for(*b_absmax);
  const float out_scale = 1.f / (a_scale * b_scale);

  f32x4 acc[4][NQ];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  const char* const Ab = reinterpret_cast<const char*>(A + (int64_t)(live ? r0 : 0) * lda);
  const int arow0 = live ? r0 : 0;
  const float* const aff1 = aff + aff_ld;
  const float* const aff2 = aff + 2 * aff_ld;
  const int fo = (int)kb + 4 * lc;
  // B: this column block's planes from the split's first pair on; wave w moves rows 4 w .. 4 w + 3 of the 16 KiB
  const char* const Pb = reinterpret_cast<const char*>(planes) + ((int64_t)(bcol0 >> 6) * pairs_all + kb / AXB_KG) * AXB_PAIR_BYTES +
                         (wave * 4) * 1024 + lane * 16;
  char* const bw = lds + 32768 + (wave * 4) * 1024;      // wave-uniform LDS destination of its rows in buffer 0
  // LDS byte addresses of this lane (inline assembly below; the generic address of a __shared__ object's first byte is 0
  // relative to the object only, so take the real offset)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned as_w = lds0 + wave * 8192 + lr * 64 + a_wc * 2;
  const unsigned as_r = lds0 + wave * 8192 + ln * 64 + a_rc * 2;
  unsigned bs_r[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int col_ = 16 * q + ln;
    bs_r[q] = lds0 + 32768 + g * 1024 + (col_ ^ ((col_ >> 3) & 7)) * 16;
  }

  static_assert(!(DBG & 4), "the wait counts assume the map-triple loads");
  float m1 = -1.f;
  asm volatile("" : "+v"(m1));
  // ---- A and the map triples land in v[176 .. 251] (a0[u] = v[176 + 4u ..], a1[u] = v[208 + 4u ..], triples v[240 .. 251]) ----
  // The loads name these registers in their text and list them as clobbers (which is what puts them into the kernel's
  // register count); the conversion's first instruction reads a row group in place, the masked variant and the triples
  // take four v_mov after the wait.
#define EOFX_ALD_(lo, hi, c0, c1, c2, c3, voff, sbase, mod)                                                    \
  asm volatile("global_load_dwordx4 v[" #lo ":" #hi "], %0, %1" mod : : "v"(voff), "s"(sbase) : "memory", c0, c1, c2, c3)
#define EOFX_ARD_(dst, r0, r1, r2, r3)                                                                          \
  asm volatile("v_mov_b32 %0, v" #r0 "\n\tv_mov_b32 %1, v" #r1 "\n\tv_mov_b32 %2, v" #r2                              \
               "\n\tv_mov_b32 %3, v" #r3 : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]) : : "memory")
#define EOFX_ALD_S0(voff, sbase, mod) EOFX_ALD_(176, 179, "v176", "v177", "v178", "v179", voff, sbase, mod)
#define EOFX_ARD_S0(dst) EOFX_ARD_(dst, 176, 177, 178, 179)
#define EOFX_ALD_S1(voff, sbase, mod) EOFX_ALD_(180, 183, "v180", "v181", "v182", "v183", voff, sbase, mod)
#define EOFX_ARD_S1(dst) EOFX_ARD_(dst, 180, 181, 182, 183)
#define EOFX_ALD_S2(voff, sbase, mod) EOFX_ALD_(184, 187, "v184", "v185", "v186", "v187", voff, sbase, mod)
#define EOFX_ARD_S2(dst) EOFX_ARD_(dst, 184, 185, 186, 187)
#define EOFX_ALD_S3(voff, sbase, mod) EOFX_ALD_(188, 191, "v188", "v189", "v190", "v191", voff, sbase, mod)
#define EOFX_ARD_S3(dst) EOFX_ARD_(dst, 188, 189, 190, 191)
#define EOFX_ALD_S4(voff, sbase, mod) EOFX_ALD_(192, 195, "v192", "v193", "v194", "v195", voff, sbase, mod)
#define EOFX_ARD_S4(dst) EOFX_ARD_(dst, 192, 193, 194, 195)
#define EOFX_ALD_S5(voff, sbase, mod) EOFX_ALD_(196, 199, "v196", "v197", "v198", "v199", voff, sbase, mod)
#define EOFX_ARD_S5(dst) EOFX_ARD_(dst, 196, 197, 198, 199)
#define EOFX_ALD_S6(voff, sbase, mod) EOFX_ALD_(200, 203, "v200", "v201", "v202", "v203", voff, sbase, mod)
#define EOFX_ARD_S6(dst) EOFX_ARD_(dst, 200, 201, 202, 203)
#define EOFX_ALD_S7(voff, sbase, mod) EOFX_ALD_(204, 207, "v204", "v205", "v206", "v207", voff, sbase, mod)
#define EOFX_ARD_S7(dst) EOFX_ARD_(dst, 204, 205, 206, 207)
#define EOFX_ALD_S8(voff, sbase, mod) EOFX_ALD_(208, 211, "v208", "v209", "v210", "v211", voff, sbase, mod)
#define EOFX_ARD_S8(dst) EOFX_ARD_(dst, 208, 209, 210, 211)
#define EOFX_ALD_S9(voff, sbase, mod) EOFX_ALD_(212, 215, "v212", "v213", "v214", "v215", voff, sbase, mod)
#define EOFX_ARD_S9(dst) EOFX_ARD_(dst, 212, 213, 214, 215)
#define EOFX_ALD_S10(voff, sbase, mod) EOFX_ALD_(216, 219, "v216", "v217", "v218", "v219", voff, sbase, mod)
#define EOFX_ARD_S10(dst) EOFX_ARD_(dst, 216, 217, 218, 219)
#define EOFX_ALD_S11(voff, sbase, mod) EOFX_ALD_(220, 223, "v220", "v221", "v222", "v223", voff, sbase, mod)
#define EOFX_ARD_S11(dst) EOFX_ARD_(dst, 220, 221, 222, 223)
#define EOFX_ALD_S12(voff, sbase, mod) EOFX_ALD_(224, 227, "v224", "v225", "v226", "v227", voff, sbase, mod)
#define EOFX_ARD_S12(dst) EOFX_ARD_(dst, 224, 225, 226, 227)
#define EOFX_ALD_S13(voff, sbase, mod) EOFX_ALD_(228, 231, "v228", "v229", "v230", "v231", voff, sbase, mod)
#define EOFX_ARD_S13(dst) EOFX_ARD_(dst, 228, 229, 230, 231)
#define EOFX_ALD_S14(voff, sbase, mod) EOFX_ALD_(232, 235, "v232", "v233", "v234", "v235", voff, sbase, mod)
#define EOFX_ARD_S14(dst) EOFX_ARD_(dst, 232, 233, 234, 235)
#define EOFX_ALD_S15(voff, sbase, mod) EOFX_ALD_(236, 239, "v236", "v237", "v238", "v239", voff, sbase, mod)
#define EOFX_ARD_S15(dst) EOFX_ARD_(dst, 236, 237, 238, 239)
#define EOFX_ALD_S16(voff, sbase, mod) EOFX_ALD_(240, 243, "v240", "v241", "v242", "v243", voff, sbase, mod)
#define EOFX_ARD_S16(dst) EOFX_ARD_(dst, 240, 241, 242, 243)
#define EOFX_ALD_S17(voff, sbase, mod) EOFX_ALD_(244, 247, "v244", "v245", "v246", "v247", voff, sbase, mod)
#define EOFX_ARD_S17(dst) EOFX_ARD_(dst, 244, 245, 246, 247)
#define EOFX_ALD_S18(voff, sbase, mod) EOFX_ALD_(248, 251, "v248", "v249", "v250", "v251", voff, sbase, mod)
#define EOFX_ARD_S18(dst) EOFX_ARD_(dst, 248, 249, 250, 251)
#define EOFX_CAT_(a, b) a##b
#define EOFX_CAT(a, b) EOFX_CAT_(a, b)
#define EOFX_SLOT_0_0 0
#define EOFX_SLOT_0_1 1
#define EOFX_SLOT_0_2 2
#define EOFX_SLOT_0_3 3
#define EOFX_SLOT_0_4 4
#define EOFX_SLOT_0_5 5
#define EOFX_SLOT_0_6 6
#define EOFX_SLOT_0_7 7
#define EOFX_SLOT_1_0 8
#define EOFX_SLOT_1_1 9
#define EOFX_SLOT_1_2 10
#define EOFX_SLOT_1_3 11
#define EOFX_SLOT_1_4 12
#define EOFX_SLOT_1_5 13
#define EOFX_SLOT_1_6 14
#define EOFX_SLOT_1_7 15
#define EOFX_ALD_A(set, u, voff) do { if (DBG & 8) EOFX_CAT(EOFX_ALD_S, EOFX_CAT(EOFX_SLOT_##set##_, u))(voff, Ab, ""); \
                                      else EOFX_CAT(EOFX_ALD_S, EOFX_CAT(EOFX_SLOT_##set##_, u))(voff, Ab, " nt"); } while (0)
#define EOFX_ARD_A(set, u, dst) EOFX_CAT(EOFX_ARD_S, EOFX_CAT(EOFX_SLOT_##set##_, u))(dst)
#define EOFX_VPAIR_S0_0 "v[176:177]"
#define EOFX_VPAIR_S0_1 "v[178:179]"
#define EOFX_VPAIR_S1_0 "v[180:181]"
#define EOFX_VPAIR_S1_1 "v[182:183]"
#define EOFX_VPAIR_S2_0 "v[184:185]"
#define EOFX_VPAIR_S2_1 "v[186:187]"
#define EOFX_VPAIR_S3_0 "v[188:189]"
#define EOFX_VPAIR_S3_1 "v[190:191]"
#define EOFX_VPAIR_S4_0 "v[192:193]"
#define EOFX_VPAIR_S4_1 "v[194:195]"
#define EOFX_VPAIR_S5_0 "v[196:197]"
#define EOFX_VPAIR_S5_1 "v[198:199]"
#define EOFX_VPAIR_S6_0 "v[200:201]"
#define EOFX_VPAIR_S6_1 "v[202:203]"
#define EOFX_VPAIR_S7_0 "v[204:205]"
#define EOFX_VPAIR_S7_1 "v[206:207]"
#define EOFX_VPAIR_S8_0 "v[208:209]"
#define EOFX_VPAIR_S8_1 "v[210:211]"
#define EOFX_VPAIR_S9_0 "v[212:213]"
#define EOFX_VPAIR_S9_1 "v[214:215]"
#define EOFX_VPAIR_S10_0 "v[216:217]"
#define EOFX_VPAIR_S10_1 "v[218:219]"
#define EOFX_VPAIR_S11_0 "v[220:221]"
#define EOFX_VPAIR_S11_1 "v[222:223]"
#define EOFX_VPAIR_S12_0 "v[224:225]"
#define EOFX_VPAIR_S12_1 "v[226:227]"
#define EOFX_VPAIR_S13_0 "v[228:229]"
#define EOFX_VPAIR_S13_1 "v[230:231]"
#define EOFX_VPAIR_S14_0 "v[232:233]"
#define EOFX_VPAIR_S14_1 "v[234:235]"
#define EOFX_VPAIR_S15_0 "v[236:237]"
#define EOFX_VPAIR_S15_1 "v[238:239]"
#define EOFX_VPAIR(set, u, h) EOFX_CAT(EOFX_CAT(EOFX_VPAIR_S, EOFX_CAT(EOFX_SLOT_##set##_, u)), _##h)
#define EOFX_VMWAIT(n) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(n) : "memory")
#define EOFX_DMA_B(pair, buf)    /* 4 of the pair's 16 rows of 1 KiB: lane l lands at byte 16 l of the row */ \
  do {                                                                                                 \
    if (!(DBG & 64)) {                                                                                 \
      const char* src_ = Pb + (int64_t)(pair) * AXB_PAIR_BYTES;                                        \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                 \
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned*)(src_ + i_ * 1024), \
                                           (__attribute__((address_space(3))) unsigned*)(bw + (buf) * 16384 + i_ * 1024), 16, 0, 0); \
    }                                                                                                  \
  } while (0)
#define EOFX_LOAD_F(chunk)                                                                             \
  do {                                                                                                 \
    const unsigned fb_ = (unsigned)(fo + (chunk) * AXB_KC) * 4u;                                       \
    EOFX_ALD_S16(fb_, aff, "");                                                                        \
    EOFX_ALD_S17(fb_, aff1, "");                                                                       \
    EOFX_ALD_S18(fb_, aff2, "");                                                                       \
  } while (0)
#define EOFX_LOAD_A1(set, u, chunk)      /* one row group: u a literal */                              \
  do {                                                                                                 \
    const int ko_ = (chunk) * AXB_KC;                                                                  \
    const bool kin_ = fo + ko_ < a_cols;                                                               \
    const unsigned kof_ = kin_ ? (unsigned)(fo + ko_) * 4u : 0u;                                       \
    unsigned vo_;                                                                                      \
    if (full) vo_ = lrl + kof_ + (unsigned)(8 * (u)) * ldab;                                           \
    else vo_ = (unsigned)(arow0 + lr + 8 * (u) < a_rows ? lr + 8 * (u) : a_rows - 1 - arow0) * ldab + kof_; \
    EOFX_ALD_A(set, u, vo_);                                                                           \
  } while (0)
#define EOFX_LOAD_A_LO(set, chunk) do { EOFX_LOAD_A1(set, 0, chunk); EOFX_LOAD_A1(set, 1, chunk); EOFX_LOAD_A1(set, 2, chunk); EOFX_LOAD_A1(set, 3, chunk); } while (0)
#define EOFX_LOAD_A_HI(set, chunk) do { EOFX_LOAD_A1(set, 4, chunk); EOFX_LOAD_A1(set, 5, chunk); EOFX_LOAD_A1(set, 6, chunk); EOFX_LOAD_A1(set, 7, chunk); } while (0)
#define EOFX_DSW64(addr, val, off) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr), "v"(val), "n"(off) : "memory")
#define EOFX_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(off) : "memory")
#define EOFX_AXB_CONV_TAIL(h)        /* t_ = x - hi  ->  (x - hi) s - lo s, split, into hi_[h] / lo_[h] */ \
  do {                                                                                                 \
    f32x2 v_;                                                                                          \
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(v_) : "v"(t_), "v"(fs_[h]), "v"(fl_[h])); \
    const fp16x2_t p_ = cvt_pk_rn(v_[0], v_[1]);                                      \
    const fp16x2_t q_ = cvt_pk_rn(__builtin_fmaf((float)p_[0], m1, v_[0]),            \
                                                   __builtin_fmaf((float)p_[1], m1, v_[1]));            \
    hi_[h] = __builtin_bit_cast(unsigned, p_);                                                         \
    lo_[h] = __builtin_bit_cast(unsigned, q_);                                                         \
  } while (0)
#define EOFX_AXB_CONV1(set, u)      /* u: a literal (the LDS offsets and the landing registers are instruction text) */ \
  do {                                                                                                 \
    u32x2 hi_, lo_;                                                                                    \
    if constexpr (!MASK && !(DBG & 2)) {   /* the first operation reads the landing registers itself */  \
      f32x2 t_;                                                                                        \
      asm volatile("v_pk_add_f32 %0, " EOFX_VPAIR(set, u, 0) ", %1 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t_) : "v"(fh_[0]) : "memory"); \
      EOFX_AXB_CONV_TAIL(0);                                                                           \
      asm volatile("v_pk_add_f32 %0, " EOFX_VPAIR(set, u, 1) ", %1 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t_) : "v"(fh_[1]) : "memory"); \
      EOFX_AXB_CONV_TAIL(1);                                                                           \
    } else {                                                                                           \
      float av_[4];                                                                                    \
      EOFX_ARD_A(set, u, av_);                                                                         \
      if (DBG & 2) {                                                                                   \
        hi_[0] = __float_as_uint(av_[0]); hi_[1] = __float_as_uint(av_[1]);                            \
        lo_[0] = __float_as_uint(av_[2]); lo_[1] = __float_as_uint(av_[3]);                            \
      } else {                                                                                         \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                \
          const f32x2 x_ = {__uint_as_float(__float_as_uint(av_[2 * h]) & mk_[2 * h]),                 \
                            __uint_as_float(__float_as_uint(av_[2 * h + 1]) & mk_[2 * h + 1])};        \
          f32x2 t_;                                                                                    \
          asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t_) : "v"(x_), "v"(fh_[h]));   \
          EOFX_AXB_CONV_TAIL(h);                                                                       \
        }                                                                                              \
      }                                                                                                \
    }                                                                                                  \
    EOFX_DSW64(as_w, hi_, 512 * (u));                                                                  \
    EOFX_DSW64(as_w, lo_, 4096 + 512 * (u));                                                           \
  } while (0)
#define EOFX_AXB_CONVERT_LO(set) EOFX_AXB_CONV1(set, 0); EOFX_AXB_CONV1(set, 1); EOFX_AXB_CONV1(set, 2); EOFX_AXB_CONV1(set, 3);
#define EOFX_AXB_CONVERT_HI(set) EOFX_AXB_CONV1(set, 4); EOFX_AXB_CONV1(set, 5); EOFX_AXB_CONV1(set, 6); EOFX_AXB_CONV1(set, 7);
  // 32 rows x NQ column tiles.  Reads in two sets (lo A + hi B first: the first MFMA group needs only those), each wait
  // names the registers it releases so that no MFMA can be scheduled above it.  LDS operations of a wave complete in
  // order: lgkmcnt(2 + NQ) after both sets = the first set has landed (an outstanding scalar load only makes it stricter).
#define EOFX_AXB_BHI(boff, hs)       /* the hi-plane B fragments (issuing them ahead of the conversion was slower: 7.24 vs 7.13 ms) */ \
  _Pragma("unroll") for (int q = 0; q < NQ; ++q) { const unsigned ad_ = bs_r[q] + (boff); EOFX_DSR128(bfh_[q], ad_, 4096 * (hs)); }
#define EOFX_AXB_MFMA(jh, boff, hs)                                                                    \
  do {                                                                                                 \
    f16x8 af_[2][2], bfh_[NQ], bfl_[NQ];                                                                       \
    EOFX_DSR128(af_[0][1], as_r, 4096 + 1024 * (2 * (jh)));                                            \
    EOFX_DSR128(af_[1][1], as_r, 4096 + 1024 * (2 * (jh) + 1));                                        \
    EOFX_AXB_BHI(boff, hs)                                                                             \
    EOFX_DSR128(af_[0][0], as_r, 1024 * (2 * (jh)));                                                   \
    EOFX_DSR128(af_[1][0], as_r, 1024 * (2 * (jh) + 1));                                               \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) { const unsigned ad_ = bs_r[q] + (boff); EOFX_DSR128(bfl_[q], ad_, 8192 + 4096 * (hs)); } \
    if constexpr (NQ == 4)                                                                             \
      asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(af_[0][1]), "+v"(af_[1][1]), "+v"(bfh_[0]), "+v"(bfh_[1]), "+v"(bfh_[2]), "+v"(bfh_[3]) : : "memory"); \
    else                                                                                               \
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af_[0][1]), "+v"(af_[1][1]), "+v"(bfh_[0]), "+v"(bfh_[1]) : : "memory"); \
    if (DBG & 1) {                                                                                     \
      if constexpr (NQ == 4)                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af_[0][0]), "+v"(af_[1][0]), "+v"(bfl_[0]), "+v"(bfl_[1]), "+v"(bfl_[2]), "+v"(bfl_[3]) : : "memory"); \
      else                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af_[0][0]), "+v"(af_[1][0]), "+v"(bfl_[0]), "+v"(bfl_[1]) : : "memory"); \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) _Pragma("unroll") for (int j = 0; j < 2; ++j)     \
          _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[2 * (jh) + j][q][r] +=                     \
              (float)af_[j][0][r] + (float)af_[j][1][r + 4] + (float)bfh_[q][r] + (float)bfl_[q][r]; \
    } else {                                                                                           \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) _Pragma("unroll") for (int j = 0; j < 2; ++j)     \
          acc[2 * (jh) + j][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af_[j][1], bfh_[q], acc[2 * (jh) + j][q], 0, 0, 0); \
      if constexpr (NQ == 4)                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af_[0][0]), "+v"(af_[1][0]), "+v"(bfl_[0]), "+v"(bfl_[1]), "+v"(bfl_[2]), "+v"(bfl_[3]) : : "memory"); \
      else                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af_[0][0]), "+v"(af_[1][0]), "+v"(bfl_[0]), "+v"(bfl_[1]) : : "memory"); \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) _Pragma("unroll") for (int j = 0; j < 2; ++j)     \
          acc[2 * (jh) + j][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af_[j][0], bfl_[q], acc[2 * (jh) + j][q], 0, 0, 0); \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) _Pragma("unroll") for (int j = 0; j < 2; ++j)     \
          acc[2 * (jh) + j][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af_[j][0], bfh_[q], acc[2 * (jh) + j][q], 0, 0, 0); \
    }                                                                                                  \
  } while (0)
#define EOFX_SLAB(set, boff, hs, next_f, next_a, WF)                                                   \
  do {                                                                                                 \
    if (live) {                                                                                        \
      EOFX_VMWAIT(WF);                                                                                 \
      float f0_[4], f1_[4], f2_[4];                                                                    \
      EOFX_ARD_S16(f0_);                                                                               \
      EOFX_ARD_S17(f1_);                                                                               \
      EOFX_ARD_S18(f2_);                                                                               \
      EOFX_LOAD_F(next_f);      /* the triples are in ordinary registers now: their AccVGPRs take the next slab's */ \
      f32x2 fh_[2], fl_[2], fs_[2];                                                                    \
      _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                  \
        fh_[h] = f32x2{f0_[2 * h], f0_[2 * h + 1]};                                                    \
        fs_[h] = f32x2{f2_[2 * h], f2_[2 * h + 1]} * a_scale;                                          \
        fl_[h] = f32x2{f1_[2 * h], f1_[2 * h + 1]} * fs_[h];                                           \
      }                                                                                                \
      unsigned mk_[4];                                                                                 \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) mk_[e] = (MASK && f2_[e] == 0.f) ? 0u : 0xffffffffu; \
      (void)mk_;                                                                                       \
      EOFX_VMWAIT(23);                                                                                 \
      EOFX_AXB_CONVERT_LO(set)                                                                         \
      EOFX_LOAD_A_LO(set, next_a);                                                                     \
      EOFX_AXB_MFMA(0, boff, hs);                                                                      \
      EOFX_VMWAIT(23);                                                                                 \
      EOFX_AXB_CONVERT_HI(set)                                                                         \
      EOFX_LOAD_A_HI(set, next_a);                                                                     \
      EOFX_AXB_MFMA(1, boff, hs);                                                                      \
    }   /* a wave without rows loads nothing: its only job is its share of the B rows */              \
  } while (0)
  // The pair barrier.  Every thread issues 22 loads between the DMA of a pair and the barrier at the end of the pair before
  // (2 x (3 map triples + 8 A)), on every path: vmcnt(22) = the DMA has landed, the A prefetch stays in flight.  The A
  // staging is private to its wave, and every wave has waited for its last fragment reads: nothing else to publish.
#define EOFX_PAIR_BARRIER(n_younger)                                                                    \
  do {                                                                                                 \
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(n_younger) : "memory");                                  \
    __builtin_amdgcn_s_barrier();                                                                      \
    asm volatile("" ::: "memory");                                                                     \
  } while (0)

  if (nslab > 0) {   // nslab is even
    const int npair = nslab / 2;
    // issue order per pair, on every path: [D4, I1] | slab 0: F3, A4, A4 | slab 1: F3, A4, A4 | barrier.  I1 = the id of the
    // pair after the next one (active list of a masked matrix; a dummy load otherwise, so that the counts below hold for every
    // instantiation), into v252.  The prologue issues a0 before the first triples so that the first pair sees the same history
    // as every other pair.  Loads younger than the target at its wait: triples of slab 0: A8 + D4 + I1 = 13; of slab 1: A8 = 8;
    // each half of an A slab: 23; the DMA and the id at the barrier: 22.
#define EOFX_LOAD_ID(idx)                                                                              \
  do {                                                                                                 \
    const unsigned io_ = listed ? (unsigned)(idx) * 4u : 0u;                                           \
    const void* ib_ = listed ? (const void*)actp : (const void*)b_absmax;                              \
    asm volatile("global_load_dword v252, %0, %1" : : "v"(io_), "s"(ib_) : "memory", "v252");          \
  } while (0)
    int qc = EOFX_PAIR(0), p1 = EOFX_PAIR(npair > 1 ? 1 : 0);
    EOFX_DMA_B(qc, 0);
    EOFX_LOAD_ID(npair > 2 ? 2 : npair - 1);
    if (live) {
      EOFX_LOAD_A_LO(0, 2 * qc);
      EOFX_LOAD_A_HI(0, 2 * qc);
      EOFX_LOAD_F(2 * qc);
      EOFX_LOAD_A_LO(1, 2 * qc + 1);
      EOFX_LOAD_A_HI(1, 2 * qc + 1);
      EOFX_PAIR_BARRIER(19);   // 3 + 16 loads behind the first DMA and id
    } else {
      EOFX_PAIR_BARRIER(0);
    }
    for (int pr = 0; pr < npair; ++pr) {
      const unsigned boff = (unsigned)(pr & 1) * 16384u;
      int p2;   // the id that landed during the pair before (or in the prologue): pair pr + 2 (past the end: the last pair again)
      asm volatile("v_readfirstlane_b32 %0, v252" : "=s"(p2) : : "memory");
      if (!listed) p2 = pr + 2 < npair ? pr + 2 : npair - 1;
      const int c2 = 2 * p1;
      EOFX_DMA_B(p1, 1 - (pr & 1));
      EOFX_LOAD_ID(pr + 3 < npair ? pr + 3 : npair - 1);
      EOFX_SLAB(0, boff, 0, 2 * qc + 1, c2, 13);
      EOFX_SLAB(1, boff, 1, c2, c2 + 1, 8);
      if (live) EOFX_PAIR_BARRIER(22);
      else EOFX_PAIR_BARRIER(0);
      qc = p1;
      p1 = p2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-reads past the end have landed when the wave ends
#undef EOFX_LOAD_ID
  }
#undef EOFX_PAIR
#undef EOFX_VMWAIT
#undef EOFX_ALD_A
#undef EOFX_ARD_A
#undef EOFX_LOAD_A1
#undef EOFX_LOAD_A_LO
#undef EOFX_LOAD_A_HI
#undef EOFX_AXB_CONVERT_LO
#undef EOFX_AXB_CONVERT_HI
#undef EOFX_DMA_B
#undef EOFX_LOAD_F
#undef EOFX_LOAD_A
#undef EOFX_DSW64
#undef EOFX_DSR128
#undef EOFX_AXB_CONV1
#undef EOFX_AXB_CONV_TAIL
#undef EOFX_AXB_MFMA
#undef EOFX_AXB_BHI
#undef EOFX_SLAB
#undef EOFX_PAIR_BARRIER

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


}  // namespace eofx
