// NOT part of the library: the LDS-DMA variant of axb_f16_kernel written in round 4 and abandoned at the ISA stage.
// hipcc (ROCm 7.2) files an LDS-DMA load and an ordinary global load under different event types of the SAME counter, treats the
// counter as out-of-order while both are pending, and emits s_waitcnt vmcnt(0) in front of every use of a loaded register: the
// two-slab prefetch of A is drained at every wait (checked in the assembly: -S --cuda-device-only, 8 x vmcnt(0) in the pair
// loop).  Issuing the DMA from inline assembly hides it from that bookkeeping, but then every compiler-counted wait is four
// loads too strict and forces loads issued one slab ago to land.  What remains is the A stream itself in hand-counted
// assembly, with loaded values that must not be copied between the load and its wait across the loop edge -- not attempted
// for an upper bound of 4 % (profiles/r04_axb_bcost.txt).  Include after eofx_kernels.hpp, inside namespace eofx.
// ---------------------------------------------------------------------------------
// axb_f16 with the B slab taken out of the waves' instruction streams (round 4).  profiles/r04_axb_bcost.txt priced the
// hand-over of the B slab in axb_f16_kernel (loads, conversion, LDS stores, their waits) at 5.8 % of the kernel; here
//   * axb_bsplit_kernel converts the panel ONCE into two fp16 planes in the exact order the LDS buffer holds them
//     ([column block][feature pair][plane][k-group of 8][column slot, XOR-swizzled][8 halves]: 16 KiB per 64 features), and
//   * axb_f16_dma_kernel moves a pair's 16 KiB with 16 LDS-DMA instructions per workgroup (global_load_lds_dwordx4: no
//     VGPRs, no VALU, no LDS store instructions), one pair ahead.
// Everything else is axb_f16_kernel: same A stream, same map, same split, same MFMA order -> the SAME BITS in C.
// The compiler makes every LDS access it can see wait for an LDS-DMA in flight (it cannot prove them disjoint), which
// would drain the A prefetch; so this kernel has ONE LDS object and touches it only from inline assembly (as
// gram_nt_kernel does): stores of the converted A, fragment reads, and their lgkmcnt waits are written by hand.  The A loads
// stay ordinary loads, and the DMA is the builtin, so the compiler's vmcnt bookkeeping stays exact.
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
    const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
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
__global__ __launch_bounds__(256, 2) void axb_f16_dma_kernel(const float* __restrict__ A, int64_t lda, int a_rows,
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

  f32x4 a0[8], a1[8], fr[3];
  float m1 = -1.f;
  asm volatile("" : "+v"(m1));
#define EOFX_AXB_LD(p_) ((DBG & 8) ? *reinterpret_cast<const f32x4*>(p_) : __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p_)))
#define EOFX_DMA_B(pair, buf)    /* 4 of the pair's 16 rows of 1 KiB: lane l lands at byte 16 l of the row */ \
  do {                                                                                                 \
    if (!(DBG & 64)) {                                                                                 \
      const char* src_ = Pb + (int64_t)(pair) * AXB_PAIR_BYTES;                                        \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                 \
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned*)(src_ + i_ * 1024), \
                                           (__attribute__((address_space(3))) unsigned*)(bw + (buf) * 16384 + i_ * 1024), 16, 0, 0); \
    }                                                                                                  \
  } while (0)
#define EOFX_LOAD_F(freg, chunk)                                                                       \
  do {                                                                                                 \
    if (!(DBG & 4)) {                                                                                  \
      const unsigned fb_ = (unsigned)(fo + (chunk) * AXB_KC) * 4u;                                     \
      freg[0] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(aff) + fb_);             \
      freg[1] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(aff1) + fb_);            \
      freg[2] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(aff2) + fb_);            \
    }                                                                                                  \
  } while (0)
#define EOFX_LOAD_A(areg, chunk, u0)                                                                   \
  do {                                                                                                 \
    const int ko_ = (chunk) * AXB_KC;                                                                  \
    const bool kin_ = fo + ko_ < a_cols;                                                               \
    const unsigned kof_ = kin_ ? (unsigned)(fo + ko_) * 4u : 0u;                                       \
    if (full) {                                                                                        \
      unsigned base_ = lrl + kof_;                                                                     \
      asm volatile("" : "+v"(base_));                                                                  \
      _Pragma("unroll") for (int u = (u0); u < (u0) + 4; ++u)                                          \
          areg[u] = EOFX_AXB_LD(Ab + (base_ + (unsigned)(8 * u) * ldab));                              \
    } else {                                                                                           \
      int lr_ = lr;                                                                                    \
      asm volatile("" : "+v"(lr_));                                                                    \
      _Pragma("unroll") for (int u = (u0); u < (u0) + 4; ++u) {                                        \
        const int r_ = arow0 + lr_ + 8 * u < a_rows ? lr_ + 8 * u : a_rows - 1 - arow0;                \
        areg[u] = EOFX_AXB_LD(Ab + ((unsigned)r_ * ldab + kof_));                                      \
      }                                                                                                \
    }                                                                                                  \
  } while (0)
#define EOFX_DSW64(addr, val, off) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr), "v"(val), "n"(off) : "memory")
#define EOFX_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(off) : "memory")
#define EOFX_AXB_CONV1(areg, u)     /* u: a literal (the LDS offsets below are instruction immediates) */ \
  do {                                                                                                 \
    u32x2 hi_, lo_;                                                                                    \
    if (DBG & 2) {                                                                                     \
      hi_[0] = __float_as_uint(areg[u][0]); hi_[1] = __float_as_uint(areg[u][1]);                      \
      lo_[0] = __float_as_uint(areg[u][2]); lo_[1] = __float_as_uint(areg[u][3]);                      \
    } else {                                                                                           \
      _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                  \
        const f32x2 x_ = {MASK ? __uint_as_float(__float_as_uint(areg[u][2 * h]) & mk_[2 * h]) : areg[u][2 * h],           \
                          MASK ? __uint_as_float(__float_as_uint(areg[u][2 * h + 1]) & mk_[2 * h + 1]) : areg[u][2 * h + 1]}; \
        f32x2 t_;                                                                                      \
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t_) : "v"(x_), "v"(fh_[h]));     \
        f32x2 v_;                                                                                      \
        asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(v_) : "v"(t_), "v"(fs_[h]), "v"(fl_[h])); \
        const fp16x2_t p_ = __builtin_amdgcn_cvt_pkrtz(v_[0], v_[1]);                                  \
        const fp16x2_t q_ = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)p_[0], m1, v_[0]),        \
                                                       __builtin_fmaf((float)p_[1], m1, v_[1]));        \
        hi_[h] = __builtin_bit_cast(unsigned, p_);                                                     \
        lo_[h] = __builtin_bit_cast(unsigned, q_);                                                     \
      }                                                                                                \
    }                                                                                                  \
    EOFX_DSW64(as_w, hi_, 512 * (u));                                                                  \
    EOFX_DSW64(as_w, lo_, 4096 + 512 * (u));                                                           \
  } while (0)
#define EOFX_AXB_CONVERT(areg, u0)                                                                     \
  EOFX_AXB_CONV1(areg, (u0) + 0); EOFX_AXB_CONV1(areg, (u0) + 1); EOFX_AXB_CONV1(areg, (u0) + 2); EOFX_AXB_CONV1(areg, (u0) + 3);
  // 32 rows x NQ column tiles.  Reads in two sets (lo A + hi B first: the first MFMA group needs only those), each wait
  // names the registers it releases so that no MFMA can be scheduled above it.  LDS operations of a wave complete in
  // order: lgkmcnt(2 + NQ) after both sets = the first set has landed (an outstanding scalar load only makes it stricter).
#define EOFX_AXB_MFMA(jh, boff, hs)                                                                    \
  do {                                                                                                 \
    f16x8 af_[2][2], bf_[2][NQ];                                                                       \
    EOFX_DSR128(af_[0][1], as_r, 4096 + 1024 * (2 * (jh)));                                            \
    EOFX_DSR128(af_[1][1], as_r, 4096 + 1024 * (2 * (jh) + 1));                                        \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) { const unsigned ad_ = bs_r[q] + (boff); EOFX_DSR128(bf_[0][q], ad_, 4096 * (hs)); } \
    EOFX_DSR128(af_[0][0], as_r, 1024 * (2 * (jh)));                                                   \
    EOFX_DSR128(af_[1][0], as_r, 1024 * (2 * (jh) + 1));                                               \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) { const unsigned ad_ = bs_r[q] + (boff); EOFX_DSR128(bf_[1][q], ad_, 8192 + 4096 * (hs)); } \
    if constexpr (NQ == 4)                                                                             \
      asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(af_[0][1]), "+v"(af_[1][1]), "+v"(bf_[0][0]), "+v"(bf_[0][1]), "+v"(bf_[0][2]), "+v"(bf_[0][3]) : : "memory"); \
    else                                                                                               \
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af_[0][1]), "+v"(af_[1][1]), "+v"(bf_[0][0]), "+v"(bf_[0][1]) : : "memory"); \
    if (DBG & 1) {                                                                                     \
      if constexpr (NQ == 4)                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af_[0][0]), "+v"(af_[1][0]), "+v"(bf_[1][0]), "+v"(bf_[1][1]), "+v"(bf_[1][2]), "+v"(bf_[1][3]) : : "memory"); \
      else                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af_[0][0]), "+v"(af_[1][0]), "+v"(bf_[1][0]), "+v"(bf_[1][1]) : : "memory"); \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) _Pragma("unroll") for (int j = 0; j < 2; ++j)     \
          _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[2 * (jh) + j][q][r] +=                     \
              (float)af_[j][0][r] + (float)af_[j][1][r + 4] + (float)bf_[0][q][r] + (float)bf_[1][q][r]; \
    } else {                                                                                           \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) _Pragma("unroll") for (int j = 0; j < 2; ++j)     \
          acc[2 * (jh) + j][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af_[j][1], bf_[0][q], acc[2 * (jh) + j][q], 0, 0, 0); \
      if constexpr (NQ == 4)                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af_[0][0]), "+v"(af_[1][0]), "+v"(bf_[1][0]), "+v"(bf_[1][1]), "+v"(bf_[1][2]), "+v"(bf_[1][3]) : : "memory"); \
      else                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af_[0][0]), "+v"(af_[1][0]), "+v"(bf_[1][0]), "+v"(bf_[1][1]) : : "memory"); \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) _Pragma("unroll") for (int j = 0; j < 2; ++j)     \
          acc[2 * (jh) + j][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af_[j][0], bf_[1][q], acc[2 * (jh) + j][q], 0, 0, 0); \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) _Pragma("unroll") for (int j = 0; j < 2; ++j)     \
          acc[2 * (jh) + j][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af_[j][0], bf_[0][q], acc[2 * (jh) + j][q], 0, 0, 0); \
    }                                                                                                  \
  } while (0)
#define EOFX_SLAB(areg, boff, hs, next_f, next_a)                                                      \
  do {                                                                                                 \
    if (live) {                                                                                        \
      f32x2 fh_[2], fl_[2], fs_[2];                                                                    \
      _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                  \
        fh_[h] = f32x2{fr[0][2 * h], fr[0][2 * h + 1]};                                                \
        fs_[h] = f32x2{fr[2][2 * h], fr[2][2 * h + 1]} * a_scale;                                      \
        fl_[h] = f32x2{fr[1][2 * h], fr[1][2 * h + 1]} * fs_[h];                                       \
      }                                                                                                \
      unsigned mk_[4];                                                                                 \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) mk_[e] = (MASK && fr[2][e] == 0.f) ? 0u : 0xffffffffu; \
      (void)mk_;                                                                                       \
      EOFX_LOAD_F(fr, next_f);                                                                         \
      EOFX_AXB_CONVERT(areg, 0)                                                                        \
      EOFX_LOAD_A(areg, next_a, 0);                                                                    \
      EOFX_AXB_MFMA(0, boff, hs);                                                                      \
      EOFX_AXB_CONVERT(areg, 4)                                                                        \
      EOFX_LOAD_A(areg, next_a, 4);                                                                    \
      EOFX_AXB_MFMA(1, boff, hs);                                                                      \
    } else {                                                                                           \
      EOFX_LOAD_F(fr, next_f);                                                                         \
      EOFX_LOAD_A(areg, next_a, 0);                                                                    \
      EOFX_LOAD_A(areg, next_a, 4);                                                                    \
    }                                                                                                  \
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
    const int q0 = EOFX_PAIR(0);
    EOFX_DMA_B(q0, 0);
    EOFX_LOAD_F(fr, 2 * q0);
    EOFX_LOAD_A(a0, 2 * q0, 0);
    EOFX_LOAD_A(a0, 2 * q0, 4);
    EOFX_LOAD_A(a1, 2 * q0 + 1, 0);
    EOFX_LOAD_A(a1, 2 * q0 + 1, 4);
    EOFX_PAIR_BARRIER((DBG & 4) ? 16 : 19);   // 3 + 16 loads behind the first DMA
    for (int pr = 0; pr < npair; ++pr) {
      const unsigned boff = (unsigned)(pr & 1) * 16384u;
      // pair ids of this and the next pair (past the end: harmless re-reads of the last pair)
      const int qc = EOFX_PAIR(pr), p1 = EOFX_PAIR(pr + 1 < npair ? pr + 1 : npair - 1);
      const int c2 = 2 * p1;
      EOFX_DMA_B(p1, 1 - (pr & 1));
      EOFX_SLAB(a0, boff, 0, 2 * qc + 1, c2);
      EOFX_SLAB(a1, boff, 1, c2, c2 + 1);
      EOFX_PAIR_BARRIER((DBG & 4) ? 16 : 22);
    }
  }
#undef EOFX_PAIR
#undef EOFX_AXB_LD
#undef EOFX_DMA_B
#undef EOFX_LOAD_F
#undef EOFX_LOAD_A
#undef EOFX_DSW64
#undef EOFX_DSR128
#undef EOFX_AXB_CONVERT
#undef EOFX_AXB_CONV1
#undef EOFX_AXB_MFMA
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

