// Fused power-iteration product  W = X (X^T Z)  in ONE pass over the sample-contiguous layout X^T
// ("a fused implementation physically reads less").  See DESIGN.md §14.
//
// Persistent kernel, one 512-thread workgroup per CU (two roles of four waves each, one wave of each role per SIMD).
// Workgroup w belongs to group g = w % 8 (the dispatcher places workgroup w on XCD w % 8 -- checked at kernel start
// with HW_REG_XCC_ID -- so a group is the 32 CUs of one XCD and shares its L2) and is member c = w / 8 of it.
// Member c owns the R = n_pad / 32 samples I_c = [c R, (c+1) R): it keeps Z[I_c] (as split-fp16 MFMA B operands,
// 16 columns per wave) and the accumulators of W_g[I_c] (16-sample tiles distributed over the waves) in registers.
// The group walks over the feature slabs s = g, g + 8, ... (32 features = one contiguous 32 x 4 n_pad byte block of X^T).
//
// Per slab (iteration `it` of the group) a member runs a software pipeline.  Role A (waves 0-3) streams the matrix and
// does phase 1, role B (waves 4-7) does the exchange and phase 2; the two instruction streams share nothing but the
// two workgroup barriers of an iteration, so the hardware overlaps one role's LDS / MFMA latencies with the other's:
//   stage      : the sub-slab X^T[32 features][I_c] (R x 128 B, coalesced 16 B/lane loads issued two iterations
//                ahead) is converted to split fp16 and written to an LDS tile in [feature][sample] order;
//   phase 1    : partial Y = sub-slab . Z[I_c]  (32 x 64) on v_mfma_f32_16x16x32_f16 (hh + hl + lh), wave w
//                computes columns 16 w .. 16 w + 15 -- no cross-wave reduction; every lane posts its 8 accumulator
//                values to the group's exchange buffer in the XCD's L2;
//   exchange   : data-tagged 16-byte granules {v, v, v, tag = slab + 1}: no flags, no fences, no drains.  Member c
//                reduces thread-slots 8 c .. 8 c + 7 of the 32 partial tiles (fixed butterfly over the members) and
//                posts them (reduce-scatter); every member then reads the complete 32 x 64 tile (all-gather).  Loads
//                of an exchange step are issued one iteration before they are consumed.  A tag mismatch (data not
//                there yet) makes that wave poll on its own, bounded by a spin limit;
//   phase 2    : LAG iterations later the same sub-slab is read again -- it is still in the XCD's L2 (4 MiB, about
//                three slabs) most of the time -- staged as f32 in [feature][sample] order and contracted over the
//                32 features with Y (split fp16): W_g[I_c] += sub-slab^T . Y on the same MFMA, 16-sample tiles.
// Same-XCD visibility: exchange stores are plain (write-through L1, kept in the L2), exchange loads carry sc1
// (served by the L2, never by the reader's L1).  At the end the 8 group partials of W are summed in a fixed order by
// splitk_reduce_kernel.  Optionally the complete Y (p_pad x 64) is written as well (the last pair of a
// randomized SVD needs both).  Scaling: X by a_scale, Z by b_scale (|scaled| <= 2^14, exact powers of two),
// the accumulated Y by s1 = 2^-(13 + ceil(log2 n_pad)) so that |Y s1| <= 2^15 whatever the data
// (|Y| <= n_pad 2^28 in scaled units); typical entries stay far above fp16's subnormal range.
#pragma once
#include <type_traits>

namespace eofx {

constexpr int FX_MEMBERS = 32;   // workgroups per group (CUs per XCD)
constexpr int FX_GROUPS = 8;     // XCDs
constexpr int FX_SLOTS = 8;      // exchange slots per group (a slab uses slot it % 8; tags tell generations apart)
constexpr int FX_LRS = 2;        // reduce-scatter of slab s: loads issued in iteration s + 1, consumed in s + 2
constexpr int FX_LAG = 4;        // all-gather + re-read of slab s: issued in s + 3, consumed (phase 2) in s + 4
constexpr int FX_TILE_GRAN = 768;   // granules per partial tile: 256 thread-slots x 3
constexpr unsigned FX_SPIN_LIMIT = 400000;

struct FxParams {
  const float* Xt;      // [p_pad x ldx], ldx = n_pad = 32 R
  int64_t ldx;
  int64_t niter;        // slabs per group = p_pad / 32 / 8
  const float* Z;       // [n_pad x 64]
  float* Wpart;         // [8][n_pad x 64]
  u32x4* E1;            // [8][SLOTS][32 members][768]   (zeroed before the launch)
  u32x4* E2;            // [8][SLOTS][768]
  float* Yout;          // [p_pad x 64] or nullptr
  const float* z_absmax;
  float a_scale, s1;
  int* ctl;             // [0] error code (0 ok, 1 time-out, 2 placement), zeroed before the launch
};

// 4 floats (already scaled) -> 4 fp16 hi (round toward zero) + 4 fp16 lo (remainder), packed 2 x 32 bit each
__device__ __forceinline__ void fx_split4(f32x4 v, unsigned (&hi)[2], unsigned (&lo)[2]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const fp16x2_t p = __builtin_amdgcn_cvt_pkrtz(v[2 * h], v[2 * h + 1]);
    hi[h] = __builtin_bit_cast(unsigned, p);
    const fp16x2_t q = __builtin_amdgcn_cvt_pkrtz(v[2 * h] - (float)p[0], v[2 * h + 1] - (float)p[1]);
    lo[h] = __builtin_bit_cast(unsigned, q);
  }
}

// DBG (probe builds only): bit 0 in-kernel phase timers, 1 no re-read loads, 2 no exchange loads, 3 no phase-2 MFMAs,
// 4 no phase-1 MFMAs, 5 stream with nt loads, 6 no staging of the re-read, 7 phase 2 only two iterations behind (with bit 2)
template <int KS, int DBG = 0>
__global__ __launch_bounds__(512) void fused2_kernel(FxParams P) {
  constexpr bool TIMING = (DBG & 1) != 0;
  constexpr int LAG = (DBG & 128) ? 2 : FX_LAG, LRS = (DBG & 128) ? 1 : FX_LRS;   // bit 7: timing probe of a short lag
  constexpr int R = 32 * KS;              // samples per member
  constexpr int NT = 2 * KS;              // 16-sample tiles per member
  constexpr int MT = (NT + 3) / 4;        // tiles per wave (role B)
  constexpr int SA = R + 16;              // halves per row of the fp16 planes (32 B pad: conflict-free b128 fragment reads)
  constexpr int SB = R + 4;               // floats per row of the f32 tile (4 rows apart = 16 banks apart)
  extern __shared__ __attribute__((aligned(16))) unsigned char fx_smem[];
  _Float16* Ahi = reinterpret_cast<_Float16*>(fx_smem);          // [32][SA]   role A
  _Float16* Alo = Ahi + 32 * SA;                                 // [32][SA]
  float* Bt = reinterpret_cast<float*>(Alo + 32 * SA);           // [32][SB]   role B
  float* wsum = Bt + 32 * SB;                                    // [4 waves][8 slots][8 values]
  float* sums = wsum + 256;                                      // [8 slots][8 values]
  u32x4* Ys = reinterpret_cast<u32x4*>(sums + 64);               // [4 column tiles][64 lanes][hi, lo]

  const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  const int tid = threadIdx.x & 255, lane = tid & 63;            // thread / lane within the role
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave within the role
  const int m16 = lane & 15, kg = lane >> 4;
  const int g = blockIdx.x % FX_GROUPS, c = blockIdx.x / FX_GROUPS;
  const int64_t ldx = P.ldx, niter = P.niter;
  const float a_scale = P.a_scale, s1 = P.s1;
  const float b_scale = f16_scale_for(*P.z_absmax);
  const int64_t total = (niter + LAG + 1) & ~(int64_t)1;      // iterations (even), the same for both roles

  if (threadIdx.x == 0) {   // every member of group g has to run on XCD g: the exchange relies on a shared L2
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((int)(xcc & 0xf) != g) __hip_atomic_store(P.ctl, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- per-thread chunk map of a sub-slab: thread = (feature f = tid / 8, j = tid % 8) takes the 16-byte chunks
  //      j + 8 u of its feature row: whole 128-byte lines per 8 lanes, and every offset is base + constant * u ----
  const int goff0 = (tid >> 3) * (int)ldx + 4 * (tid & 7);                      // floats; chunk u: + 32 u
  const int loff0 = (tid >> 3) * (role == 0 ? SA : SB) + 4 * (tid & 7);         // halves / floats; chunk u: + 32 u
  const float* xbase = P.Xt + (int64_t)c * R + (int64_t)g * 32 * ldx;   // slab `it` of this group: + it * 256 ldx
  const int64_t slab_stride = (int64_t)FX_GROUPS * 32 * ldx;
  auto slab_ptr = [&](int64_t it) {
    it = it < 0 ? 0 : (it < niter ? it : niter - 1);
    return xbase + it * slab_stride;
  };
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = TIMING ? wall_clock64() : 0;
  auto tick = [&](int k) {
    if (TIMING) {
      const long long now = wall_clock64();
      tacc[k] += now - tprev;
      tprev = now;
    }
  };

  if (role == 0) {
    // =====================================  role A: stream + phase 1  =====================================
    // Z[I_c], columns 16 wave .. +15, as B operands: k = 8 kg + t  <->  sample 32 ks + 8 kg + t
    f16x8 zb[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f32x8 v;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        v[t] = P.Z[((int64_t)c * R + 32 * ks + 8 * kg + t) * 64 + 16 * wave + m16] * b_scale;
      split_f16(v, zb[ks]);
    }
    u32x4* e1w = P.E1 + ((size_t)g * FX_SLOTS * FX_MEMBERS + c) * FX_TILE_GRAN + tid * 3;   // + slot * 32 * 768
    f32x4 xa[2][KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) xa[0][u] = *reinterpret_cast<const f32x4*>(slab_ptr(0) + goff0 + 32 * u);
#pragma unroll
    for (int u = 0; u < KS; ++u) xa[1][u] = *reinterpret_cast<const f32x4*>(slab_ptr(1) + goff0 + 32 * u);

    auto body = [&](auto par_c, int64_t it) {
      constexpr int par = decltype(par_c)::value;
      // (1) stage the sub-slab of this iteration as split fp16
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        unsigned hi[2], lo[2];
        fx_split4(xa[par][u] * a_scale, hi, lo);
        *reinterpret_cast<uint2*>(Ahi + loff0 + 32 * u) = make_uint2(hi[0], hi[1]);
        *reinterpret_cast<uint2*>(Alo + loff0 + 32 * u) = make_uint2(lo[0], lo[1]);
      }
      tick(0);
      tick(1);
      __syncthreads();
      tick(2);
      // (3) phase 1 of slab `it`, post the partial tile
      {
        const float* s1p = slab_ptr(it + 2);
        f32x4 a1[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int h = 0; h < 2; ++h) a1[mt][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < ((DBG & 16) ? 0 : KS); ++ks) {
          f16x8 ah[2], al[2];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const _Float16* ph = Ahi + (16 * mt + m16) * SA + 32 * ks + 8 * kg;
            ah[mt] = *reinterpret_cast<const f16x8*>(ph);
            al[mt] = *reinterpret_cast<const f16x8*>(ph + 32 * SA);
          }
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            a1[mt][ks & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], zb[ks][0], a1[mt][ks & 1], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            a1[mt][ks & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], zb[ks][1], a1[mt][ks & 1], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            a1[mt][ks & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], zb[ks][0], a1[mt][ks & 1], 0, 0, 0);
          // the stream, two iterations ahead (plain loads: the lines stay in the L2 for phase 2): one 16-byte chunk per
          // k-step, so the address unit is fed evenly instead of in one burst per iteration
          xa[par][ks] = (DBG & 32) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(s1p + goff0 + 32 * ks))
                                   : *reinterpret_cast<const f32x4*>(s1p + goff0 + 32 * ks);
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // 4 LDS reads
          __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);   // 6 MFMAs
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 global load
        }
        const f32x4 v0 = a1[0][0] + a1[0][1], v1 = a1[1][0] + a1[1][1];
        const unsigned tag = (unsigned)(it + 1);
        u32x4* dst = e1w + (size_t)(it & (FX_SLOTS - 1)) * (FX_MEMBERS * FX_TILE_GRAN);
        dst[0] = u32x4{__float_as_uint(v0[0]), __float_as_uint(v0[1]), __float_as_uint(v0[2]), tag};
        dst[1] = u32x4{__float_as_uint(v0[3]), __float_as_uint(v1[0]), __float_as_uint(v1[1]), tag};
        dst[2] = u32x4{__float_as_uint(v1[2]), __float_as_uint(v1[3]), tag, tag};
      }
      tick(3);
      __syncthreads();   // the fp16 planes are rewritten by the next iteration
      tick(4);
    };
    for (int64_t it = 0; it < total; it += 2) {
      body(std::integral_constant<int, 0>{}, it);
      body(std::integral_constant<int, 1>{}, it + 1);
    }
    if (TIMING && blockIdx.x == 9 && tid == 0) {
      long long* ts = reinterpret_cast<long long*>(P.ctl + 2);
      for (int k = 0; k < 5; ++k) ts[k] = tacc[k];
      ts[16] = total;
    }
  } else {
    // =====================================  role B: exchange + phase 2  =====================================
    const float y_out_scale = 1.f / (a_scale * b_scale);
    f32x4 accW[MT][4];
#pragma unroll
    for (int tl = 0; tl < MT; ++tl)
#pragma unroll
      for (int q = 0; q < 4; ++q) accW[tl][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const auto e1r = __builtin_amdgcn_make_buffer_rsrc(P.E1 + (size_t)g * FX_SLOTS * FX_MEMBERS * FX_TILE_GRAN, 0,
                                                       FX_SLOTS * FX_MEMBERS * FX_TILE_GRAN * 16, 0x00020000);
    const auto e2r = __builtin_amdgcn_make_buffer_rsrc(P.E2 + (size_t)g * FX_SLOTS * FX_TILE_GRAN, 0,
                                                       FX_SLOTS * FX_TILE_GRAN * 16, 0x00020000);
    u32x4* e2w = P.E2 + (size_t)g * FX_SLOTS * FX_TILE_GRAN;
    // reduce-scatter read: member m = tid / 8, thread-slot 8 c + (tid % 8)
    const int rs_voff = (((tid >> 3) * FX_TILE_GRAN) + (8 * c + (tid & 7)) * 3) * 16;        // + slot * 32 * 768 * 16
    const int ag_voff = tid * 3 * 16;                                                          // + slot * 768 * 16
    f32x4 xb[KS];
    u32x4 rsv[3], agv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) rsv[i] = agv[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < KS; ++u) xb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_rs = [&](int64_t s) {
      const int slot = (int)(s & (FX_SLOTS - 1));
#pragma unroll
      for (int i = 0; i < 3; ++i)
        rsv[i] = __builtin_amdgcn_raw_buffer_load_b128(e1r, rs_voff + slot * (FX_MEMBERS * FX_TILE_GRAN * 16) + 16 * i, 0, 16);
    };
    auto load_ag = [&](int64_t s) {
      const int slot = (int)(s & (FX_SLOTS - 1));
#pragma unroll
      for (int i = 0; i < 3; ++i)
        agv[i] = __builtin_amdgcn_raw_buffer_load_b128(e2r, ag_voff + slot * (FX_TILE_GRAN * 16) + 16 * i, 0, 16);
    };
    auto tags_ok = [&](int64_t s_rs, int64_t s_ag) -> bool {
      const bool rs_valid = s_rs >= 0 && s_rs < niter, ag_valid = s_ag >= 0 && s_ag < niter;
      const unsigned trs = (unsigned)(s_rs + 1), tag_ = (unsigned)(s_ag + 1);
      bool ok = true;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        ok = ok && (!rs_valid || rsv[i][3] == trs);
        ok = ok && (!ag_valid || agv[i][3] == tag_);
      }
      ok = ok && (!rs_valid || rsv[2][2] == trs) && (!ag_valid || agv[2][2] == tag_);
      return ok;
    };

    auto body = [&](int64_t it) {
      const int64_t s_rs = it - LRS, s_ag = it - LAG;
      // (1) consume what the previous iteration issued.  The re-read sub-slab goes to LDS first: its registers are
      //     reloaded below, after the exchange work has given the LDS writes time to drain.
#pragma unroll
      for (int u = 0; u < KS; ++u)
        if (!(DBG & 64)) *reinterpret_cast<f32x4*>(Bt + loff0 + 32 * u) = xb[u];
      // the exchange:  Tags first; a wave whose data has not landed polls on its own.
      if (!(DBG & 4) && __any(!tags_ok(s_rs, s_ag))) {
        unsigned spins = 0;
        for (;;) {
          if (__hip_atomic_load(P.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;   // failed elsewhere: wind down
          load_rs(s_rs);
          load_ag(s_ag);
          if (!__any(!tags_ok(s_rs, s_ag))) break;
          if (++spins > FX_SPIN_LIMIT) {   // a member never posted
            __hip_atomic_store(P.ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(4);
        }
      }
      {   // reduce-scatter values: sum over the 8 members of this wave (lane bits 3..5), fixed butterfly
        float v[8] = {__uint_as_float(rsv[0][0]), __uint_as_float(rsv[0][1]), __uint_as_float(rsv[0][2]),
                      __uint_as_float(rsv[1][0]), __uint_as_float(rsv[1][1]), __uint_as_float(rsv[1][2]),
                      __uint_as_float(rsv[2][0]), __uint_as_float(rsv[2][1])};
#pragma unroll
        for (int o = 8; o < 64; o <<= 1)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += __shfl_xor(v[e], o);
        if (lane < 8) {
          f32x4* dst = reinterpret_cast<f32x4*>(wsum + (wave * 8 + lane) * 8);
          dst[0] = f32x4{v[0], v[1], v[2], v[3]};
          dst[1] = f32x4{v[4], v[5], v[6], v[7]};
        }
      }
      {   // all-gathered Y at this thread's own tile positions -> split fp16 B operand of column tile `wave`
        const bool ag_valid = s_ag >= 0 && s_ag < niter;
        f32x8 y;
        y[0] = __uint_as_float(agv[0][0]); y[1] = __uint_as_float(agv[0][1]);
        y[2] = __uint_as_float(agv[0][2]); y[3] = __uint_as_float(agv[1][0]);
        y[4] = __uint_as_float(agv[1][1]); y[5] = __uint_as_float(agv[1][2]);
        y[6] = __uint_as_float(agv[2][0]); y[7] = __uint_as_float(agv[2][1]);
#pragma unroll
        for (int t = 0; t < 8; ++t) y[t] = ag_valid ? y[t] * s1 : 0.f;
        f16x8 yh[2];
        split_f16(y, yh);
        Ys[tid * 2 + 0] = __builtin_bit_cast(u32x4, yh[0]);
        Ys[tid * 2 + 1] = __builtin_bit_cast(u32x4, yh[1]);
      }
      tick(0);
      // (2) issue the loads of the next iteration: exchange first, the re-read of the sub-slab last
      if (!(DBG & 4)) {
        load_rs(it + 1 - LRS);
        load_ag(it + 1 - LAG);
      }
      __builtin_amdgcn_sched_barrier(0);
      tick(1);
      __syncthreads();
      tick(2);
      // (3a) finish the reduce-scatter of slab it - LRS (first wave of the role): 4 wave partials in wave order
      if (wave == 0 && s_rs >= 0 && s_rs < niter) {
        const float acc = ((wsum[lane] + wsum[64 + lane]) + wsum[128 + lane]) + wsum[192 + lane];   // lane = 8 slot + value
        sums[lane] = acc;
        if (P.Yout) {
          const int ts = 8 * c + (lane >> 3), e = lane & 7;          // thread-slot, value index
          const int col = 16 * (ts >> 6) + (ts & 15), row = 16 * (e >> 2) + 4 * ((ts & 63) >> 4) + (e & 3);
          P.Yout[((int64_t)(g + FX_GROUPS * s_rs) * 32 + row) * 64 + col] = acc * y_out_scale;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 24) {
          const int j = lane / 3, i = lane - 3 * j;
          const float* sv = sums + 8 * j + 3 * i;
          const unsigned tg = (unsigned)(s_rs + 1);
          const unsigned w0 = __float_as_uint(sv[0]), w1 = __float_as_uint(sv[1]);
          const unsigned w2 = i < 2 ? __float_as_uint(sv[2]) : tg;
          e2w[(size_t)(s_rs & (FX_SLOTS - 1)) * FX_TILE_GRAN + (8 * c + j) * 3 + i] = u32x4{w0, w1, w2, tg};
        }
      }
      tick(3);
      // (3b) phase 2 of slab it - LAG
      {
        const float* s2 = slab_ptr(it + 1 - LAG);
        f16x8 yb[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          yb[q][0] = __builtin_bit_cast(f16x8, Ys[(q * 64 + lane) * 2 + 0]);
          yb[q][1] = __builtin_bit_cast(f16x8, Ys[(q * 64 + lane) * 2 + 1]);
        }
#pragma unroll
        for (int tl = 0; tl < MT; ++tl) {
          const int tile = wave + 4 * tl;
          if (!(DBG & 8) && (NT % 4 == 0 || tile < NT)) {
            const float* px = Bt + (4 * kg) * SB + 16 * tile + m16;
            f32x8 x;
#pragma unroll
            for (int t = 0; t < 8; ++t) x[t] = px[((t & 3) + 16 * (t >> 2)) * SB] * a_scale;
            f16x8 af[2];
            split_f16(x, af);
#pragma unroll
            for (int q = 0; q < 4; ++q) accW[tl][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1], yb[q][0], accW[tl][q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) accW[tl][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], yb[q][1], accW[tl][q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) accW[tl][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], yb[q][0], accW[tl][q], 0, 0, 0);
          }
          // the re-read of the next phase-2 sub-slab, spread over the tiles
#pragma unroll
          for (int u = tl * KS / MT; u < (tl + 1) * KS / MT; ++u)
            if (!(DBG & 2)) xb[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(s2 + goff0 + 32 * u));
        }
      }
      tick(4);
      __syncthreads();   // the f32 tile, the wave partials and Y are rewritten by the next iteration
      tick(5);
    };
    for (int64_t it = 0; it < total; ++it) body(it);
    if (TIMING && blockIdx.x == 9 && tid == 0) {
      long long* ts = reinterpret_cast<long long*>(P.ctl + 2);
      for (int k = 0; k < 6; ++k) ts[8 + k] = tacc[k];
    }
    // ---- this group's partial of W[I_c]:  acc / (a^2 b s1) ----
    const float inv_a = 1.f / a_scale;
    const float w_scale2 = 1.f / (b_scale * s1);
    float* Wg = P.Wpart + (size_t)g * (size_t)(FX_MEMBERS * R) * 64;
#pragma unroll
    for (int tl = 0; tl < MT; ++tl) {
      const int tile = wave + 4 * tl;
      if (NT % 4 == 0 || tile < NT) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Wg[((int64_t)c * R + 16 * tile + 4 * kg + r) * 64 + 16 * q + m16] = ((accW[tl][q][r] * inv_a) * inv_a) * w_scale2;
      }
    }
  }
}

}  // namespace eofx
