// Fused power-iteration product  W = X (X^T Z)  in ONE pass over the sample-contiguous layout X^T
// ("a fused implementation physically reads less").  EXPERIMENTAL -- see DESIGN.md §14.
//
// Persistent cooperative kernel, one workgroup per CU.  Workgroup w belongs to group g = w % 8 (the
// dispatcher assigns workgroups to XCDs round robin, so a group is the 32 CUs of one XCD and shares its L2)
// and is member c = w / 8 of it.  Member c owns the sample range I_c = [c R, (c+1) R), R = n_pad / 32:
// it keeps Z[I_c] (as split-fp16 MFMA operands) and the accumulators of W_g[I_c] in registers.  The group
// walks over the feature slabs t = g, g + 8, ... (32 features each).  Per slab:
//   1. load the sub-slab X^T[32 features][I_c] into LDS (R * 128 B, fully coalesced 4 R-byte row pieces);
//   2. phase 1: partial Y = sub-slab . Z[I_c]  (32 x 64, scaled split-fp16 MFMA), summed over the 4 waves;
//   3. exchange through the XCD's L2: every member posts its partial, member c reduces piece c of the tile
//      over the 32 partials in a fixed order, posts it, and everybody reads the complete Y (32 x 64);
//      flags are relaxed agent-scope atomics (0.9 us per barrier, tools/probes/sync_probe.hip), data move as
//      relaxed atomic stores / loads, "s_waitcnt vmcnt(0)" orders a member's stores before its flag;
//   4. phase 2: W_g[I_c] += sub-slab^T . Y  (exact f32 MFMA 32x32x2 from the same LDS tile).
// At the end the 8 group partials of W are summed in a fixed order by splitk_reduce_kernel.
#pragma once

namespace eofx {

constexpr int FX_MEMBERS = 32;   // workgroups per group (CUs per XCD)
constexpr int FX_GROUPS = 8;     // XCDs
constexpr int FX_SLOTS = 4;      // exchange slots per group (generation counted)
constexpr int FX_MAXT = 3;       // 32-sample tiles per wave   (rows per CU <= 4 * 3 * 32 = 384)
constexpr int FX_MAXK = 6;       // 16-sample k-steps per wave (rows per CU <= 4 * 6 * 16 = 384)
constexpr long FX_SPIN_LIMIT = 4000000;   // ~2 s; after the first time-out every wait returns at once

// Exchange traffic stays inside the XCD: stores are written through to the L2 and loads bypass the CU's
// vector L1 (scope bit sc0 = "group"), nothing goes out to the fabric (agent-scope sc1 accesses cost ~10 us
// per 8 KB tile).  Valid because all members of a group sit on the same XCD (checked at kernel start).
__device__ __forceinline__ void fx_store4(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void fx_load4x2(const float* p0, const float* p1, f32x4& a, f32x4& b) {
  asm volatile("global_load_dwordx4 %0, %2, off sc0\n\tglobal_load_dwordx4 %1, %3, off sc0\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(b) : "v"(p0), "v"(p1) : "memory");
}
// thread 0 signals (after every thread's stores have reached the L2) and waits for all members
__device__ __forceinline__ void fx_signal_and_wait(int* flag, int target, int* err) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      ++spins;
      if (spins > FX_SPIN_LIMIT) {      // a member never arrived: flag the failure, everybody winds down fast
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      if ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    }
  }
  __syncthreads();
}

// Xt: [p_pad x ldx] (ldx = n_pad); Z: [n_pad x 64]; Wpart: [8][n_pad x 64]
// scratchP: [8][FX_SLOTS][32][2048], scratchR: [8][FX_SLOTS][2048], flags: [8][FX_SLOTS][2] (zeroed)
__global__ __launch_bounds__(256, 1) void fused_xxt_kernel(const float* __restrict__ Xt, int64_t ldx,
                                                            int64_t p_pad, int rows_per_cu,
                                                            const float* __restrict__ Z,
                                                            float* __restrict__ Wpart, float* scratchP,
                                                            float* scratchR, int* flags, float a_scale,
                                                            const float* __restrict__ z_absmax, int* err) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int R = rows_per_cu;
  const int tstride = R + 4;                    // floats per feature row of the tile (padded)
  float* T = smem;                              // [32][tstride]
  float* red = smem + 32 * tstride;             // [4][32][64] partial Y of the four waves
  float* Ys = red + 4 * 2048;                   // [32][64] complete Y of the slab

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int g = blockIdx.x % FX_GROUPS, c = blockIdx.x / FX_GROUPS;
  const int64_t i0 = (int64_t)c * R;            // first sample of this member
  const int ntiles = R / 32, nksteps = R / 16;
  const float b_scale = f16_scale_for(*z_absmax);
  const float out_scale = 1.f / (a_scale * b_scale);

  // ---- Z[I_c] as B operands of the phase-1 MFMAs: k-steps js = wave + 4 jl ----
  f16x8 zb[FX_MAXK][2][2];
#pragma unroll
  for (int jl = 0; jl < FX_MAXK; ++jl) {
    const int js = wave + 4 * jl;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x8 v;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        v[t] = (js < nksteps) ? Z[(i0 + 16 * js + 8 * lh + t) * 64 + 32 * q + li] * b_scale : 0.f;
      split_f16(v, zb[jl][q]);
    }
  }
  f32x16 accW[FX_MAXT][2];
#pragma unroll
  for (int tl = 0; tl < FX_MAXT; ++tl)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) accW[tl][q][r] = 0.f;

  float* myP = scratchP + ((size_t)g * FX_SLOTS) * FX_MEMBERS * 2048;
  float* myR = scratchR + ((size_t)g * FX_SLOTS) * 2048;
  int* myF = flags + (size_t)g * FX_SLOTS * 2;

  if (tid == 0) {   // every member of group g has to run on XCD g (round-robin dispatch): the exchange relies on a shared L2
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((int)(xcc & 0xf) != g) __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int64_t nslabs = p_pad / 32;
  long long tacc[6] = {0, 0, 0, 0, 0, 0};
  long long tprev = wall_clock64();
#define FX_TICK(k)                                      \
  do {                                                  \
    const long long now_ = wall_clock64();              \
    tacc[k] += now_ - tprev;                            \
    tprev = now_;                                       \
  } while (0)
  int it = 0;
  for (int64_t slab = g; slab < nslabs; slab += FX_GROUPS, ++it) {
    if ((it & 15) == 0) {          // wind down after a time-out anywhere (decision made uniform by the barrier)
      const int e = (tid == 0) ? (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1) : 0;
      if (__syncthreads_or(e)) break;
    }
    const int slot = it % FX_SLOTS, gen = it / FX_SLOTS;
    // 1. sub-slab -> LDS
    {
      const float* src = Xt + slab * 32 * ldx + i0;
      const int r4 = R / 4;
      for (int idx = tid; idx < 32 * r4; idx += 256) {
        const int f = idx / r4, c4 = idx - f * r4;
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (int64_t)f * ldx + 4 * c4));
        *reinterpret_cast<f32x4*>(&T[f * tstride + 4 * c4]) = v;
      }
    }
    __syncthreads();
    FX_TICK(0);
    // 2. phase 1: partial Y[32 features x 64] over this wave's k-steps
    {
      f32x16 acc[2];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
#pragma unroll
      for (int jl = 0; jl < FX_MAXK; ++jl) {
        const int js = wave + 4 * jl;
        if (js < nksteps) {
          const float* tp = &T[li * tstride + 16 * js + 8 * lh];
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(tp), x1 = *reinterpret_cast<const f32x4*>(tp + 4);
          f32x8 x;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            x[t] = x0[t] * a_scale;
            x[4 + t] = x1[t] * a_scale;
          }
          f16x8 af[2];
          split_f16(x, af);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], zb[jl][q][0], acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], zb[jl][q][1], acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], zb[jl][q][0], acc[q], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          red[wave * 2048 + row * 64 + 32 * q + li] = acc[q][r];
        }
    }
    __syncthreads();
    FX_TICK(1);
    // 3a. post this member's partial (sum of the four waves, fixed order): 512 float4, two per thread
    {
      float* dst = myP + ((size_t)slot * FX_MEMBERS + c) * 2048;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e4 = tid + 256 * u;
        f32x4 v;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int e = 4 * e4 + x;
          v[x] = (((red[e] + red[2048 + e]) + red[4096 + e]) + red[6144 + e]) * out_scale;
        }
        fx_store4(dst + 4 * e4, v);
      }
    }
    fx_signal_and_wait(myF + slot * 2, FX_MEMBERS * (gen + 1), err);
    FX_TICK(2);
    // 3b. reduce piece c (64 elements = 16 float4) over the 32 members: thread (v4, m) fetches members m and
    //     m + 16, the 32 values of every element are then summed in member order (fixed)
    {
      const int v4 = tid & 15, m = tid >> 4;
      const float* src = myP + (size_t)slot * FX_MEMBERS * 2048 + 64 * c + 4 * v4;
      f32x4 a, b;
      fx_load4x2(src + (size_t)m * 2048, src + (size_t)(m + 16) * 2048, a, b);
      *reinterpret_cast<f32x4*>(&red[m * 64 + 4 * v4]) = a;
      *reinterpret_cast<f32x4*>(&red[(m + 16) * 64 + 4 * v4]) = b;
      __syncthreads();
      if (tid < 16) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int mm = 0; mm < FX_MEMBERS; ++mm) sum += *reinterpret_cast<const f32x4*>(&red[mm * 64 + 4 * tid]);
        fx_store4(myR + (size_t)slot * 2048 + 64 * c + 4 * tid, sum);
      }
    }
    fx_signal_and_wait(myF + slot * 2 + 1, FX_MEMBERS * (gen + 1), err);
    FX_TICK(3);
    // 3c. the complete Y of this slab
    {
      const float* src = myR + (size_t)slot * 2048;
      f32x4 a, b;
      fx_load4x2(src + 4 * tid, src + 4 * (tid + 256), a, b);
      *reinterpret_cast<f32x4*>(&Ys[4 * tid]) = a;
      *reinterpret_cast<f32x4*>(&Ys[4 * (tid + 256)]) = b;
    }
    __syncthreads();
    FX_TICK(4);
    // 4. phase 2: W[I_c] += sub-slab^T . Y   (exact f32 MFMA; tiles tile = wave + 4 tl)
    {
      float yb[16][2];
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) yb[s][q] = Ys[(2 * s + lh) * 64 + 32 * q + li];
#pragma unroll
      for (int tl = 0; tl < FX_MAXT; ++tl) {
        const int tile = wave + 4 * tl;
        if (tile < ntiles) {
#pragma unroll
          for (int s = 0; s < 16; ++s) {
            const float a = T[(2 * s + lh) * tstride + 32 * tile + li];
#pragma unroll
            for (int q = 0; q < 2; ++q)
              accW[tl][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, yb[s][q], accW[tl][q], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();   // T, red, Ys are rewritten by the next slab
    FX_TICK(5);
  }
  if (blockIdx.x == 9 && tid == 0) {
    long long* ts = reinterpret_cast<long long*>(err + 2);
    for (int k = 0; k < 6; ++k) ts[k] = tacc[k];
    ts[6] = it;
  }
#undef FX_TICK
  // ---- this group's partial of W[I_c] ----
  float* Wg = Wpart + (size_t)g * (size_t)(FX_MEMBERS * R) * 64;
#pragma unroll
  for (int tl = 0; tl < FX_MAXT; ++tl) {
    const int tile = wave + 4 * tl;
    if (tile < ntiles) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          Wg[(i0 + 32 * tile + row) * 64 + 32 * q + li] = accW[tl][q][r];
        }
    }
  }
}

}  // namespace eofx
