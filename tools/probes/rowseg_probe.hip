// VERDICT r04 item 4, the untested hypothesis about the in-place sample-side product (axb_f16): "each wave instruction touches
// 8 rows x 128 B; atb touches 2 rows x 512 B -- the row-segment length, not the B slab, is the difference".  Load-only floors
// of the SAME tile walk (a workgroup = 256 rows x one split-K range of the row-major 10000 x 1036800 field, a wave = 64 rows,
// 16-byte loads, 16 loads in flight per lane) with 128 / 256 / 512 / 1024 bytes of a row per wave instruction, next to a flat
// read of the same bytes.  Nothing but loads and one v_max per register: the floor of each pattern.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/rowseg_probe.hip -o build/rowseg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// SEG: bytes of one row per wave instruction; lanes per row = SEG / 16, rows per instruction = 1024 / SEG
template <int SEG, bool NT>
__global__ __launch_bounds__(256, 2) void rowseg_kernel(const float* __restrict__ A, int64_t ld, int n, int64_t kps, int rt, float* __restrict__ out) {
  constexpr int LPR = SEG / 16, RPI = 64 / LPR, NI = 64 / RPI;        // instructions per slab of SEG bytes x 64 rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x % rt, split = blockIdx.x / rt;
  const int64_t row0 = (int64_t)tile * 256 + wave * 64;
  const int64_t k0 = (int64_t)split * kps, k1 = (k0 + kps < ld) ? k0 + kps : ld;
  const int lr = lane / LPR, lc = (lane % LPR) * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int NB = NI < 16 ? NI : 16;                                // loads in flight per lane
  for (int64_t k = k0; k < k1; k += SEG / 4) {
#pragma unroll
    for (int i0 = 0; i0 < NI; i0 += NB) {
      f32x4 v[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int64_t r = row0 + (int64_t)(i0 + i) * RPI + lr;
        const float* p = A + (r < n ? r : n - 1) * ld + k + lc;
        v[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)) : *reinterpret_cast<const f32x4*>(p);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        acc.x = fmaxf(acc.x, v[i].x); acc.y = fmaxf(acc.y, v[i].y); acc.z = fmaxf(acc.z, v[i].z); acc.w = fmaxf(acc.w, v[i].w);
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;      // (never: keeps the loads)
}

// the MFMA-operand pattern of v_mfma_f32_32x32x8_f16 read straight from the field: lane l holds 4 consecutive features (one
// float4) of row l % 32, at feature offset 4 (l / 32): one wave instruction = 32 rows x 32 bytes; four consecutive instructions
// (k += 8) finish the 128-byte lines.  No LDS trip for the field if this pattern streams at the rate of the coalesced one.
template <bool NT>
__global__ __launch_bounds__(256, 2) void operand_kernel(const float* __restrict__ A, int64_t ld, int n, int64_t kps, int rt, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x % rt, split = blockIdx.x / rt;
  const int64_t row0 = (int64_t)tile * 256 + wave * 64;
  const int64_t k0 = (int64_t)split * kps, k1 = (k0 + kps < ld) ? k0 + kps : ld;
  const int lr = lane & 31, lc = (lane >> 5) * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int64_t ra = row0 + lr < n ? row0 + lr : n - 1, rb = row0 + 32 + lr < n ? row0 + 32 + lr : n - 1;
  const float* pa = A + ra * ld + lc;
  const float* pb = A + rb * ld + lc;
  for (int64_t k = k0; k < k1; k += 64) {       // 64 features = 256 bytes of each of the 64 rows: 16 loads per lane
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[2 * i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pa + k + 8 * i)) : *reinterpret_cast<const f32x4*>(pa + k + 8 * i);
      v[2 * i + 1] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pb + k + 8 * i)) : *reinterpret_cast<const f32x4*>(pb + k + 8 * i);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc.x = fmaxf(acc.x, v[i].x); acc.y = fmaxf(acc.y, v[i].y); acc.z = fmaxf(acc.z, v[i].z); acc.w = fmaxf(acc.w, v[i].w);
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

template <bool NT>
__global__ __launch_bounds__(256) void flat_kernel(const float* __restrict__ A, size_t count4, float* __restrict__ out) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const f32x4* p = reinterpret_cast<const f32x4*>(A);
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * stride < count4; i += 8 * stride) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x = fmaxf(acc.x, v[u].x); acc.y = fmaxf(acc.y, v[u].y); acc.z = fmaxf(acc.z, v[u].z); acc.w = fmaxf(acc.w, v[u].w); }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

__global__ void fill_kernel(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 32);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
    p[i] = 270.f + 20.f * (float)(x >> 8) * (1.f / 16777216.f);
  }
}

template <int SEG, bool NT>
static float run(const float* A, int64_t ld, int n, int S, float* out) {
  const int64_t kps = (ld / 256 + S - 1) / S * 256;        // multiples of 1024 bytes
  const int s_eff = (int)((ld + kps - 1) / kps), rt = (n + 255) / 256;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rowseg_kernel<SEG, NT>), dim3(rt * s_eff), dim3(256), 0, 0, A, ld, n, kps, rt, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
  }
  printf("%4d bytes of a row per wave instruction (%2d rows / instruction)%s, %3d splits: %.3f ms -> %.0f GB/s\n", SEG, 1024 / SEG, NT ? " nt" : "   ", s_eff, best,
         (double)n * ld * 4 / best / 1e6);
  return best;
}

int main() {
  const int n = 10000; const int64_t ld = 1036800;
  float *A, *out;
  CK(hipMalloc(&A, (size_t)n * ld * 4)); CK(hipMalloc(&out, 1 << 22));
  hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, A, (size_t)n * ld);
  CK(hipDeviceSynchronize());
  for (int round = 0; round < 2; ++round) {
    for (int S : {96, 48}) {
      run<128, false>(A, ld, n, S, out);
      run<256, false>(A, ld, n, S, out);
      run<512, false>(A, ld, n, S, out);
      run<1024, false>(A, ld, n, S, out);
      run<128, true>(A, ld, n, S, out);
      run<512, true>(A, ld, n, S, out);
      for (int nt = 0; nt < 2; ++nt) {
        const int64_t kps = (ld / 256 + S - 1) / S * 256;
        const int s_eff = (int)((ld + kps - 1) / kps), rt = (n + 255) / 256;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipEventRecord(e0));
          if (nt) hipLaunchKernelGGL(operand_kernel<true>, dim3(rt * s_eff), dim3(256), 0, 0, A, ld, n, kps, rt, out);
          else hipLaunchKernelGGL(operand_kernel<false>, dim3(rt * s_eff), dim3(256), 0, 0, A, ld, n, kps, rt, out);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
        }
        printf("  32 bytes of a row per wave instruction (32 rows / instruction: the MFMA operand layout)%s, %3d splits: %.3f ms -> %.0f GB/s\n", nt ? " nt" : "   ",
               s_eff, best, (double)n * ld * 4 / best / 1e6);
      }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nt = 0; nt < 2; ++nt) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        if (nt) hipLaunchKernelGGL(flat_kernel<true>, dim3(256 * 16), dim3(256), 0, 0, A, (size_t)n * ld / 4, out);
        else hipLaunchKernelGGL(flat_kernel<false>, dim3(256 * 16), dim3(256), 0, 0, A, (size_t)n * ld / 4, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
      }
      printf("flat read of the same bytes%s: %.3f ms -> %.0f GB/s\n", nt ? " nt" : "   ", best, (double)n * ld * 4 / best / 1e6);
    }
  }
  return 0;
}
