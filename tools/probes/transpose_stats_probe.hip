// Can the statistics pass of an in-place preprocess write the sample-contiguous (transposed) RAW field on its way -- for the
// Hilbert stage, which needs whole series per feature and today pays a 14 ms transposing copy on top of the 5.3 ms pass?
// Thread = 4 adjacent features (16-byte loads), 32 rows per block in registers, then 4 x 128 bytes (one full line per
// feature) written as 8 x 16 bytes each; no LDS.  Config-5 shape 8000 x 1 036 800.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/transpose_stats_probe.hip -o build/transpose_stats_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool WRITE, bool STATS>
__global__ __launch_bounds__(256) void tstats_kernel(const float* __restrict__ X, int64_t n, int64_t P, int64_t ld, int64_t rows_per_split,
                                                      float* __restrict__ Xt, int64_t n_pad, double* __restrict__ sum, double* __restrict__ sumsq) {
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= P) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
  const int64_t r1 = (r0 + rows_per_split < n) ? r0 + rows_per_split : n;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  for (int64_t r = r0; r + 32 <= r1; r += 32) {
    f32x4 v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(X + (r + u) * ld + c));
    if (STATS) {
#pragma unroll
      for (int u = 0; u < 32; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const double d = (double)v[u][e]; s[e] += d; q[e] += d * d; }
    }
    if (WRITE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float* dst = Xt + (c + e) * n_pad + r;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          *reinterpret_cast<f32x4*>(dst + 4 * j) = f32x4{v[4 * j][e], v[4 * j + 1][e], v[4 * j + 2][e], v[4 * j + 3][e]};
      }
    }
  }
  if (STATS) {
    const int64_t o = (int64_t)blockIdx.y * P + c;
#pragma unroll
    for (int e = 0; e < 4; ++e) { sum[o + e] = s[e]; sumsq[o + e] = q[e]; }
  }
}

__global__ void fill_kernel(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 32);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
    p[i] = 270.f + 20.f * (float)(x >> 8) * (1.f / 16777216.f);
  }
}

template <bool W, bool S>
static void run(const char* what, const float* X, int64_t n, int64_t P, float* Xt, int64_t n_pad, double* sum, double* sq, int RS) {
  const int64_t rps = (n / RS + 31) / 32 * 32;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((tstats_kernel<W, S>), dim3((unsigned)((P / 4 + 255) / 256), (unsigned)((n + rps - 1) / rps)), dim3(256), 0, 0, X, n, P, P, rps, Xt, n_pad, sum, sq);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
  }
  printf("%-34s row splits %2d: %.2f ms  (%.0f GB/s read%s)\n", what, RS, best, (double)n * P * 4 / best / 1e6, W ? " + the same written" : "");
}

int main() {
  const int64_t n = 8000, P = 1036800, n_pad = 8000;
  float *X, *Xt; double *sum, *sq;
  CK(hipMalloc(&X, (size_t)n * P * 4)); CK(hipMalloc(&Xt, (size_t)n_pad * P * 4));
  CK(hipMalloc(&sum, (size_t)64 * P * 8)); CK(hipMalloc(&sq, (size_t)64 * P * 8));
  hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, X, (size_t)n * P);
  CK(hipDeviceSynchronize());
  for (int RS : {5, 10, 25}) {
    run<false, true>("statistics only", X, n, P, Xt, n_pad, sum, sq, RS);
    run<true, false>("transposed write only", X, n, P, Xt, n_pad, sum, sq, RS);
    run<true, true>("statistics + transposed write", X, n, P, Xt, n_pad, sum, sq, RS);
  }
  // spot check of the transposition
  std::vector<float> a(64), b(64);
  CK(hipMemcpy(a.data(), Xt + (size_t)12345 * n_pad + 640, 256, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 64; ++i) { float x; CK(hipMemcpy(&x, X + (size_t)(640 + i) * P + 12345, 4, hipMemcpyDeviceToHost)); bad += x != a[i]; }
  printf("transposition spot check: %d of 64 wrong\n", bad);
  return 0;
}
