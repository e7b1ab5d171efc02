// Can the Hilbert kernel read its series from the field WHERE IT LIES (sample-major: row = sample, features contiguous)?
// One workgroup of 1024 threads owns a feature pair (8 bytes of every row); the 32 workgroups of an XCD take 32 ADJACENT
// pairs (256 bytes of every row) at the same time, so every 128-byte line is fetched from HBM once per XCD and hit in its
// 4 MB L2 by the other 15 pairs (working set 8000 rows x 256 B = 2 MB).  Measured here: the time of the loads alone and with
// `spin` dependent fma per loaded value standing in for the transform (the real kernel: ~15 us per pair), for the blocked
// mapping and for a naive one (consecutive pairs on consecutive workgroups = different XCDs).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/colread_probe.hip -o build/colread_probe ; build/colread_probe [n] [p] [spin]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MAP>
__global__ __launch_bounds__(1024) void colread_kernel(const float* __restrict__ X, int64_t ld, int n, int64_t npairs, int spin,
                                                       float* out) {
  const int t = threadIdx.x;
  const int w = blockIdx.x, G = gridDim.x;
  float acc = 0.f;
  // MAP 1: workgroup w sits on XCD w % 8 (round-robin dispatch); per sweep the XCD takes block (sweep * 8 + xcd) of G/8 pairs
  const int per_xcd = G / 8;
  const int64_t nblocks = (npairs + per_xcd - 1) / per_xcd;
  for (int64_t it = 0;; ++it) {
    int64_t pair;
    if (MAP == 1) {
      const int64_t b = it * 8 + (w & 7);
      if (b >= nblocks) break;
      pair = b * per_xcd + (w >> 3);
    } else {
      pair = it * G + w;
      if (it * G >= npairs) break;
    }
    if (pair >= npairs) continue;
    const float* col = X + 2 * pair;
    f32x2 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int row = t + 1024 * c;
      v[c] = row < n ? __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(col + (int64_t)row * ld)) : f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float a = v[c].x, b = v[c].y;
      for (int s = 0; s < spin; ++s) { a = __builtin_fmaf(a, 1.0000001f, b); b = __builtin_fmaf(b, 0.9999999f, a); }
      acc += a + b;
    }
    __syncthreads();
  }
  if (acc == 12345.678f) *out = acc;
}

template <int MAP>
static void run(const float* X, int64_t ld, int n, int64_t npairs, int spin, int grid, float* out, const char* name) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((colread_kernel<MAP>), dim3(grid), dim3(1024), 0, 0, X, ld, n, npairs, spin, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 2) printf("%-34s grid %4d spin %3d: %8.3f ms -> %6.0f GB/s of field bytes (%.2f us per pair and workgroup)\n", name, grid, spin, ms,
                         (double)n * npairs * 8 / ms / 1e6, ms * 1e3 * grid / (double)npairs);
  }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8000;
  const int64_t p = argc > 2 ? atoll(argv[2]) : 1036800;
  const int spin = argc > 3 ? atoi(argv[3]) : 0;
  float *X, *out;
  CK(hipMalloc(&X, (size_t)n * p * 4)); CK(hipMalloc(&out, 4));
  CK(hipMemset(X, 0, (size_t)n * p * 4));
  const int64_t npairs = p / 2;
  for (int sp : {0, spin}) {
    run<1>(X, p, n, npairs, sp, 256, out, "XCD-blocked (32 adjacent pairs)");
    run<0>(X, p, n, npairs, sp, 256, out, "naive (pair = sweep * grid + w)");
    if (sp == spin) break;
  }
  // sample-contiguous reference: the same bytes as 8000 x 4-byte rows per feature (what the kernel reads today)
  return 0;
}
