// What bounds the in-place sample-side product?  axb_f16_kernel at config 4 (10000 x 1036800, L = 64) with parts
// switched off (its DBG template bits): 1 no MFMA, 2 no conversion either, 4 no B / map loads, 8 non-temporal A loads.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/axb_probe.hip -o build/axb_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "../../xeofs_amd/csrc/eofx_kernels.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using namespace eofx;

template <int DBG>
static void run(const float* A, int64_t ld, int n, const float* aff, const float* B, float* C, int64_t rows_pad, int S, const float* bmax) {
  const int64_t K = ld;
  const int64_t kps = (K / 64 + S - 1) / S * 64;
  const int s_eff = (int)((K + kps - 1) / kps);
  const int rt = (int)(rows_pad / 256);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((axb_f16_kernel<4, DBG>), dim3(8 * rt * ((s_eff + 7) / 8), 1), dim3(256), 0, 0, A, ld, n, ld, aff, ld, B, 64, C, 64,
                       rows_pad, K, kps, s_eff, rt, 0, 512.0f, bmax);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  printf("DBG %2d  splits %3d: %.3f ms -> %.0f GB/s\n", DBG, s_eff, best, (double)n * ld * 4 / best / 1e6);
}

__global__ void fill_kernel(float* p, size_t n, float lo, float hi, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)(i * 2654435761u) ^ seed ^ (unsigned)(i >> 32);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = lo + (hi - lo) * (float)(x >> 8) * (1.f / 16777216.f);
  }
}

int main(int argc, char** argv) {
  const int n = 10000; const int64_t ld = 1036800, rows_pad = 10240;
  float *A, *aff, *B, *C, *bmax;
  CK(hipMalloc(&A, (size_t)n * ld * 4)); CK(hipMalloc(&aff, (size_t)3 * ld * 4)); CK(hipMalloc(&B, (size_t)ld * 64 * 4));
  CK(hipMalloc(&C, (size_t)264 * rows_pad * 64 * 4)); CK(hipMalloc(&bmax, 4));
  if (argc > 1) {   // zeros: lower power, higher clocks -- not representative
    CK(hipMemset(A, 0, (size_t)n * ld * 4)); CK(hipMemset(aff, 0, (size_t)3 * ld * 4)); CK(hipMemset(B, 0, (size_t)ld * 64 * 4));
  } else {
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, A, (size_t)n * ld, 270.f, 290.f, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, aff, (size_t)ld, 279.f, 281.f, 2u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, aff + ld, (size_t)ld, -1e-5f, 1e-5f, 3u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, aff + 2 * ld, (size_t)ld, 0.5f, 1.5f, 4u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, B, (size_t)ld * 64, -1.f, 1.f, 5u);
    CK(hipDeviceSynchronize());
  }
  const float one = 1.f; CK(hipMemcpy(bmax, &one, 4, hipMemcpyHostToDevice));
  for (int S : {96}) {
    run<0>(A, ld, n, aff, B, C, rows_pad, S, bmax);
    run<1>(A, ld, n, aff, B, C, rows_pad, S, bmax);
    run<3>(A, ld, n, aff, B, C, rows_pad, S, bmax);
    run<7>(A, ld, n, aff, B, C, rows_pad, S, bmax);
    run<8>(A, ld, n, aff, B, C, rows_pad, S, bmax);
    run<16>(A, ld, n, aff, B, C, rows_pad, S, bmax);
    run<20>(A, ld, n, aff, B, C, rows_pad, S, bmax);
    run<32>(A, ld, n, aff, B, C, rows_pad, S, bmax);
    run<0>(A, ld, n, aff, B, C, rows_pad, S, bmax);
  }
  return 0;
}
