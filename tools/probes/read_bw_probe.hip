// Practical HBM read ceiling on this box: streaming sum over a 41 GB buffer with 16 B non-temporal loads.
// hipcc --offload-arch=gfx950 -O3 tools/probes/read_bw_probe.hip -o build/read_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ p, size_t n4, float* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  f32x4 acc = {0, 0, 0, 0};
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  for (; i < n4; i += stride) acc += p[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *out = 1.f;
}

int main() {
  const size_t bytes = (size_t)41472 * 1000 * 1000;   // the C4 matrix
  f32x4* buf; float* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(buf, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(read_kernel<8>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 16, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("read 41.5 GB, %5d workgroups x 256, unroll 8: %.3f ms -> %.0f GB/s\n", blocks, ms, bytes / ms / 1e6);
    }
  }
  return 0;
}
