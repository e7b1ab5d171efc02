// Practical ceiling of the preprocess "apply" pass: read N bytes, write 2 N bytes (two output streams).
// hipcc --offload-arch=gfx950 -O3 tools/probes/copy_bw_probe.hip -o build/copy_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy2_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ o1, f32x4* __restrict__ o2,
                                                    size_t n4, int nt_store) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = __builtin_nontemporal_load(in + i);
    if (nt_store) {
      __builtin_nontemporal_store(v, o1 + i);
      __builtin_nontemporal_store(v * 2.f, o2 + i);
    } else {
      o1[i] = v;
      o2[i] = v * 2.f;
    }
  }
}

int main() {
  const size_t bytes = (size_t)41472 * 1000 * 1000;
  f32x4 *in, *o1, *o2;
  CK(hipMalloc(&in, bytes)); CK(hipMalloc(&o1, bytes)); CK(hipMalloc(&o2, bytes));
  CK(hipMemset(in, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nt = 0; nt < 2; ++nt)
    for (int blocks : {2048, 8192, 32768}) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(copy2_kernel, dim3(blocks), dim3(256), 0, 0, in, o1, o2, bytes / 16, nt);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("read 41.5 GB + write 2 x 41.5 GB, %5d workgroups, %s stores: %.3f ms -> %.0f GB/s total traffic\n", blocks,
                        nt ? "non-temporal" : "plain", ms, 3.0 * bytes / ms / 1e6);
      }
    }
  return 0;
}
