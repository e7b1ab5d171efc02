// A workgroup fills its whole LDS allocation (bytes given at launch) with a pattern, idles, and checks it: does anything write into
// another workgroup's LDS?  Also a register canary: every lane keeps 64 VGPRs with a pattern through the idle loop.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/lds_canary.hip -o build/liblds_canary.so
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(256) void lds_canary_kernel(unsigned* __restrict__ errs, int words, int spin) {
  extern __shared__ unsigned sm[];
  const unsigned tag = 0xA5A50000u + blockIdx.x;
  for (int i = threadIdx.x; i < words; i += 256) sm[i] = tag ^ (unsigned)i;
  unsigned r[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) { r[j] = tag + 977u * j + threadIdx.x; asm volatile("" : "+v"(r[j])); }
  __syncthreads();
  for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(64);
  __syncthreads();
  unsigned bad = 0, badr = 0;
  for (int i = threadIdx.x; i < words; i += 256) bad += sm[i] != (tag ^ (unsigned)i);
#pragma unroll
  for (int j = 0; j < 64; ++j) { asm volatile("" : "+v"(r[j])); badr += r[j] != tag + 977u * j + threadIdx.x; }
  if (bad) atomicAdd(errs, bad);
  if (badr) atomicAdd(errs + 1, badr);
}
extern "C" int lds_canary_run(void* stream, unsigned* errs_dev, int blocks, int lds_bytes, int spin) {
  hipFuncSetAttribute((const void*)lds_canary_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, errs_dev, lds_bytes / 4, spin);
  return (int)hipGetLastError();
}
