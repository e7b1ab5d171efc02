// Microbenchmark for a fused power-iteration design: cost of a flag-based barrier among the 32 workgroups of
// one XCD (vs 32 workgroups spread over all XCDs) and of int64 atomic accumulation of a 32x64 tile.
// hipcc --offload-arch=gfx950 -O3 tools/probes/sync_probe.hip -o build/sync_probe && build/sync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void sync_kernel(int* flags, long long* ybuf, int iters, int mode, int natom,
                                                   int variant, int* err, unsigned long long* xcc_ids) {
  const int wg = blockIdx.x;
  const int group = mode ? (wg / 32) : (wg % 8);
  if (threadIdx.x == 0) {
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc_ids[wg] = xcc & 0xf;
  }
  for (int it = 0; it < iters; ++it) {
    long long* y = ybuf + ((size_t)group * iters + it) * 2048;
    long long keep = 0;
    for (int i = threadIdx.x; i < natom; i += 256) {
      if (variant == 0)
        __hip_atomic_fetch_add(&y[i], (long long)(wg / 8 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else   // returning atomic: the wave waits for completion before it signals
        keep += __hip_atomic_fetch_add(&y[i], (long long)(wg / 8 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (variant == 0) __threadfence();
    if (keep == -12345) *err = 3;
    __syncthreads();
    int* f = flags + (size_t)group * iters + it;
    if (threadIdx.x == 0) {
      if (variant == 0) {
        __hip_atomic_fetch_add(f, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < 32)
          if (++spins > 20000000) { *err = 1; break; }
      } else {
        __hip_atomic_fetch_add(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 32)
          if (++spins > 20000000) { *err = 1; break; }
      }
    }
    __syncthreads();
    // consume: read the reduced tile; with group = wg % 8 the members are wg/8 = 0..31 -> sum 1..32 = 528
    for (int i = threadIdx.x; i < natom; i += 256) {
      const long long v = __hip_atomic_load(&y[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (mode == 0 && v != 528) atomicAdd(err + 1, 1);
    }
  }
}

int main() {
  int iters = 2000;
  int *flags, *err; long long* ybuf; unsigned long long* xcc;
  CK(hipMalloc(&flags, sizeof(int) * 8 * iters));
  CK(hipMalloc(&ybuf, sizeof(long long) * 8 * (size_t)iters * 2048));
  CK(hipMalloc(&err, sizeof(int) * 2));
  CK(hipMalloc(&xcc, sizeof(unsigned long long) * 256));
  for (int variant = 0; variant < 2; ++variant)
  for (int mode = 0; mode < 2; ++mode)
    for (int natom : {0, 2048}) {
      CK(hipMemset(flags, 0, sizeof(int) * 8 * iters));
      CK(hipMemset(ybuf, 0, sizeof(long long) * 8 * (size_t)iters * 2048));
      CK(hipMemset(err, 0, sizeof(int) * 2));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      void* args[] = {&flags, &ybuf, &iters, &mode, &natom, &variant, &err, &xcc};
      CK(hipEventRecord(e0));
      CK(hipLaunchCooperativeKernel((void*)sync_kernel, dim3(256), dim3(256), args, 0, 0));
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      int herr[2]; CK(hipMemcpy(herr, err, sizeof(int) * 2, hipMemcpyDeviceToHost));
      printf("variant %d (%s) mode %d (%s) natom %4d: %.3f us per barrier+tile  (err %d, wrong tile values %d)\n", variant,
             variant ? "relaxed + returning atomics" : "fence + release/acquire", mode,
             mode ? "group = wg/32" : "group = wg%8", natom, 1e3 * ms / iters, herr[0], herr[1]);
    }
  std::vector<unsigned long long> h(256);
  CK(hipMemcpy(h.data(), xcc, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost));
  printf("XCC id of workgroups 0..15:");
  for (int i = 0; i < 16; ++i) printf(" %llu", h[i]);
  int same = 1; for (int i = 0; i < 256; ++i) if (h[i] != h[i % 8]) same = 0;
  printf("\nworkgroup -> XCC is round robin (wg %% 8): %s\n", same ? "yes" : "no");
  return 0;
}
