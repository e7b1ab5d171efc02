// Layout check of v_mfma_f32_16x16x32_f16: A[i][k] = 100 i + k, B = one-hot -> read the fragment maps off D.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int kk, int mode) {
  const int l = threadIdx.x, i16 = l & 15, kg = l >> 4;
  f16x8 a, b;
  for (int t = 0; t < 8; ++t) {
    const int kidx = 8 * kg + t;                  // assumed: lane holds k = 8 kg + t
    a[t] = (_Float16)(float)(mode == 0 ? (i16 + 1) : 1) * (kidx == kk ? 1 : 0);
    b[t] = (_Float16)(float)(mode == 0 ? 1 : (i16 + 1)) * (kidx == kk ? 1 : 0);
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * 4);
  float h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 5, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (0: A[i][5] = i+1, B[5][j] = 1;  1: A = 1, B[5][j] = j+1)\n", mode);
    for (int l = 0; l < 64; l += 1) printf("lane %2d: %4.0f %4.0f %4.0f %4.0f\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  }
  return 0;
}
