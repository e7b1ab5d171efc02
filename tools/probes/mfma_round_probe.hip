// Does the fp16 MFMA round its float32 accumulation to nearest, or truncate?  One wave accumulates N MFMAs of a constant
// positive product into the same accumulator (the situation of a coherent sum in axb_f16 / atb_f16) and compares with the
// exact value; then the same sum in chunks of C MFMAs added to a float32 total with v_add_f32.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_round_probe.hip -o /tmp/mfma_round_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(int N, int chunk, float av, float bv, float* out) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
  f32x4 acc = {0, 0, 0, 0}, tot = {0, 0, 0, 0};
  for (int i = 0; i < N; ++i) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (chunk && (i + 1) % chunk == 0) { tot += acc; acc = f32x4{0, 0, 0, 0}; }
  }
  if (chunk) tot += acc; else tot = acc;
  if (threadIdx.x == 0) out[0] = tot[0];
}

int main() {
  float* d;
  hipMalloc(&d, 4);
  const float vals[3][2] = {{0.1f, 0.3f}, {1.7f, 0.013f}, {0.7f, 0.9f}};
  for (auto& v : vals) {
    const double fa = (double)(_Float16)v[0], fb = (double)(_Float16)v[1];
    for (int N : {64, 512, 2048, 16384}) {
      for (int chunk : {0, 8, 32}) {
        probe<<<1, 64>>>(N, chunk, v[0], v[1], d);
        float h;
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        const double exact = (double)N * 32.0 * fa * fb;
        printf("a=%.3f b=%.3f N=%6d chunk=%2d: rel err %+.3e\n", v[0], v[1], N, chunk, ((double)h - exact) / exact);
      }
    }
  }
  return 0;
}
