// Where do the 80 us of chol_rinv_kernel go?  Timestamps (wall_clock64, 100 MHz) at its phase boundaries, l = 60 of L = 64,
// a Gram matrix of a random 4096 x 60 panel.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/rinv_phase_probe.hip -o build/rinv_phase_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../../xeofs_amd/csrc/eofx_kernels.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using namespace eofx;
int main() {
  const int L = 64;
  for (int l : {60, 30, 64}) {
    std::vector<double> P(4096 * l), G(L * L, 0.0);
    unsigned x = 12345u;
    for (auto& v : P) { x = x * 1664525u + 1013904223u; v = ((x >> 8) * (1.0 / 16777216.0)) - 0.5; }
    for (int i = 0; i < l; ++i) for (int j = 0; j < l; ++j) { double s = 0; for (int r = 0; r < 4096; ++r) s += P[r * l + i] * P[r * l + j]; G[i * L + j] = s; }
    double *dG, *dR; unsigned long long* dT;
    CK(hipMalloc(&dG, sizeof(double) * L * L)); CK(hipMalloc(&dR, sizeof(double) * L * L)); CK(hipMalloc(&dT, 64)); CK(hipMemset(dT, 0, 64));
    CK(hipMemcpy(dG, G.data(), sizeof(double) * L * L, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemset(dT, 0, 64));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(chol_rinv_kernel, dim3(1), dim3(256), 0, 0, (const double*)dG, L, l, dR, 1e-13, (const double*)nullptr, dT);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
    }
    unsigned long long T[6]; CK(hipMemcpy(T, dT, sizeof(T), hipMemcpyDeviceToHost));
    std::vector<double> R(L * L); CK(hipMemcpy(R.data(), dR, sizeof(double) * L * L, hipMemcpyDeviceToHost));
    // check: Rinv^T G Rinv = I
    double err = 0;
    for (int i = 0; i < l; ++i) for (int j = 0; j < l; ++j) {
      double s = 0;
      for (int a = 0; a < l; ++a) { double t = 0; for (int b = 0; b < l; ++b) t += G[a * L + b] * R[b * L + j]; s += R[a * L + i] * t; }
      err = fmax(err, fabs(s - (i == j)));
    }
    printf("l %2d: events %.1f us | load %.2f  factor %.2f (panels %.2f)  inverse %.2f  store %.2f us | max |Rinv^T G Rinv - I| %.2e\n", l, best * 1e3,
           (T[1] - T[0]) * 0.01, (T[2] - T[1]) * 0.01, T[5] * 0.01, (T[3] - T[2]) * 0.01, (T[4] - T[3]) * 0.01, err);
  }
  return 0;
}
