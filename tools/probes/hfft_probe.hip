// Stand-alone check + timing of the fused Hilbert kernel (xeofs_amd/csrc/eofx_hfft.hpp).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I xeofs_amd/csrc tools/probes/hfft_probe.hip -o build/hfft_probe
//   build/hfft_probe n p [padding=1] [want_real=0] [reps=3]
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>
#include <vector>
#include "eofx_hfft.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static double kappa(int64_t N, int64_t d) {
  int64_t m = d % N; if (m < 0) m += N;
  if (m == 0) return 0.0;
  const double x = M_PI * (double)d / (double)N;
  if (N % 2 == 0) return (m % 2) ? (2.0 / (double)N) / std::tan(x) : 0.0;
  const double sgn = (std::llabs(d) % 2) ? -1.0 : 1.0;
  return (1.0 / std::tan(x) - sgn / std::sin(x)) / (double)N;
}

static void fft_host(std::vector<std::complex<double>>& a) {   // forward, radix 2, in place
  const size_t N = a.size();
  for (size_t i = 1, j = 0; i < N; ++i) {
    size_t bit = N >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  for (size_t len = 2; len <= N; len <<= 1) {
    for (size_t i = 0; i < N; i += len)
      for (size_t k = 0; k < len / 2; ++k) {
        const double ang = -2.0 * M_PI * (double)k / (double)len;
        const std::complex<double> w(std::cos(ang), std::sin(ang));
        const auto x = a[i + k], y = a[i + k + len / 2] * w;
        a[i + k] = x + y; a[i + k + len / 2] = x - y;
      }
  }
}

template <int L, int MODE> static void launch(const float* Xt, int64_t n_pad, int n, int64_t p, int padding, const float* hperm,
                                    const float* u, float* Bt, float* At, unsigned* bmax, unsigned* amax, int cus) {
  using PL = hfft::plan<L>;
  auto kern = hfft::hilbert_fft_kernel<L, MODE>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL::lds));
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(163840 / PL::lds, 2048 / PL::WG));
  const int64_t groups = MODE ? p : (p + 1) / 2;
  const int grid = (int)std::min<int64_t>(groups, (int64_t)cus * per_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(PL::WG), PL::lds, 0, Xt, n_pad, n, p, padding, hperm, u, Bt, At, bmax, amax, (const float*)nullptr, (int64_t)0, (const float*)nullptr);
}

int main(int argc, char** argv) {
  const int n = atoi(argv[1]);
  const int64_t p = atoll(argv[2]);
  const int padding = argc > 3 ? atoi(argv[3]) : 1;
  const int want_real = argc > 4 ? atoi(argv[4]) : 0;
  const int reps = argc > 5 ? atoi(argv[5]) : 3;
  const int64_t n_pad = (n + 511) / 512 * 512;   // the library's padding of the sample axis
  int L = 10;
  while ((1 << L) < 2 * n) ++L;
  const int single = (argc > 6 ? atoi(argv[6]) : 0) || L == 15;   // one feature per workgroup through the half-length transform
  if (L > 15 || (single && L < 11)) { printf("n out of range\n"); return 1; }
  const int P = 1 << L;                   // circular length of the real convolution
  const int LK = single ? L - 1 : L;      // log2 of the complex transform the kernel runs
  const int64_t N = padding ? 3 * (int64_t)n : n;
  printf("n=%d p=%lld P=%d L=%d padding=%d single=%d\n", n, (long long)p, P, L, padding, single);
  // filter table
  std::vector<std::complex<double>> c((size_t)P, 0.0);
  for (int64_t d = -(n - 1); d <= n - 1; ++d) c[(size_t)((d % P + P) % P)] = kappa(N, d) / (double)P;
  fft_host(c);
  std::vector<float> hperm((size_t)P);
  double maxre = 0;
  for (int pos = 0; pos < P; ++pos) maxre = std::max(maxre, std::fabs(c[pos].real()));
  if (single) {   // [hm | hp2] over the M = P/2 positions of the half-length transform
    const int M = P / 2;
    for (int pos = 0; pos < M; ++pos) {
      const int64_t k = hfft::position_frequency(LK, pos);
      const double hk = c[(size_t)k].imag(), hkp = k ? c[(size_t)(M - k)].imag() : 0.0;
      hperm[pos] = (float)(hk - hkp);
      hperm[M + pos] = (float)(0.5 * (hk + hkp));
    }
  } else {
    for (int pos = 0; pos < P; ++pos) hperm[pos] = (float)c[(size_t)hfft::position_frequency(L, pos)].imag();
  }
  printf("max |Re spectrum| = %.3e (should be ~0)\n", maxre);
  std::vector<float> hu((size_t)4 * n);
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& v : hu) v = 0.3f * nd(rng);
  // data: a few distinct rows, tiled
  const int nrows = 5;
  std::vector<float> rows((size_t)nrows * n_pad, 0.f);
  for (int r = 0; r < nrows; ++r)
    for (int i = 0; i < n; ++i) rows[(size_t)r * n_pad + i] = nd(rng) + 0.5f * r + 0.001f * i * (r - 2);
  float *dX, *dB, *dA = nullptr, *dh, *du;
  unsigned *dmax;
  const size_t bytes = (size_t)p * n_pad * 4;
  CK(hipMalloc((void**)&dX, bytes)); CK(hipMalloc((void**)&dB, bytes));
  if (want_real) CK(hipMalloc((void**)&dA, bytes));
  CK(hipMalloc((void**)&dh, P * 4)); CK(hipMalloc((void**)&du, (4 * n + 4) * 4)); CK(hipMalloc((void**)&dmax, 8));
  CK(hipMemset(dmax, 0, 8));
  CK(hipMemset(dB, 0xFF, bytes));   // NaN everywhere: every element of the padded rows has to be written
  if (dA) CK(hipMemset(dA, 0xFF, bytes));
  CK(hipMemcpy(dh, hperm.data(), P * 4, hipMemcpyHostToDevice));
  {
    std::vector<float> il((size_t)4 * n + 4);   // the four vectors interleaved per sample, then their means
    for (int i = 0; i < n; ++i) for (int k = 0; k < 4; ++k) il[(size_t)4 * i + k] = hu[(size_t)k * n + i];
    for (int k = 0; k < 4; ++k) { double m = 0; for (int i = 0; i < n; ++i) m += hu[(size_t)k * n + i]; il[(size_t)4 * n + k] = (float)(m / n); }
    CK(hipMemcpy(du, il.data(), (4 * n + 4) * 4, hipMemcpyHostToDevice));
  }
  {
    std::vector<float> chunk;
    const int64_t per = std::min<int64_t>(p, 4096);
    chunk.resize((size_t)per * n_pad);
    for (int64_t f = 0; f < per; ++f) std::copy(rows.begin() + (f % nrows) * n_pad, rows.begin() + (f % nrows + 1) * n_pad, chunk.begin() + f * n_pad);
    for (int64_t f0 = 0; f0 < p; f0 += per) {   // per is a multiple of nrows?  keep row identity = f % nrows by aligning
      const int64_t cnt = std::min(per, p - f0);
      if (f0 % nrows != 0) { for (int64_t f = 0; f < cnt; ++f) std::copy(rows.begin() + ((f0 + f) % nrows) * n_pad, rows.begin() + ((f0 + f) % nrows + 1) * n_pad, chunk.begin() + f * n_pad); }
      CK(hipMemcpy(dX + f0 * n_pad, chunk.data(), (size_t)cnt * n_pad * 4, hipMemcpyHostToDevice));
    }
  }
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  auto run = [&]() {
    switch (LK + 100 * single) {
#define CASE(LL) case LL: launch<LL, 0>(dX, n_pad, n, p, padding, dh, du, dB, dA, dmax, dmax + 1, cus); break; \
                 case 100 + LL: launch<LL, 1>(dX, n_pad, n, p, padding, dh, du, dB, dA, dmax, dmax + 1, cus); break;
      CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
#undef CASE
    }
  };
  run();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) run();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double gb = (double)p * n * 4.0 * (want_real ? 4 : 2) / 1e9;
  printf("kernel %.3f ms  (%.1f GB algorithmic -> %.0f GB/s)\n", ms, gb, gb / ms * 1e3);
  // reference for rows 0..nrows-1 and the last feature
  std::vector<double> kap((size_t)2 * n - 1);
  for (int d = -(n - 1); d <= n - 1; ++d) kap[(size_t)(d + n - 1)] = kappa(N, d);
  double worst = 0, worst_a = 0, scale = 0;
  std::vector<float> got((size_t)n_pad), gota((size_t)n_pad);
  float refmax = 0;
  for (int r = 0; r < nrows + 1; ++r) {
    const int64_t f = r < nrows ? std::min<int64_t>(r, p - 1) : p - 1;
    const float* y = rows.data() + (f % nrows) * n_pad;
    double sy = 0, sty = 0; const double tbar = 0.5 * (n - 1);
    for (int i = 0; i < n; ++i) { sy += y[i]; sty += (i - tbar) * y[i]; }
    const double stt = (double)n * ((double)n * n - 1.0) / 12.0, c1 = n > 1 ? sty / stt : 0.0, c0 = sy / n - c1 * tbar;
    const float a1 = (float)(y[0] - c0), a2 = (float)(y[n - 1] - (c0 + c1 * (n - 1))), a3 = (float)c0, a4 = (float)c1;
    std::vector<double> out((size_t)n);
    double mean = 0;
    for (int i = 0; i < n; ++i) {
      double acc = 0;
      for (int s = 0; s < n; ++s) acc += kap[(size_t)(i - s + n - 1)] * (double)y[s];
      if (padding) acc += (double)a1 * hu[i] + (double)a2 * hu[n + i] + (double)a3 * hu[2 * n + i] + (double)a4 * hu[3 * n + i];
      out[i] = acc; mean += acc;
    }
    mean /= n;
    CK(hipMemcpy(got.data(), dB + f * n_pad, n_pad * 4, hipMemcpyDeviceToHost));
    if (want_real) CK(hipMemcpy(gota.data(), dA + f * n_pad, n_pad * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n_pad; ++i) {
      const double ref = i < n ? out[i] - mean : 0.0;
      worst = std::max(worst, std::isfinite(got[i]) ? std::fabs(ref - (double)got[i]) : 1e30);
      scale = std::max(scale, std::fabs(ref));
      refmax = std::max(refmax, (float)std::fabs(ref));
      if (want_real) worst_a = std::max(worst_a, std::fabs((i < n ? (double)y[i] - sy / n : 0.0) - (double)gota[i]));
    }
  }
  unsigned hm[2]; CK(hipMemcpy(hm, dmax, 8, hipMemcpyDeviceToHost));
  float fm[2]; memcpy(fm, hm, 8);
  printf("max |err| = %.3e  (scale %.3e, rel %.2e)   real-part err %.3e   absmax dev %.5f (checked rows' max %.5f) amax %.5f\n",
         worst, scale, worst / scale, worst_a, fm[0], refmax, fm[1]);
  return (worst / scale < 2e-5) ? 0 : 2;
}
