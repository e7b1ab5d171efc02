// gram_probe.hip -- correctness and speed of the split-fp16 Gram kernels (xeofs_amd/csrc/eofx_gram.hpp) on their own.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I xeofs_amd/csrc tools/probes/gram_probe.hip -o build/probe/gram_probe
//   build/probe/gram_probe [n] [p] [S]        (defaults: the config-3 field 5000 x 129600, S chosen by the plan)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "eofx_gram.hpp"

using namespace eofx;

#define HC(e)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (e);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "%s -> %s (%s:%d)\n", #e, hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                             \
    }                                                                                      \
  } while (0)

static double half_to_double(uint16_t h) {
  const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
  double v = e == 0 ? std::ldexp((double)m, -24) : std::ldexp((double)(m | 1024), e - 25);
  return s ? -v : v;
}

__global__ void fill_kernel(float* x, int64_t n, unsigned seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    unsigned h2 = h * 0x9e3779b9u + 12345u;
    h2 ^= h2 >> 15;
    // roughly normal: sum of four uniforms, plus an offset so that the map has something to subtract
    const float u = ((h & 0xffff) + (h >> 16) + (h2 & 0xffff) + (h2 >> 16)) * (1.f / 65536.f) - 2.f;
    x[i] = 280.f + 7.f * u;
  }
}

static int run_case(int64_t n, int64_t p, int S_req, bool check, int reps, int var = 1) {
  const int64_t n_pad = (n + 511) / 512 * 512, kpad = (p + 63) / 64 * 64, aff_ld = (p + 511) / 512 * 512;
  float *X, *aff, *G;
  _Float16* planes;
  HC(hipMalloc(&X, sizeof(float) * n * p));
  HC(hipMalloc(&aff, sizeof(float) * 3 * aff_ld));
  HC(hipMalloc(&planes, (size_t)n_pad * kpad * 4));
  HC(hipMalloc(&G, sizeof(float) * n_pad * n_pad));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, X, n * p, 17u);
  std::vector<float> haff(3 * aff_ld, 0.f);
  std::mt19937 rng(5);
  for (int64_t k = 0; k < p; ++k) {
    const double mean = 280.0 + 0.001 * (rng() % 1000);
    haff[k] = (float)mean;
    haff[aff_ld + k] = (float)(mean - (double)haff[k]);
    haff[2 * aff_ld + k] = (k % 97 == 13) ? 0.f : 0.5f + (rng() % 1000) * 0.001f;     // some masked features
  }
  HC(hipMemcpy(aff, haff.data(), sizeof(float) * haff.size(), hipMemcpyHostToDevice));
  const float a_scale = std::ldexp(1.f, 14 - 6);    // |x'| < 2^6
  hipEvent_t e0, e1;
  HC(hipEventCreate(&e0));
  HC(hipEventCreate(&e1));
  float ms_split = 0.f;
  for (int r = 0; r < reps; ++r) {
    HC(hipEventRecord(e0));
    hipLaunchKernelGGL(planes_split_kernel, dim3((unsigned)((kpad + 2047) / 2048), (unsigned)((n_pad + 63) / 64)), dim3(256), 0, 0, X, p, n, p, aff,
                       aff_ld, a_scale, planes, n_pad, kpad, 64);
    HC(hipEventRecord(e1));
    HC(hipEventSynchronize(e1));
    HC(hipEventElapsedTime(&ms_split, e0, e1));
  }
  HC(hipGetLastError());
  const int nt = (int)(n_pad / GR_BM), nst = (int)(kpad / GR_BK);
  GramPlan pl;
  gram_plan_build(nt, nt, true, nst, S_req, pl);
  GramItem* items;
  int2* tiles;
  float* Cp;
  HC(hipMalloc(&items, sizeof(GramItem) * pl.items.size()));
  HC(hipMalloc(&tiles, sizeof(int2) * pl.tiles.size()));
  HC(hipMalloc(&Cp, sizeof(float) * (size_t)pl.T * pl.S * GR_BM * GR_BM));
  HC(hipMemcpy(items, pl.items.data(), sizeof(GramItem) * pl.items.size(), hipMemcpyHostToDevice));
  HC(hipMemcpy(tiles, pl.tiles.data(), sizeof(int2) * pl.tiles.size(), hipMemcpyHostToDevice));
  const float out_scale = 1.f / (a_scale * a_scale);
  float ms_gram = 0.f, ms_fin = 0.f, best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    HC(hipEventRecord(e0));
    hipLaunchKernelGGL(gram_nt_kernel, dim3(pl.grid), dim3(512), 0, 0, planes, planes, kpad * 4, items, Cp, out_scale);
    HC(hipEventRecord(e1));
    HC(hipEventSynchronize(e1));
    HC(hipEventElapsedTime(&ms_gram, e0, e1));
    best = std::min(best, ms_gram);
    HC(hipEventRecord(e0));
    hipLaunchKernelGGL(gram_finish_kernel, dim3(pl.T, 16), dim3(256), 0, 0, Cp, tiles, pl.S, G, n_pad);
    HC(hipEventRecord(e1));
    HC(hipEventSynchronize(e1));
    HC(hipEventElapsedTime(&ms_fin, e0, e1));
  }
  HC(hipGetLastError());
  const double flop = 2.0 * 3.0 * (double)pl.T * GR_BM * GR_BM * (double)kpad;
  printf("var %d n %lld (pad %lld) p %lld: tiles %d S %d (%d stages each) grid %d | split %.3f ms (%.0f GB/s r+w) | gram %.3f ms best %.3f -> %.0f TFLOP/s fp16 issued "
         "(%.1f %% of 2500) | finish %.3f ms\n",
         var, (long long)n, (long long)n_pad, (long long)p, pl.T, pl.S, pl.st_per_split, pl.grid, ms_split,
         (n * p * 4.0 + n_pad * kpad * 4.0) / ms_split * 1e-6, ms_gram, best, flop / best * 1e-9, flop / best * 1e-9 / 25.0, ms_fin);
  int bad = 0;
  if (check) {
    std::vector<uint16_t> hp((size_t)n_pad * kpad * 2);
    std::vector<float> hx((size_t)n * p), hG((size_t)n_pad * n_pad);
    HC(hipMemcpy(hp.data(), planes, hp.size() * 2, hipMemcpyDeviceToHost));
    HC(hipMemcpy(hx.data(), X, hx.size() * 4, hipMemcpyDeviceToHost));
    HC(hipMemcpy(hG.data(), G, hG.size() * 4, hipMemcpyDeviceToHost));
    // planes vs the map in double
    double worst = 0.0;
    for (int64_t r = 0; r < n_pad; ++r)
      for (int64_t k = 0; k < kpad; ++k) {
        const double hi = half_to_double(hp[(size_t)r * 2 * kpad + (k / 32) * 64 + (k & 31)]);
        const double lo = half_to_double(hp[(size_t)r * 2 * kpad + (k / 32) * 64 + 32 + (k & 31)]);
        double want = 0.0;
        if (r < n && k < p && haff[2 * aff_ld + k] != 0.f)
          want = (((double)hx[(size_t)r * p + k] - haff[k]) - haff[aff_ld + k]) * haff[2 * aff_ld + k] * a_scale;
        const double err = std::fabs(hi + lo - want);
        if (err > worst) worst = err;
      }
    printf("  planes: max |hi + lo - map| = %.3e (scaled units; values up to %.0f)\n", worst, 64.0 * a_scale);
    if (worst > 64.0 * a_scale * 1e-6) ++bad;
    // Gram vs double from the planes (hh + hl + lh)
    std::vector<double> hh((size_t)n_pad * kpad), ll((size_t)n_pad * kpad);
    for (int64_t r = 0; r < n_pad; ++r)
      for (int64_t k = 0; k < kpad; ++k) {
        hh[(size_t)r * kpad + k] = half_to_double(hp[(size_t)r * 2 * kpad + (k / 32) * 64 + (k & 31)]);
        ll[(size_t)r * kpad + k] = half_to_double(hp[(size_t)r * 2 * kpad + (k / 32) * 64 + 32 + (k & 31)]);
      }
    double werr = 0.0, gmax = 0.0;
    for (int64_t i = 0; i < n_pad; ++i)
      for (int64_t j = 0; j < n_pad; ++j) {
        double acc = 0.0;
        const double *hi_ = &hh[(size_t)i * kpad], *hj = &hh[(size_t)j * kpad], *li = &ll[(size_t)i * kpad], *lj = &ll[(size_t)j * kpad];
        for (int64_t k = 0; k < kpad; ++k) acc += hi_[k] * hj[k] + hi_[k] * lj[k] + li[k] * hj[k];
        acc *= out_scale;
        gmax = std::max(gmax, std::fabs(acc));
        werr = std::max(werr, std::fabs(acc - (double)hG[(size_t)i * n_pad + j]));
      }
    printf("  gram: max |G - ref| = %.3e, max |G| = %.3e -> rel %.2e\n", werr, gmax, werr / gmax);
    if (!(werr <= 2e-6 * gmax)) ++bad;
  }
  HC(hipFree(X)); HC(hipFree(aff)); HC(hipFree(planes)); HC(hipFree(G)); HC(hipFree(items)); HC(hipFree(tiles)); HC(hipFree(Cp));
  return bad;
}

int main(int argc, char** argv) {
  int bad = 0;
  if (argc <= 1) {
    for (int var = 0; var < 1; ++var) {
      bad += run_case(300, 1000, 1, true, 1, var);
      bad += run_case(300, 1000 - 32, 1, true, 1, var);
      bad += run_case(300, 1000 - 64, 1, true, 1, var);
      bad += run_case(700, 2100, 3, true, 1, var);
      bad += run_case(1000, 999, 0, true, 1, var);
    }
    printf(bad ? "CHECK FAILED\n" : "checks ok\n");
    for (int rep = 0; rep < 2; ++rep)
      for (int var = 0; var < 1; ++var)
        for (int S : {0, 1, 9}) bad += run_case(5000, 129600, S, false, 3, var);
    bad += run_case(10000, 129600, 0, false, 2, 0);
  } else {
    const int64_t n = atoll(argv[1]), p = argc > 2 ? atoll(argv[2]) : 129600;
    bad += run_case(n, p, argc > 3 ? atoi(argv[3]) : 0, false, 3);
  }
  return bad ? 1 : 0;
}
