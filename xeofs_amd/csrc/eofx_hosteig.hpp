// Host-side eigen-solver for the Rayleigh-Ritz step of the complex block-Krylov decomposition (eofx_rsvd_c64, row R9
// of SURVEY.md §8a: the reference runs scipy's svds(solver="lobpcg") there, xeofs/linalg/decomposer.py:149-160).
//
// The projected matrix H = K^H (Z^H Z) K is complex Hermitian of order m = (power iterations + 1) x sketch width
// (240 at config 5) and only its leading `nev` (= sketch width) eigenvectors are wanted, as an orthonormal basis of
// their span -- the final projection pass of the decomposition recomputes the values.  So:
//   1. Householder reduction of the Hermitian matrix to a REAL symmetric tridiagonal one (the reflectors are chosen
//      so that every off-diagonal element comes out real), split real / imaginary storage, full rows (the rank-two
//      update touches both triangles: twice the arithmetic of a triangle, but every inner loop is a contiguous
//      element-wise update the compiler vectorises; AVX2 + FMA where the host has them);
//   2. all eigenvalues of the tridiagonal matrix by the implicit QL iteration without vectors (O(m^2));
//   3. the `nev` leading eigenvectors by inverse iteration (tridiagonal LU with partial pivoting, every vector
//      orthogonalised against ALL earlier ones: clusters need no special rule because only the span matters);
//   4. the reflectors applied to those `nev` vectors.
// (16/3 + 16/3) m^3 real operations for step 1, O(nev m^2) for the rest: ~7 ms for m = 240 on one core, against
// ~100 ms for the real symmetric embedding (order 2 m) through the general-purpose solver eofx_host_eigh_f64.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace hosteig {

// p += conj(A[k, :]) v_k  (the column-oriented Hermitian product on full row storage), rows/cols [j0, m)
template <int DUMMY>
static inline __attribute__((always_inline)) void tridiag_core(double* __restrict__ Ar, double* __restrict__ Ai, int m,
                                                               double* __restrict__ d, double* __restrict__ e,
                                                               double* __restrict__ Vr, double* __restrict__ Vi,
                                                               double* __restrict__ taur, double* __restrict__ taui) {
  std::vector<double> pr(m), pi(m), wr(m), wi(m), vr(m), vi(m);
  double* __restrict__ prp = pr.data();
  double* __restrict__ pip = pi.data();
  double* __restrict__ wrp = wr.data();
  double* __restrict__ wip = wi.data();
  double* __restrict__ vrp = vr.data();
  double* __restrict__ vip = vi.data();
  for (int j = 0; j + 1 < m; ++j) {
    const int j1 = j + 1;
    // x = A[j1:, j]
    const double ar = Ar[(size_t)j1 * m + j], ai = Ai[(size_t)j1 * m + j];
    double xn2 = 0.0;
    for (int i = j1 + 1; i < m; ++i) {
      const double xr = Ar[(size_t)i * m + j], xi = Ai[(size_t)i * m + j];
      xn2 += xr * xr + xi * xi;
    }
    d[j] = Ar[(size_t)j * m + j];
    if (xn2 == 0.0 && ai == 0.0) {        // nothing to annihilate
      e[j] = ar;
      taur[j] = taui[j] = 0.0;
      for (int i = j1; i < m; ++i) Vr[(size_t)j * m + i] = Vi[(size_t)j * m + i] = 0.0;
      continue;
    }
    const double nrm = std::sqrt(ar * ar + ai * ai + xn2);
    const double beta = ar >= 0.0 ? -nrm : nrm;
    // tau = (beta - alpha) / beta ; v = x / (alpha - beta), v[0] = 1
    const double tr = (beta - ar) / beta, ti = -ai / beta;
    const double dr = ar - beta, di = ai, dn = 1.0 / (dr * dr + di * di);
    const double ir = dr * dn, ii = -di * dn;             // 1 / (alpha - beta)
    vrp[j1] = 1.0;
    vip[j1] = 0.0;
    for (int i = j1 + 1; i < m; ++i) {
      const double xr = Ar[(size_t)i * m + j], xi = Ai[(size_t)i * m + j];
      vrp[i] = xr * ir - xi * ii;
      vip[i] = xr * ii + xi * ir;
    }
    e[j] = beta;
    taur[j] = tr;
    taui[j] = ti;
    for (int i = j1; i < m; ++i) {
      Vr[(size_t)j * m + i] = vrp[i];
      Vi[(size_t)j * m + i] = vip[i];
    }
    // p = A22 v, column by column: A22[:, k] = conj(A22[k, :])
    for (int i = j1; i < m; ++i) prp[i] = pip[i] = 0.0;
    for (int k = j1; k < m; ++k) {
      const double* __restrict__ rr = Ar + (size_t)k * m;
      const double* __restrict__ ri = Ai + (size_t)k * m;
      const double cr = vrp[k], ci = vip[k];
      for (int i = j1; i < m; ++i) {
        prp[i] += rr[i] * cr + ri[i] * ci;
        pip[i] += rr[i] * ci - ri[i] * cr;
      }
    }
    // p *= tau ; w = p - (tau / 2) (p^H v) v
    double hr = 0.0, hi = 0.0;      // p^H v
    for (int i = j1; i < m; ++i) {
      const double a = prp[i] * tr - pip[i] * ti, b = prp[i] * ti + pip[i] * tr;
      prp[i] = a;
      pip[i] = b;
      hr += a * vrp[i] + b * vip[i];
      hi += a * vip[i] - b * vrp[i];
    }
    const double gr = 0.5 * (tr * hr - ti * hi), gi = 0.5 * (tr * hi + ti * hr);
    for (int i = j1; i < m; ++i) {
      wrp[i] = prp[i] - (gr * vrp[i] - gi * vip[i]);
      wip[i] = pip[i] - (gr * vip[i] + gi * vrp[i]);
    }
    // A22 -= v w^H + w v^H  (full rows)
    for (int i = j1; i < m; ++i) {
      double* __restrict__ rr = Ar + (size_t)i * m;
      double* __restrict__ ri = Ai + (size_t)i * m;
      const double a = vrp[i], b = vip[i], c = wrp[i], dd = wip[i];
      for (int k = j1; k < m; ++k) {
        rr[k] -= (a * wrp[k] + b * wip[k]) + (c * vrp[k] + dd * vip[k]);
        ri[k] -= (b * wrp[k] - a * wip[k]) + (dd * vrp[k] - c * vip[k]);
      }
    }
  }
  d[m - 1] = Ar[(size_t)(m - 1) * m + (m - 1)];
}

static void tridiag_generic(double* Ar, double* Ai, int m, double* d, double* e, double* Vr, double* Vi, double* taur,
                            double* taui) {
  tridiag_core<0>(Ar, Ai, m, d, e, Vr, Vi, taur, taui);
}
__attribute__((target("avx2,fma"))) static void tridiag_avx2(double* Ar, double* Ai, int m, double* d, double* e, double* Vr,
                                                             double* Vi, double* taur, double* taui) {
  tridiag_core<1>(Ar, Ai, m, d, e, Vr, Vi, taur, taui);
}

// eigenvalues of the real symmetric tridiagonal (d[m], e[m - 1]) by implicit QL, ascending on return
static int tridiag_eigvals(std::vector<double> d, std::vector<double> e, std::vector<double>& w) {
  const int m = (int)d.size();
  e.resize(m, 0.0);
  e[m - 1] = 0.0;
  for (int l = 0; l < m; ++l) {
    int iter = 0, mm;
    do {
      for (mm = l; mm < m - 1; ++mm) {
        const double dd = std::fabs(d[mm]) + std::fabs(d[mm + 1]);
        if (std::fabs(e[mm]) <= 2.220446049250313e-16 * dd) break;
      }
      if (mm != l) {
        if (iter++ == 300) return -1;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[mm] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = mm - 1; i >= l; --i) {
          double f = s * e[i];
          const double b = c * e[i];
          e[i + 1] = (r = std::hypot(f, g));
          if (r == 0.0) {
            d[i + 1] -= p;
            e[mm] = 0.0;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          d[i + 1] = g + (p = s * r);
          g = c * r - b;
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[mm] = 0.0;
      }
    } while (mm != l);
  }
  std::sort(d.begin(), d.end());
  w = d;
  return 0;
}

// x <- (T - lam I)^-1 x for the tridiagonal (d, e): LU with partial pivoting (two superdiagonals of fill)
struct TriLU {
  std::vector<double> u0, u1, u2, l;
  std::vector<char> sw;
  void factor(const std::vector<double>& d, const std::vector<double>& e, double lam, double tiny) {
    const int m = (int)d.size();
    u0.assign(m, 0.0);
    u1.assign(m, 0.0);
    u2.assign(m, 0.0);
    l.assign(m, 0.0);
    sw.assign(m, 0);
    double p = d[0] - lam, q = m > 1 ? e[0] : 0.0, r = 0.0;
    for (int i = 0; i + 1 < m; ++i) {
      const double c = e[i], a1 = d[i + 1] - lam, b1 = i + 2 < m ? e[i + 1] : 0.0;
      if (std::fabs(p) >= std::fabs(c)) {
        if (std::fabs(p) < tiny) p = p < 0.0 ? -tiny : tiny;
        const double mult = c / p;
        u0[i] = p;
        u1[i] = q;
        u2[i] = r;
        l[i] = mult;
        p = a1 - mult * q;
        q = b1 - mult * r;
        r = 0.0;
      } else {
        const double mult = p / c;
        u0[i] = c;
        u1[i] = a1;
        u2[i] = b1;
        l[i] = mult;
        sw[i] = 1;
        p = q - mult * a1;
        q = r - mult * b1;
        r = 0.0;
      }
    }
    if (std::fabs(p) < tiny) p = p < 0.0 ? -tiny : tiny;
    u0[m - 1] = p;
  }
  void solve(std::vector<double>& x) const {
    const int m = (int)x.size();
    for (int i = 0; i + 1 < m; ++i) {
      if (sw[i]) std::swap(x[i], x[i + 1]);
      x[i + 1] -= l[i] * x[i];
    }
    x[m - 1] /= u0[m - 1];
    if (m > 1) x[m - 2] = (x[m - 2] - u1[m - 2] * x[m - 1]) / u0[m - 2];
    for (int i = m - 3; i >= 0; --i) x[i] = (x[i] - u1[i] * x[i + 1] - u2[i] * x[i + 2]) / u0[i];
  }
};

// Leading `nev` eigenpairs of the complex Hermitian matrix (Hr + i Hi), row-major m x m (symmetrised on entry).
// w[nev] descending; Xr / Xi: m x nev row-major, orthonormal columns.  0 on success.
static int zheigh_top(const double* Hr, const double* Hi, int m, int nev, double* w, double* Xr, double* Xi) {
  if (m <= 0 || nev <= 0 || nev > m) return -1;
  std::vector<double> Ar((size_t)m * m), Ai((size_t)m * m);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) {
      Ar[(size_t)i * m + j] = 0.5 * (Hr[(size_t)i * m + j] + Hr[(size_t)j * m + i]);
      Ai[(size_t)i * m + j] = 0.5 * (Hi[(size_t)i * m + j] - Hi[(size_t)j * m + i]);
    }
  std::vector<double> d(m), e(std::max(m - 1, 1), 0.0), Vr((size_t)m * m, 0.0), Vi((size_t)m * m, 0.0), taur(m, 0.0), taui(m, 0.0);
  if (m > 1) {
    static const bool fast = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    if (fast) tridiag_avx2(Ar.data(), Ai.data(), m, d.data(), e.data(), Vr.data(), Vi.data(), taur.data(), taui.data());
    else tridiag_generic(Ar.data(), Ai.data(), m, d.data(), e.data(), Vr.data(), Vi.data(), taur.data(), taui.data());
  } else {
    d[0] = Ar[0];
  }
  e.resize(m > 1 ? m - 1 : 0);
  std::vector<double> ev;
  if (tridiag_eigvals(d, e, ev) != 0) return -2;
  double tnorm = 0.0;
  for (int i = 0; i < m; ++i) tnorm = std::max(tnorm, std::fabs(d[i]) + (i > 0 ? std::fabs(e[i - 1]) : 0.0) + (i + 1 < m ? std::fabs(e[i]) : 0.0));
  if (!(tnorm > 0.0)) tnorm = 1.0;
  const double eps = 2.220446049250313e-16, sep = 10.0 * eps * tnorm, tiny = eps * tnorm;
  std::vector<std::vector<double>> X(nev, std::vector<double>(m));
  std::vector<double> lam(nev);
  TriLU lu;
  uint64_t rs = 0x9E3779B97F4A7C15ull;
  for (int j = 0; j < nev; ++j) {
    double lj = ev[m - 1 - j];
    w[j] = lj;
    if (j > 0 && lam[j - 1] - lj < sep) lj = lam[j - 1] - sep;      // separate (numerically) multiple eigenvalues
    lam[j] = lj;
    lu.factor(d, e, lj, tiny);
    std::vector<double>& x = X[j];
    for (int i = 0; i < m; ++i) {
      rs = rs * 6364136223846793005ull + 1442695040888963407ull;
      x[i] = (double)((rs >> 11) & 0xFFFFF) / 1048576.0 - 0.5;
    }
    bool ok = false;
    for (int it = 0; it < 6 && !ok; ++it) {
      lu.solve(x);
      double big = 0.0;
      for (int i = 0; i < m; ++i) big = std::max(big, std::fabs(x[i]));
      if (!(big > 0.0) || !std::isfinite(big)) return -3;
      for (int i = 0; i < m; ++i) x[i] /= big;
      for (int pass = 0; pass < 2; ++pass)
        for (int g = 0; g < j; ++g) {
          double dot = 0.0;
          for (int i = 0; i < m; ++i) dot += X[g][i] * x[i];
          for (int i = 0; i < m; ++i) x[i] -= dot * X[g][i];
        }
      double nrm = 0.0;
      for (int i = 0; i < m; ++i) nrm += x[i] * x[i];
      nrm = std::sqrt(nrm);
      if (!(nrm > 0.0)) return -3;
      for (int i = 0; i < m; ++i) x[i] /= nrm;
      if (it >= 1) {     // residual of the pair against the unperturbed eigenvalue
        double r2 = 0.0;
        for (int i = 0; i < m; ++i) {
          double t = (d[i] - w[j]) * x[i];
          if (i > 0) t += e[i - 1] * x[i - 1];
          if (i + 1 < m) t += e[i] * x[i + 1];
          r2 += t * t;
        }
        ok = std::sqrt(r2) <= 1e-9 * tnorm;
      }
    }
    if (!ok) return -4;
  }
  // x = H_0 H_1 ... H_{m-2} x_T
  for (int j = 0; j < nev; ++j) {
    std::vector<double> xr(X[j]), xi(m, 0.0);
    for (int r = m - 2; r >= 0; --r) {
      const double tr = taur[r], ti = taui[r];
      if (tr == 0.0 && ti == 0.0) continue;
      const double* vr = &Vr[(size_t)r * m];
      const double* vi = &Vi[(size_t)r * m];
      double sr = 0.0, si = 0.0;      // v^H x
      for (int i = r + 1; i < m; ++i) {
        sr += vr[i] * xr[i] + vi[i] * xi[i];
        si += vr[i] * xi[i] - vi[i] * xr[i];
      }
      const double cr = tr * sr - ti * si, ci = tr * si + ti * sr;
      for (int i = r + 1; i < m; ++i) {
        xr[i] -= cr * vr[i] - ci * vi[i];
        xi[i] -= cr * vi[i] + ci * vr[i];
      }
    }
    for (int i = 0; i < m; ++i) {
      Xr[(size_t)i * nev + j] = xr[i];
      Xi[(size_t)i * nev + j] = xi[i];
    }
  }
  return 0;
}

}  // namespace hosteig
