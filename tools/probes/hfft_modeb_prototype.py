"""numpy prototype of the one-feature mode of the fused Hilbert kernel: the real-input split, the filter and the merge back
as ONE step on the pair (Z[k], Z[M-k]) of the half-length transform (eofx_hfft.hpp, MODE 1).  Prints the errors against
the full-length transform and a direct convolution (all ~1e-15)."""
import numpy as np
rng = np.random.default_rng(0)
n = 700; P = 2048; M = P // 2
# real odd kernel c on lags (-n, n)
lags = np.arange(-(n-1), n)
kap = np.where(lags % 2 != 0, 1.0 / np.tan(np.pi * lags / (3*n)) * 2 / (3*n), 0.0)
c = np.zeros(P); c[lags % P] = kap / P
h = np.fft.fft(c).imag          # FFT(c) = i h
assert np.abs(np.fft.fft(c).real).max() < 1e-12
y = rng.standard_normal(n)
ypad = np.zeros(P); ypad[:n] = y
ref = np.fft.ifft(np.fft.fft(ypad) * 1j * h).real * P      # unnormalised inverse with 1/P in c
direct = np.array([sum(kap[(i - s) + n - 1] * y[s] for s in range(n)) for i in range(n)])
print("ref vs direct", np.abs(ref[:n] - direct).max())
# mode B: half-size complex transform
z = ypad[0::2] + 1j * ypad[1::2]
Z = np.fft.fft(z)                       # length M
k = np.arange(M)
kp = (M - k) % M
w = np.exp(-2j * np.pi * k / P)         # omega^k
A = Z; Bc = np.conj(Z[kp])
Y = 0.5 * (A + Bc) - 0.5j * w * (A - Bc)               # Y[k], k in [0, M)
F = 1j * h[:M] * Y
# F[M-k] for k in [1, M-1]; F[M] = 0 (h[M] = 0)
Fp = np.where(k == 0, 0.0, F[kp])                      # F[M - k]; at k = 0 that is F[M] = 0
W = (F + np.conj(Fp)) + 1j * np.conj(w) * (F - np.conj(Fp))
wout = np.fft.ifft(W) * M                               # unnormalised inverse of length M
o = np.empty(P); o[0::2] = wout.real; o[1::2] = wout.imag
print("mode B vs ref", np.abs(o[:n] - ref[:n]).max(), np.abs(ref[:n]).max())
# per-pair closed form: W[k] = alpha_k Z[k] + beta_k conj(Z[M-k]) ?
hk = h[:M]; hkp = np.where(k == 0, 0.0, h[(M - k) % P])   # h[M-k]; h[M] = 0
hm = hk - hkp; hp2 = 0.5 * (hk + hkp)
B = Z[kp]
W2 = 1j * hm * A + hp2 * (w * (A - np.conj(B)) - np.conj(w) * (A + np.conj(B)))
print("simplified W vs W", np.abs(W2 - W).max() / np.abs(W).max())
print("h[M] =", h[M], " hm[0], hp2[0] =", hm[0], hp2[0])
