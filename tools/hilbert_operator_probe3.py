"""Full-size check with an exactly known answer: X = T W S^T (rank r, factors known), so the analytic signal is
Z = H(T) W S^T and its singular values follow from two small QR factorisations in float64.
usage: python tools/hilbert_operator_probe3.py [n] [p] [r]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import eof_oracle as orc
from xeofs_amd import engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 720 * 1440
r = int(sys.argv[3]) if len(sys.argv) > 3 else 60
k = 20
rng = np.random.default_rng(7)
T = np.empty((n, r))
e = rng.standard_normal((n, r))
T[0] = e[0]
for i in range(1, n):
    T[i] = 0.8 * T[i - 1] + 0.6 * e[i]
T[:, :4] += np.cumsum(rng.standard_normal((n, 4)), axis=0) * 0.05          # a few drifting series
W = 0.93 ** np.arange(r) * 3.0
S = rng.standard_normal((p, r))
dev = "cuda"
Td = torch.as_tensor((T * W).astype(np.float32), device=dev)
Sd = torch.as_tensor(S.astype(np.float32), device=dev)
X = torch.empty((n, p), dtype=torch.float32, device=dev)
for c0 in range(0, p, 65536):
    X[:, c0:c0 + 65536] = Td @ Sd[c0:c0 + 65536].T
# exact: the factors as the engine sees them (float32-rounded), centred over the samples
T32 = Td.cpu().numpy().astype(np.float64)
S32 = Sd.cpu().numpy().astype(np.float64)
Tc = T32 - T32.mean(0)
ZT = orc.hilbert_transform(Tc, padding="exp", decay_factor=0.2)
R1 = np.linalg.qr(ZT, mode="r")
R2 = np.linalg.qr(S32, mode="r")
se = np.linalg.svd(R1 @ R2.T, compute_uv=False)[:k]
ctx = engine.default_context(0)
A, _ = engine.preprocess(ctx, X, want_stats=False, in_place=True, for_hilbert=True)
# the REAL decomposition of the same field against its exact values (the same streaming kernels, no Hilbert stage)
sr = np.linalg.svd(np.linalg.qr(Tc, mode="r") @ R2.T, compute_uv=False)[:k]
for prec in ("f16x3", "f32"):
    ctx.set_precision(prec, prec)
    _, s0_, _ = engine.rsvd(ctx, A, k, random_state=5, device_out=True)
    print(f"real rSVD {prec}: max |s - exact| / s0 {np.abs(s0_ - sr).max() / sr[0]:.2e} per mode {(np.abs(s0_ - sr) / sr).max():.2e}  s0 {s0_[0]:.3f} exact {sr[0]:.3f}", flush=True)
ctx.set_precision("f16x3", "f16x3")
sq = engine.hilbert_sumsq(ctx, A, "exp", 0.2)
tv_im_exact = (np.abs((ZT.imag @ R2.T)) ** 2).sum()
print(f"sum of squares of Im: engine {sq:.6e} exact {tv_im_exact:.6e} rel {abs(sq - tv_im_exact) / tv_im_exact:.2e}")
for rule in ("auto", "converge"):
    _, s1, _ = engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, random_state=5, n_iter=rule, device_out=True)
    print(f"operator {rule}: max |s - exact| / s0 {np.abs(s1 - se).max() / se[0]:.2e} per mode {(np.abs(s1 - se) / se).max():.2e}  s0 {s1[0]:.3f} exact {se[0]:.3f}", flush=True)
B, _ = engine.hilbert(ctx, A, "exp", 0.2)
for rule in ("auto", "converge"):
    _, s2, _ = engine.rsvd_c64(ctx, A, B, k, random_state=5, n_iter=rule, device_out=True)
    print(f"two-part {rule}: max |s - exact| / s0 {np.abs(s2 - se).max() / se[0]:.2e} per mode {(np.abs(s2 - se) / se).max():.2e}  s0 {s2[0]:.3f} exact {se[0]:.3f}", flush=True)
