"""parity evidence for the complex branch (R9) on a field WITH THE BENCH FIELD'S SPECTRUM at the largest
size the host holds -- modes 7..20 of BASELINE config 5 sit inside the noise bulk, and the gate sample of bench.py (2000 x 3200)
does not show how the timed rule behaves there.

For the bench's synthetic field at n x (nlat x nlon) (default 8000 x (128 x 512) = 8000 x 65 536; complex128 on the host: 8.4 GB):

    exact       singular values of the analytic signal (oracle Hilbert transform, padding "exp"; float64 Hermitian Gram + LAPACK eigh)
    lobpcg      the REFERENCE'S solver on the same matrix: scipy.sparse.linalg.svds(solver="lobpcg")  (xeofs/linalg/decomposer.py:149-160)
    auto        the engine's timed rule, eofx_rsvd_hilbert_c64 with n_iter="auto" (7 products, 16 passes)
    converge    the engine's n_iter="converge" (residual <= 1e-5 per wanted Ritz pair, at most 20 products)

and prints the per-mode relative errors against `exact` with the verdict per mode: auto <= max(1e-5, lobpcg's own error).
`run()` is also what tests/test_gpu_complex.py calls at a smaller size."""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n=8000, nlat=128, nlon=512, k=20, seed=5, device="cuda:0", ctx=None, verbose=False, exact_from=None):
    """exact_from: a JSON of an earlier run on the SAME field (it is deterministic): the host legs (minutes) are taken from it"""
    import torch

    import bench
    from oracle import eof_oracle as orc          # checker only
    from xeofs_amd import engine

    ctx = ctx or engine.default_context(0)
    P = nlat * nlon
    X = bench.make_field(n, nlat, nlon, 0, P, torch.device(device))
    out = {"shape": [n, P], "k": k}
    A, _ = engine.preprocess(ctx, X, want_stats=False, in_place=True)
    engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, random_state=seed, n_iter=1)      # (builds the operator for this n: not timed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, s_auto, _ = engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, random_state=seed)
    torch.cuda.synchronize()
    out["auto_ms"] = 1e3 * (time.perf_counter() - t0)
    out["auto_products"] = engine.last_iterations(ctx)
    t0 = time.perf_counter()
    _, s_conv, _ = engine.rsvd_hilbert_c64(ctx, A, k, "exp", 0.2, random_state=seed, n_iter="converge")
    torch.cuda.synchronize()
    out["converge_ms"] = 1e3 * (time.perf_counter() - t0)
    out["converge_products"] = engine.last_iterations(ctx)
    A.free()
    if exact_from is not None:
        with open(exact_from) as f:
            prev = json.load(f)
        assert prev["shape"] == [n, P] and prev["k"] == k
        s_exact = np.asarray(prev["s_exact"])
        e = lambda s: np.abs(np.asarray(s, dtype=np.float64) - s_exact[:k]) / s_exact[:k]
        e_auto, e_conv, e_lob = e(s_auto), e(s_conv), np.asarray(prev["err_lobpcg"])
        out.update(host_hilbert_s=prev["host_hilbert_s"], host_exact_s=prev["host_exact_s"], host_lobpcg_s=prev["host_lobpcg_s"],
                   s_exact=[float(v) for v in s_exact], err_auto=[float(v) for v in e_auto], err_converge=[float(v) for v in e_conv],
                   err_lobpcg=[float(v) for v in e_lob], auto_ok=[bool(a <= max(1e-5, b)) for a, b in zip(e_auto, e_lob)],
                   converge_ok=[bool(a <= max(1e-5, b)) for a, b in zip(e_conv, e_lob)])
        return out
    x64 = X.cpu().numpy().astype(np.float64)
    del X
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    z = orc.hilbert_transform(x64 - x64.mean(axis=0), padding="exp", decay_factor=0.2)
    del x64
    out["host_hilbert_s"] = time.perf_counter() - t0
    if verbose:
        print(f"[r9] analytic signal on the host: {out['host_hilbert_s']:.1f} s", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    G = z @ z.conj().T                                     # n x n Hermitian Gram matrix, complex128
    w = np.linalg.eigvalsh(G)[::-1]
    del G
    s_exact = np.sqrt(np.maximum(w[:k + 12], 0.0))
    out["host_exact_s"] = time.perf_counter() - t0
    if verbose:
        print(f"[r9] exact values (Gram + eigvalsh): {out['host_exact_s']:.1f} s", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                    # (lobpcg reports the modes it did not converge)
        _, s_lob, _ = orc.complex_svds(z, k, random_state=seed)
    out["host_lobpcg_s"] = time.perf_counter() - t0
    del z
    e = lambda s: np.abs(np.asarray(s, dtype=np.float64) - s_exact[:k]) / s_exact[:k]
    e_auto, e_conv, e_lob = e(s_auto), e(s_conv), e(s_lob)
    out.update(s_exact=[float(v) for v in s_exact], err_auto=[float(v) for v in e_auto], err_converge=[float(v) for v in e_conv],
               err_lobpcg=[float(v) for v in e_lob],
               auto_ok=[bool(a <= max(1e-5, b)) for a, b in zip(e_auto, e_lob)],
               converge_ok=[bool(a <= max(1e-5, b)) for a, b in zip(e_conv, e_lob)])
    return out


def table(out):
    k = out["k"]
    se = out["s_exact"]
    lines = [f"# R9 evidence: bench field {out['shape'][0]} x {out['shape'][1]}, k = {k}; relative errors of the singular values vs the exact ones",
             f"# engine auto: {out['auto_products']} products, {out['auto_ms']:.1f} ms; converge: {out['converge_products']} products, "
             f"{out['converge_ms']:.1f} ms; host: Hilbert {out['host_hilbert_s']:.0f} s, exact {out['host_exact_s']:.0f} s, "
             f"scipy svds(lobpcg) {out['host_lobpcg_s']:.0f} s",
             "# mode  s_exact        gap_to_next  err_auto    err_converge  err_lobpcg(reference)  auto<=max(1e-5,lobpcg)  converge<=..."]
    for j in range(k):
        gap = (se[j] - se[j + 1]) / se[j]
        lines.append(f"{j + 1:5d}  {se[j]:13.6f}  {gap:10.3e}  {out['err_auto'][j]:10.3e}  {out['err_converge'][j]:10.3e}    "
                     f"{out['err_lobpcg'][j]:10.3e}             {'yes' if out['auto_ok'][j] else 'NO '}                     "
                     f"{'yes' if out['converge_ok'][j] else 'NO '}")
    lines.append(f"# auto: {sum(out['auto_ok'])} of {k} modes within max(1e-5, the reference solver's own error); converge: "
                 f"{sum(out['converge_ok'])} of {k}")
    return "\n".join(lines)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsamples", type=int, default=8000)
    ap.add_argument("--nlat", type=int, default=128)
    ap.add_argument("--nlon", type=int, default=512)
    ap.add_argument("--modes", type=int, default=20)
    ap.add_argument("--json", default=None)
    ap.add_argument("--exact-from", default=None, help="JSON of an earlier run on the same field: skip the host legs")
    a = ap.parse_args()
    res = run(a.nsamples, a.nlat, a.nlon, a.modes, verbose=True, exact_from=a.exact_from)
    print(table(res))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f)
