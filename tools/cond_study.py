"""Accuracy of the engine as a function of the spectrum's peakedness: GPU singular values (k = 40 modes, 36 of them
in the unconverged noise bulk) against the float64 oracle for growing leading-mode amplitudes.

Round-2 finding (replaces the round-1 reading of this sweep): without re-normalising the tall panel inside the
iteration the error follows eps_f32 * sigma_1 / sigma_k; with it (EOFX_FORCE_ORTH_TALL=1, or the adaptive rule the
drivers now apply: eofx_orth_tall_rule + eofx_peaked_spectrum) the default passes stay within 1e-5 up to
sigma_1 / sigma_k = 4e4.  Run with EOFX_FORCE_ORTH_TALL=0 / 1 / unset to see the three curves; a test-side study,
like the fuzz_* sweeps it uses the oracle as the checker."""
import sys, os, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from oracle import eof_oracle as orc
    from xeofs_amd import engine
    ctx = engine.Context(0)
    out = []
    for peak in (1.0, 3.0, 10.0, 30.0, 100.0, 300.0):
        rng = np.random.default_rng(5)
        n, p, k = 200, 70000, 40
        amp = peak * 0.6 ** np.arange(4)
        X = ((rng.standard_normal((n, 4)) * amp) @ rng.standard_normal((4, p)) + rng.standard_normal((n, p))).astype(np.float32)
        X -= X.mean(0)
        Uo, so, Vo = orc.decomposer_fit(X.astype(np.float64), k, random_state=2, solver="randomized")
        mat = engine.from_dense(ctx, X)
        U, s, V = engine.rsvd(ctx, mat, k, random_state=2)
        mat.free()
        out.append((peak, float(so[0] / so[-1]), float(np.max(np.abs(s - so) / so))))
    print("RESULT " + json.dumps(out))
else:
    r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    print("peak amplitude, sigma_1 / sigma_k, max rel error of s vs float64 oracle:")
    print([(pk, round(c, 1), f"{e:.1e}") for pk, c, e in json.loads(line[0][7:])] if line else r.stderr[-500:])
