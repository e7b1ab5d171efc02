"""The feature-side panel kernels of a config-4 fit on their own (1 036 800 x 64 float32 panel): float64 Gram
(gram_mfma_kernel + f64_reduce_kernel) for several numbers of row partials, panel_matmul, Cholesky factor inverse.
python tools/small_kernel_probe.py"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:      # child: one setting of EOFX_GRAM_PARTS
    import numpy as np, torch
    from xeofs_amd import engine
    ctx = engine.Context(0)
    rows, L = 1036800, 64
    P = torch.randn((rows, L), device="cuda", dtype=torch.float32)
    M = torch.randn((L, L), device="cuda", dtype=torch.float64)

    def t(fn, reps=30):
        fn(); torch.cuda.synchronize(); a = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - a) / reps

    g = t(lambda: engine.panel_gram(ctx, P))
    G = engine.panel_gram(ctx, P)
    ref = (P[:200000].double().T @ P[:200000].double())
    G2 = engine.panel_gram(ctx, P[:200000].contiguous())
    err = float((G2 - ref).abs().max() / ref.abs().max())
    line = f"gram parts={os.environ.get('EOFX_GRAM_PARTS', 'default')}: {g:.1f} us  (check {err:.1e})"
    if sys.argv[1] == "all":
        mm = t(lambda: engine.panel_matmul(ctx, P, M))
        Gs = engine.panel_gram(ctx, P[:10240].contiguous())
        ri = t(lambda: engine.panel_rinv(ctx, Gs, L))
        Pn = P[:10240].contiguous()
        gn = t(lambda: engine.panel_gram(ctx, Pn))
        mn = t(lambda: engine.panel_matmul(ctx, Pn, M))
        line += f"; panel_matmul {mm:.1f} us; n-side (10240 rows): gram {gn:.1f} us, rinv {ri:.1f} us, matmul {mn:.1f} us  [wall per call incl. launch]"
    print(line)
else:
    for parts in ("default", "256", "1024", "2048", "4096"):
        env = dict(os.environ)
        if parts != "default":
            env["EOFX_GRAM_PARTS"] = parts
        out = subprocess.run([sys.executable, __file__, "all" if parts == "default" else "one"], env=env, capture_output=True, text=True)
        print((out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
