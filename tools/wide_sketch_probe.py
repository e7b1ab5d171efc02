"""EOF with 55 .. 118 modes (128-column panels) at config-4 size: rSVD time with the X^T Z passes as two 64-column launches
(two reads of the field per pass: EOFX_ATB_WIDE_MIN=256, the rule until round 3) and as one 128-column tile (the default now).  python tools/wide_sketch_probe.py"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np, torch
    from xeofs_amd import engine
    import bench
    n, nlat, nlon = 10000, 720, 1440
    ctx = engine.Context(0)
    X = bench.make_field(n, nlat, nlon, 0, nlat * nlon, torch.device("cuda:0"))
    mat, st = engine.preprocess(ctx, X, want_stats=False, in_place=True)
    for k in (100, 150):
        engine.rsvd(ctx, mat, k, random_state=5, device_out=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        U, s, V = engine.rsvd(ctx, mat, k, random_state=5, device_out=True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"EOFX_ATB_WIDE_MIN={os.environ.get('EOFX_ATB_WIDE_MIN', '-')} EOFX_NO_WIDE_XT={os.environ.get('EOFX_NO_WIDE_XT', '-')} sample layout built: {mat.has_sample_layout()}: k={k} rsvd {1e3 * (t1 - t0):.1f} ms  s[:2]={s[:2]} s[-1]={s[-1]:.4f}", flush=True)
else:
    for v, noxt in (("256", "1"), (None, "1"), (None, None)):
        env = dict(os.environ)
        if v:
            env["EOFX_ATB_WIDE_MIN"] = v
        if noxt:
            env["EOFX_NO_WIDE_XT"] = noxt      # keep the X Y passes on axb_f16 (64 columns per launch) as until round 3
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print("\n".join(l for l in out.stdout.splitlines() if "rsvd" in l) or out.stderr[-400:], flush=True)
