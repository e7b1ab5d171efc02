"""In-place layout against the two-layout matrix: the sample-side product X Y (axb_f16_kernel vs atb_f16_kernel on the
sample-contiguous copy), the on-demand sample-contiguous layout (bitwise the copy apply_kernel writes), and a whole rSVD.
python tools/inplace_check.py [n p]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine

shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(100, 5000), (1000, 40000), (333, 77780), (2000, 3000), (64, 1024)]
ctx = engine.Context(0)
for n, p in shapes:
    g = torch.Generator(device="cuda").manual_seed(n + p)
    X = torch.randn((n, p), device="cuda", generator=g) * (1 + 3 * torch.rand(p, device="cuda", generator=g)) + 280.0
    w = (0.2 + torch.rand(p, device="cuda", generator=g)).cpu().numpy().astype(np.float64)
    for std in (False, True):
        m0, s0 = engine.preprocess(ctx, X, True, std, w)
        m2, s2 = engine.preprocess(ctx, X, True, std, w, in_place=True)
        assert not m2.has_sample_layout()
        for L in (64, 32, 96):
            Y = torch.randn((m0.p_pad, L), device="cuda", generator=g); Y[p:] = 0
            W0 = engine.panel_mul(ctx, m0, Y, prec="f16x3")
            W2 = engine.panel_mul(ctx, m2, Y, prec="f16x3")
            W64 = (torch.as_tensor(m0.download(), device="cuda").double() @ Y[:p].double())
            e0 = float((W0[:n].double() - W64).norm() / W64.norm()); e2 = float((W2[:n].double() - W64).norm() / W64.norm())
            assert float(W2[n:].abs().max()) == 0.0 if m0.n_pad > n else True
            print(f"n={n} p={p} std={std} L={L}: rel err vs float64 product: copy {e0:.2e}  in place {e2:.2e}   max|diff| {float((W0-W2).abs().max()):.2e} of {float(W0.abs().max()):.2e}")
            assert e2 < 3 * max(e0, 1e-7)
        assert not m2.has_sample_layout()
        U0, sv0, V0 = engine.rsvd(ctx, m0, 10, random_state=3)
        U2, sv2, V2 = engine.rsvd(ctx, m2, 10, random_state=3)
        print(f"    rsvd s rel diff {np.max(np.abs(sv0 - sv2) / sv0):.2e}   |cos| min {np.min(np.abs(np.sum(V0 * V2, 0))):.7f}")
        d0 = m0.download(); d2 = m2.download()          # builds the layouts of the in-place matrix on demand
        assert m2.has_sample_layout() and np.array_equal(d0, d2), "on-demand layout differs from the written one"
        m0.free(); m2.free()
print("ok")
