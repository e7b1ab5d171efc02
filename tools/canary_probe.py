"""Stray device writes: surround the engine's allocations with canary buffers, run the decomposition, look for changed canaries."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
rng = np.random.default_rng(0)
fields = [(rng.standard_normal((n, 5)) @ rng.standard_normal((5, p)) + 0.2 * rng.standard_normal((n, p)) + 1.0).astype(np.float32)
          for n, p in ((400, 4096), (700, 2048), (300, 8192), (1000, 1000))]
PAT = 0x7FC0DEAD      # a NaN pattern no kernel produces
def canaries(k, mb):
    out = []
    for _ in range(k):
        t = torch.full((mb * 262144,), PAT, dtype=torch.int32, device="cuda")
        out.append(t)
    return out
c0 = canaries(24, 8)
ctx = engine.Context(0)
c1 = canaries(24, 8)
mats = []
for X in fields:
    mats.append(engine.preprocess(ctx, X, True, False, None, in_place=True)[0])
    c1 += canaries(4, 8)
which = os.environ.get("WHAT", "rsvd")
for rep in range(30):
    for m, X in zip(mats, fields):
        if which == "rsvd":
            engine.rsvd(ctx, m, 6, random_state=3)
        else:
            mm, st, U, s, V = engine.fit(ctx, X, 6, random_state=3); mm.free()
    c1 += canaries(2, 8) if rep < 8 else []
torch.cuda.synchronize()
hits = 0
for i, t in enumerate(c0 + c1):
    badm = t != PAT
    if bool(badm.any()):
        idx = torch.nonzero(badm).flatten()
        vals = t[idx[:8]].cpu().numpy()
        hits += 1
        print(f"canary {i} at 0x{t.data_ptr():x}: {int(badm.sum())} words changed, first offsets {idx[:6].tolist()} (bytes {[int(x) * 4 for x in idx[:3].tolist()]}), as float {vals.view(np.float32)[:6]}", flush=True)
print("canaries", len(c0 + c1), "hit", hits)
