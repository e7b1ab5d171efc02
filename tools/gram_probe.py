import sys, time; sys.path.insert(0, "/root/repo")
import torch, numpy as np
from xeofs_amd import engine
ctx = engine.default_context(0)
for rows, L in ((1036800, 64), (10240, 64), (1036800, 32), (1036800, 128)):
    P = torch.randn((rows, L), device="cuda")
    G = engine.panel_gram(ctx, P); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): G = engine.panel_gram(ctx, P)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    ref = (P.double().T @ P.double())
    print(rows, L, f"{dt*1e6:.0f} us", "max rel err", float((G - ref).abs().max() / ref.abs().max()), "sym", bool(torch.equal(G, G.T)))
    M = torch.eye(L, dtype=torch.float64, device="cuda")
    O = engine.panel_matmul(ctx, P, M); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): O = engine.panel_matmul(ctx, P, M)
    torch.cuda.synchronize(); print("   matmul", f"{(time.perf_counter()-t)/10*1e6:.0f} us")
