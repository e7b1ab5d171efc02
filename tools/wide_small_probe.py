"""The float64 small-side kernels of the PCA route at its sizes: Gram matrix of a (rows x 1536) panel (gram_mfma_kernel) and the
product of a (rows x 1536) panel with a 1536 x 1536 float64 matrix (panel_matmul_kernel, windowed form).
usage: [EOFX_GRAM_PARTS=k] [EOFX_PMM_KW=128|256] python tools/wide_small_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xeofs_amd import engine

ctx = engine.Context(0)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    return min(ts)
for rows, L in ((129600 + 448, 1536), (5120, 1536), (1036800 + 256, 64), (5120, 512)):
    rows = (rows + 511) // 512 * 512
    P = torch.randn((rows, L), generator=g, device=dev, dtype=torch.float32)
    t = timeit(lambda: engine.panel_gram(ctx, P))
    fl = rows * L * L          # 2 rows L^2 / 2 (symmetric)
    print(f"panel_gram  rows {rows} L {L}: {t:.3f} ms = {fl / t / 1e9:.1f} TFLOP/s float64 (of 78.6)  parts={os.environ.get('EOFX_GRAM_PARTS', 'default')}", flush=True)
    if rows <= 200000 or L == 64:
        M = torch.randn((L, L), generator=g, device=dev, dtype=torch.float64)
        t = timeit(lambda: engine.panel_matmul(ctx, P, M))
        print(f"panel_matmul rows {rows} L {L}: {t:.3f} ms = {2.0 * rows * L * L / t / 1e9:.1f} TFLOP/s float64  KW={os.environ.get('EOFX_PMM_KW', 'default')}", flush=True)
    del P
