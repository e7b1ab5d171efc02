"""Can rocSOLVER's syevd (torch.linalg.eigh, order ~1500, float64) be captured in a HIP graph and replayed?  The call is ~9000
launch-bound kernels (36 ms); a replayed graph would shed the per-launch host cost.  Prints direct vs replay time and whether the
replayed results equal the direct ones bit for bit."""
import sys
import time

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1510
dev = "cuda"
g = torch.Generator(device=dev)
g.manual_seed(1)


def spd(seed):
    g.manual_seed(seed)
    A = torch.randn(n, 3 * n, generator=g, device=dev, dtype=torch.float64)
    return (A @ A.T) / (3 * n)


M1, M2 = spd(1), spd(2)


def timed(fn, reps=5):
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0))
    return best, out


t_direct, (w1, V1) = timed(lambda: torch.linalg.eigh(M1))
_, (w2, V2) = timed(lambda: torch.linalg.eigh(M2), 1)
print(f"order {n}: direct eigh {t_direct:.2f} ms")
static = M1.clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        torch.linalg.eigh(static)
torch.cuda.current_stream().wait_stream(side)
try:
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ws, Vs = torch.linalg.eigh(static)
except Exception as e:      # noqa: BLE001
    print("capture failed:", repr(e)[:300])
    sys.exit(0)


def replay(M):
    static.copy_(M)
    graph.replay()
    return ws.clone(), Vs.clone()


t_rep, (wr1, Vr1) = timed(lambda: replay(M1))
_, (wr2, Vr2) = timed(lambda: replay(M2), 1)
print(f"order {n}: graph replay {t_rep:.2f} ms; equal to direct: values {torch.equal(wr1, w1)} / {torch.equal(wr2, w2)}, "
      f"vectors {torch.equal(Vr1, V1)} / {torch.equal(Vr2, V2)}; max |dw| {float((wr2 - w2).abs().max()):.2e}")
