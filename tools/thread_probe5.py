"""is `torch.linalg.eigh` (rocSOLVER, order 1500 -- the library call `cross/cpcca.py::_run_two` used to run on a
side stream) a victim of the two-stream effect of DESIGN section 10?  Victim threads: eigh of a fixed symmetric matrix (and, as the
known-bad control, torch.fft.fft) on their own streams, compared bit for bit with their first result; load threads: another
context's streaming kernels (LOAD=inplace: split-fp16 in-place X^T Z; LOAD=gram: the fp16 sample-space Gram matrix; LOAD=none)."""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from xeofs_amd import engine

rng = np.random.default_rng(0)
X = (rng.standard_normal((1500, 5)) @ rng.standard_normal((5, 8192)) + rng.standard_normal((1500, 8192)) + 1.0).astype(np.float32)
REPS = int(os.environ.get("REPS", "60"))
bad = {"eigh_values": 0, "eigh_vectors": 0, "fft": 0}
stop = False


def victim(tid):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.Generator(device="cuda")
        g.manual_seed(7 + tid)
        A = torch.randn(1500, 1500, generator=g, device="cuda", dtype=torch.float64)
        M = A @ A.T / 1500.0
        x = torch.randn(2048, 1024, generator=g, device="cuda")
        w0, V0 = torch.linalg.eigh(M)
        f0 = torch.fft.fft(x, dim=1)
        st.synchronize()
        for _ in range(REPS):
            w, V = torch.linalg.eigh(M)
            f = torch.fft.fft(x, dim=1)
            st.synchronize()
            bad["eigh_values"] += int(not torch.equal(w, w0))
            bad["eigh_vectors"] += int(not torch.equal(V, V0))
            bad["fft"] += int(not torch.equal(f, f0))


def load(tid):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx = engine.Context(0)
        keep = engine.preprocess(ctx, X, True, False, None, in_place=True)[0]
        Zn = torch.randn(keep.n_pad, 64, device="cuda")
        Zn[keep.n:] = 0
        kind = os.environ.get("LOAD", "inplace")
        while not stop:
            if kind == "none":
                import time

                time.sleep(0.01)
                continue
            for _ in range(10):
                if kind == "gram":
                    keep.gram(0)
                else:
                    engine.panel_tmul(ctx, keep, Zn, prec="f16x3")
            st.synchronize()


ths = [threading.Thread(target=victim, args=(t,)) for t in range(2)]
lds = [threading.Thread(target=load, args=(t,)) for t in range(2)]
for t in lds + ths:
    t.start()
for t in ths:
    t.join()
stop = True
for t in lds:
    t.join()
print(f"LOAD={os.environ.get('LOAD', 'inplace')}: 2 x {REPS} repetitions; results that differ from the first:", bad)
