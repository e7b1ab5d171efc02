"""Is the interference specific to the engine's transform kernel?  Victims that are not the engine's: torch.fft.fft and an elementwise
kernel on their own streams, beside the same load (the in-place streaming kernel of another context)."""
import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
rng = np.random.default_rng(0)
X = (rng.standard_normal((700, 5)) @ rng.standard_normal((5, 2048)) + 1.0).astype(np.float32)
bad = {"fft": 0, "ew": 0, "sort": 0}
stop = False
def victim(tid):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        x = torch.randn(2048, 1024, device="cuda")
        ref_f = torch.fft.fft(x, dim=1); ref_e = torch.sin(x) * 2 + x; ref_s = torch.sort(x, dim=1).values
        st.synchronize()
        for rep in range(300):
            f = torch.fft.fft(x, dim=1); e = torch.sin(x) * 2 + x; s = torch.sort(x, dim=1).values
            st.synchronize()
            bad["fft"] += int(not torch.equal(f, ref_f)); bad["ew"] += int(not torch.equal(e, ref_e)); bad["sort"] += int(not torch.equal(s, ref_s))
def load(tid):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx = engine.Context(0)
        keep = engine.preprocess(ctx, X, True, False, None, in_place=True)[0]
        Zn = torch.randn(keep.n_pad, 64, device="cuda"); Zn[keep.n:] = 0
        kind = os.environ.get("LOAD", "inplace")
        keepc = engine.preprocess(ctx, X, True, False, None)[0] if kind == "copy" else None
        while not stop:
            if kind == "none":
                import time; time.sleep(0.01); continue
            for _ in range(10):
                engine.panel_tmul(ctx, keepc if kind == "copy" else keep, Zn, prec=("f32" if kind == "f32" else "f16x3"))
ths = [threading.Thread(target=victim, args=(t,)) for t in range(2)]
lds = [threading.Thread(target=load, args=(t,)) for t in range(2)]
for t in lds + ths: t.start()
for t in ths: t.join()
stop = True
for t in lds: t.join()
print("2 x 300 repetitions beside the in-place kernel; results that differ from the first:", bad)
