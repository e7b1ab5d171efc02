"""LDS / register canaries beside the streaming kernels: workgroups that fill their LDS allocation and 64 VGPRs per lane with a
pattern, sleep, and check it (tools/probes/lds_canary.hip), on their own stream while another context runs its passes."""
import sys, os, threading, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xeofs_amd import engine
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "liblds_canary.so"))
lib.lds_canary_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
rng = np.random.default_rng(0)
X = (rng.standard_normal((700, 5)) @ rng.standard_normal((5, 2048)) + 1.0).astype(np.float32)
stop = False
res = {}
def victim(tid, lds_bytes):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        errs = torch.zeros(2, dtype=torch.int32, device="cuda")
        for rep in range(300):
            rc = lib.lds_canary_run(C.c_void_p(st.cuda_stream), C.c_void_p(errs.data_ptr()), 512, lds_bytes, 40)
            assert rc == 0, rc
        st.synchronize()
        res[(tid, lds_bytes)] = errs.cpu().tolist()
def load(tid):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx = engine.Context(0)
        kind = os.environ.get("LOAD", "inplace")
        keep = engine.preprocess(ctx, X, True, False, None, in_place=(kind != "copy"))[0]
        Zn = torch.randn(keep.n_pad, 64, device="cuda"); Zn[keep.n:] = 0
        Yp = torch.randn(keep.p_pad, 64, device="cuda"); Yp[keep.p:] = 0
        while not stop:
            if kind == "none":
                import time; time.sleep(0.01); continue
            for _ in range(10):
                engine.panel_tmul(ctx, keep, Zn, prec=("f32" if kind == "f32" else "f16x3"))
                engine.panel_mul(ctx, keep, Yp, prec=("f32" if kind == "f32" else "f16x3"))
ths = [threading.Thread(target=victim, args=(t, b)) for t, b in enumerate((16384, 65536, 131072))]
lds = [threading.Thread(target=load, args=(t,)) for t in range(2)]
for t in lds + ths: t.start()
for t in ths: t.join()
stop = True
for t in lds: t.join()
print("LOAD", os.environ.get("LOAD", "inplace"), "[LDS words changed, registers changed] per victim (lds bytes):", {k[1]: v for k, v in res.items()})
