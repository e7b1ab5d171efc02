import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from xeofs_amd import engine
for shape in [(10000, 60), (5000, 60), (129600, 30)]:
    ref = np.random.RandomState(5).normal(size=shape).astype(np.float32)
    for T in ("4", "8", ""):
        if T: os.environ["EOFX_SKETCH_THREADS"] = T
        else: os.environ.pop("EOFX_SKETCH_THREADS", None)
        for _ in range(3): om = engine.sketch_matrix(shape[0], shape[1], 5)
        t = time.perf_counter()
        for _ in range(20): om = engine.sketch_matrix(shape[0], shape[1], 5)
        dt = (time.perf_counter() - t) / 20
        print(f"pin={os.environ.get('EOFX_SKETCH_PIN','0')} {shape} threads={T or 'default'}: {dt * 1e3:.3f} ms  ok {np.array_equal(om, ref)}", flush=True)
