"""Wall time of the native sketch generator (eofx_sketch_gaussian_f32: numpy's legacy RandomState stream, bit for bit) for
the sketches of configs 4 / 2 / 3, by worker-thread count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xeofs_amd import engine

for shape in [(10000, 60), (5000, 60), (129600, 30)]:
    ref = np.random.RandomState(5).normal(size=shape).astype(np.float32)
    for T in ("1", "4", "8", "16", ""):
        if T:
            os.environ["EOFX_SKETCH_THREADS"] = T
        else:
            os.environ.pop("EOFX_SKETCH_THREADS", None)
        for _ in range(3):
            om = engine.sketch_matrix(shape[0], shape[1], 5)
        t = time.perf_counter()
        for _ in range(20):
            om = engine.sketch_matrix(shape[0], shape[1], 5)
        dt = (time.perf_counter() - t) / 20
        print(f"{shape} threads={T or 'default'}: {dt * 1e3:.3f} ms  bit-identical to numpy: {np.array_equal(om, ref)}", flush=True)
