import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xeofs_amd import engine
for shape in [(10000, 60), (129600, 30)]:
    a = engine.sketch_matrix(*shape, 5); b = np.random.RandomState(5).normal(size=shape).astype(np.float32)
    assert np.array_equal(a, b)
    for th in ("1", "4", "16"):
        os.environ["EOFX_SKETCH_THREADS"] = th
        t = time.perf_counter(); [engine.sketch_matrix(*shape, 5) for _ in range(5)]; tn = (time.perf_counter() - t) / 5 * 1e3
        print(shape, f"native {th:>2} threads {tn:7.2f} ms")
    t = time.perf_counter(); [np.random.RandomState(5).normal(size=shape).astype(np.float32) for _ in range(3)]; tp = (time.perf_counter() - t) / 3 * 1e3
    print(shape, f"numpy               {tp:7.2f} ms")
