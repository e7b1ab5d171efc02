import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from xeofs_amd import engine
for shape in ((10000, 60), (5000, 60), (8000, 30), (129600, 30)):
    engine.sketch_matrix(*shape, 5)
    t = time.perf_counter()
    for _ in range(20):
        engine.sketch_matrix(*shape, 5)
    d = (time.perf_counter() - t) / 20
    t = time.perf_counter()
    for _ in range(20):
        engine.SketchFuture(*shape, 5).result()
    f = (time.perf_counter() - t) / 20
    print(f"sketch {shape}: direct {1e3*d:.3f} ms, through SketchFuture {1e3*f:.3f} ms", flush=True)
