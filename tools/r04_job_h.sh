#!/bin/bash
# round 4 job H: adaptive iteration count of the complex rSVD -- complex tests, then the whole driver line
mkdir -p gpurun_out/r04h
python -m pytest tests/test_gpu_complex.py tests/test_gpu_complex_cross.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -x -q -m gpu -k "complex or g6 or config5 or hilbert" > gpurun_out/r04h/tests_complex.txt 2>&1
tail -4 gpurun_out/r04h/tests_complex.txt
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04h/bench.json 2> gpurun_out/r04h/bench.err
echo rc=$? wall=${SECONDS}s
tail -3 gpurun_out/r04h/bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r04h/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"])
for k in ("config3", "config5"):
    print(k, json.dumps(d["configs"][k])[:1500])
PY
