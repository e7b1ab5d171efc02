#!/bin/bash
# round 4 job J: whole GPU suite + the driver line (state after the bench gates / complex n_iter=-2)
mkdir -p gpurun_out/r04j
python -m pytest tests -x -q -m gpu > gpurun_out/r04j/tests_gpu.txt 2>&1
tail -3 gpurun_out/r04j/tests_gpu.txt
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04j/bench.json 2> gpurun_out/r04j/bench.err
echo rc=$? wall=${SECONDS}s
tail -2 gpurun_out/r04j/bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r04j/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"])
print("config5", json.dumps(d["configs"]["config5"])[:1800])
PY
