#!/bin/bash
# round 4 job G: the whole driver line with the new gates
mkdir -p gpurun_out/r04g
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04g/bench.json 2> gpurun_out/r04g/bench.err
echo rc=$? wall=${SECONDS}s
tail -5 gpurun_out/r04g/bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r04g/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"].get("mfma_frac"), d["roofline"]["by_kernel"])
print(json.dumps(d["parity"], indent=1)[:1500])
for k, v in d["configs"].items():
    print(k, json.dumps(v)[:900])
PY
