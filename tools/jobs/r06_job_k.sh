#!/bin/bash
# round 6, job k: GPU suite + smoke + the driver's command on the FINAL tree
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -h "passed\|failed" gpurun_out/r06_pytest_gpu.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_full_driver_command.json 2> gpurun_out/r06_bench_full.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_full_driver_command.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
c = d["configs"]
print("c2", c["config2"]["ms"], "c3", c["config3"]["ms"], "c5", c["config5"]["ms"], "auto", c["config5"]["n_iter_auto"]["ms"], "model", {k: v["phase_ms"] for k, v in c["model_level"].items()})
PY
