#!/bin/bash
# round 6, job d: the whole GPU suite and the driver's bench command on the current tree
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/ -q -m gpu -x 2>&1 | tail -30 > gpurun_out/r06_pytest_gpu.txt
tail -12 gpurun_out/r06_pytest_gpu.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_full_driver_command.json 2> gpurun_out/r06_bench_full_driver_command.err
tail -5 gpurun_out/r06_bench_full_driver_command.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_full_driver_command.json").read().strip().splitlines()[-1])
print("value", d["value"], d["unit"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "by_kernel", d["roofline"].get("by_kernel"))
c = d.get("configs", {})
for k in ("config1", "config2", "config3"):
    print(k, c.get(k, {}).get("ms"), c.get(k, {}).get("frac"))
c5 = c.get("config5", {})
print("config5", c5.get("ms"), c5.get("phase_ms"), c5.get("power_iterations"), "frac", c5.get("frac"), "auto:", c5.get("n_iter_auto"))
print("config5 gate", c5.get("parity", {}).get("oracle_gate"))
print("model_level", c.get("model_level"))
print("default", c.get("config3", {}).get("default_arguments"), c.get("config3", {}).get("default_arguments_cca"))
PY
