#!/bin/bash
# round 6, job g: the multi-rank form of the driver's command -- headline + configs_sharded -- on 2 ranks sharing the GPU (gloo
# callback binding) and on world-1 RCCL (--force-sharded)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --backend gloo --same-gpu --steps 3 --warmup 1 > gpurun_out/r06_bench_2rank_same_gpu.json 2> gpurun_out/r06_bench_2rank_same_gpu.err
echo "rc=$?"; tail -4 gpurun_out/r06_bench_2rank_same_gpu.err | cut -c1-300
python bench.py --force-sharded --steps 5 --warmup 2 > gpurun_out/r06_force_sharded_world1.json 2> gpurun_out/r06_force_sharded_world1.err
echo "rc=$?"; tail -4 gpurun_out/r06_force_sharded_world1.err | cut -c1-300
python - <<'PY'
import json
for f in ("gpurun_out/r06_bench_2rank_same_gpu.json", "gpurun_out/r06_force_sharded_world1.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "NO JSON", e); continue
    print(f, "value", d["value"], "ms", d["ms_per_step"], "n_gpus", d["n_gpus"], "entry", d["config"]["entry"][:60])
    print("   comm", {k: d["comm"].get(k) for k in ("ranks_seen", "allreduce_calls_per_fit", "binding")})
    for k, v in (d.get("configs_sharded") or {}).items():
        print("   ", k, v.get("ms_per_step"), v.get("value"), v.get("config", {}).get("entry", "")[:70], v.get("parity"), v.get("comm", {}).get("allreduce_calls_per_fit"))
PY
