#!/bin/bash
# round 3, GPU job A: the fused fit and the restructured rSVD core -- parity tests first, then the bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt; tail -5 $O/parity.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err; echo "bench fused rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --two-step > $O/bench_twostep.json 2> $O/bench_twostep.err; echo "bench two-step rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json
for f in ("bench_fused","bench_twostep"):
    try:
        d=json.loads(open(f"gpurun_out/r03a/{f}.json").read())
        print(f, d["ms_per_step"], d["value"], d["phase_ms"], d["roofline"].get("by_kernel"), d["parity"], d["config"].get("field_reads_per_fit"))
    except Exception as e:
        print(f, "failed", e)
PY
