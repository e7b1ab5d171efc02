#!/bin/bash
# round 3, GPU job P: the driver's bench command with roofline.traffic re-counted in the run (two rocprofv3 PMC passes inside
# bench.py), then smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03p2; mkdir -p $O
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s" | tee $O/summary.txt
grep -v amdgpu.ids $O/bench_full.err | tail -5
python -c "
import json;d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1]);r=d['roofline'];print(d['ms_per_step'], d['value'], r['frac'], r['traffic'], r['traffic_by_kernel']); print(r['traffic_source'])"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
