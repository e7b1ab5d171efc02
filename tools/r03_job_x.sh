#!/bin/bash
# round 3, GPU job X: what the driver runs at round end, on the final tree: full GPU suite, smoke(), the bench command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03x; mkdir -p $O
timeout 2700 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; echo "gpu tests rc=$?" | tee $O/summary.txt; grep -n "passed\|failed" $O/gputests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $O/summary.txt
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s" | tee -a $O/summary.txt
python -c "
import json;d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1]);r=d['roofline'];print(d['ms_per_step'], d['value'], r['frac'], r['traffic'], {k:(v.get('ms') if isinstance(v,dict) else v) for k,v in d['configs'].items()}, d['vs_baseline'])"
