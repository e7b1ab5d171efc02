#!/bin/bash
# round 3, GPU job M: sketch latency, sharded driver at world 1 after the merged collectives, full GPU suite, driver bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O
python tools/sketch_latency_probe.py > $O/sketch_latency.txt 2>&1; cat $O/sketch_latency.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --nlon 180 --no-cpu-baseline --no-configs --force-sharded > $O/eighth_sharded.json 2> $O/eighth_sharded.err; python -c "
import json;d=json.loads(open('$O/eighth_sharded.json').read().strip().splitlines()[-1]);print('eighth sharded', d['ms_per_step'], d['phase_ms'], d['comm'])"
timeout 2700 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; echo "gpu tests rc=$?" | tee $O/summary.txt; tail -6 $O/gputests.log
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s" | tee -a $O/summary.txt
tail -3 $O/bench_full.err
python -c "
import json;d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['roofline']['frac'], {k:(v.get('ms') if isinstance(v,dict) else v) for k,v in d['configs'].items()})"
