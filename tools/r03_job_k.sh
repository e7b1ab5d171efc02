#!/bin/bash
# round 3, GPU job K: host overhead of a fit, the amax-atomics fix at 1/8 and full size, parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
timeout 600 python tools/fit_overhead_probe.py 180 > $O/overhead_eighth.txt 2>&1; head -30 $O/overhead_eighth.txt | cut -c1-150
timeout 600 python tools/fit_overhead_probe.py 1440 > $O/overhead_full.txt 2>&1; head -3 $O/overhead_full.txt | cut -c1-200
python bench.py --gpus 1 --steps 20 --warmup 5 --nlon 180 --no-cpu-baseline --no-configs > $O/eighth.json 2> $O/eighth.err; python -c "
import json;d=json.loads(open('$O/eighth.json').read().strip().splitlines()[-1]);print('eighth', d['ms_per_step'], d['phase_ms'])"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/full.json 2> $O/full.err; python -c "
import json;d=json.loads(open('$O/full.json').read().strip().splitlines()[-1]);print('full', d['ms_per_step'], d['phase_ms'], d['roofline']['by_kernel'])"
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
