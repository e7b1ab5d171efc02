#!/bin/bash
# round 3, GPU job T: axb_f16 with the bank-conflict-free A staging: A/B against the previous build on the SAME box
# (build/libeofx_prev.so = HEAD before the change), then the tests that exercise the kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03t; mkdir -p $O
cp xeofs_amd/lib/libeofx.so /tmp/libeofx_new.so
for round in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then cp build/libeofx_prev.so xeofs_amd/lib/libeofx.so; else cp /tmp/libeofx_new.so xeofs_amd/lib/libeofx.so; fi
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic > $O/ab_${which}_$round.json 2>/dev/null
    python -c "
import json;d=json.loads(open('$O/ab_${which}_$round.json').read().strip().splitlines()[-1]);print('$which', $round, d['ms_per_step'], {k:v['mean_launch_ms'] for k,v in d['roofline']['by_kernel'].items()})" | tee -a $O/ab.txt
  done
done
cp /tmp/libeofx_new.so xeofs_amd/lib/libeofx.so
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_complex.py tests/test_gpu_bootstrap.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -3 $O/tests.log
