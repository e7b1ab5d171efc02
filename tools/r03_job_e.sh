#!/bin/bash
# round 3, GPU job E: masked in-place layout (tests + land-mask probe), then the driver's own bench command (all legs)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py tests/test_gpu_bootstrap.py tests/test_gpu_rotation.py -x -q -m gpu > $O/tests1.log 2>&1; echo "tests1 rc=$?" | tee $O/summary.txt; tail -4 $O/tests1.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "masked or baseline_sizes" > $O/tests2.log 2>&1; echo "tests2 rc=$?" | tee -a $O/summary.txt; tail -4 $O/tests2.log
timeout 600 python tools/nan_probe.py > $O/nan_probe.txt 2>&1; cat $O/nan_probe.txt | tail -7
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s" | tee -a $O/summary.txt
tail -3 $O/bench_full.err
