#!/bin/bash
# round 3, GPU job D: the driver's own bench command (all legs) + the full GPU test suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
/usr/bin/time -v python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?" | tee $O/summary.txt
tail -3 $O/bench_full.err
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt; tail -15 $O/gputests.log
