#!/bin/bash
# round 4: whole GPU suite + the driver's bench command on the tree after the N1 / N4 changes
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04p; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
S=$SECONDS
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$? wall=$((SECONDS-S))s" | tee $O/summary.txt
tail -5 $O/pytest_gpu.txt
S=$SECONDS
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full_driver_command.json 2> $O/bench_full.err; echo "bench rc=$? wall=$((SECONDS-S))s" | tee -a $O/summary.txt
tail -c 3000 $O/bench_full_driver_command.json
tail -5 $O/bench_full.err
