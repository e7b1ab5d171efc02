#!/bin/bash
# round 3, GPU job F: full GPU suite after the masked layout / bootstrap kernels / wide complex tile / tolerance changes,
# complex probe at k = 20 and k = 40, bootstrap probe, then the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; echo "gpu tests rc=$?" | tee $O/summary.txt; tail -12 $O/gputests.log
ENGINE_ONLY=1 timeout 600 python tools/complex_probe.py 8000 720 1440 20 > $O/complex_k20.txt 2>&1; tail -2 $O/complex_k20.txt
ENGINE_ONLY=1 timeout 600 python tools/complex_probe.py 8000 720 1440 40 > $O/complex_k40.txt 2>&1; tail -2 $O/complex_k40.txt
timeout 600 python tools/bootstrap_probe.py > $O/bootstrap_probe.txt 2>&1; tail -4 $O/bootstrap_probe.txt
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s" | tee -a $O/summary.txt
tail -3 $O/bench_full.err
