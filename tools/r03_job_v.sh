#!/bin/bash
# round 3, GPU job V: repeated full GPU suite (flakiness / determinism check of the final code) + fuzz sweeps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
for i in 1 2; do
  timeout 2700 python -m pytest tests -x -q -m gpu > $O/gputests_$i.log 2>&1; echo "gpu tests run $i rc=$?" | tee -a $O/summary.txt; tail -2 $O/gputests_$i.log | head -1
done
for seed in 1 2; do timeout 900 python tools/fuzz_complex.py $seed 40 2>&1 | grep -v amdgpu.ids | tail -5; done | tee $O/fuzz_complex.txt
timeout 900 python tools/fuzz_cpcca.py 3 30 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/fuzz_cpcca.txt
timeout 900 python tools/fuzz_large.py 2 6 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/fuzz_large.txt
