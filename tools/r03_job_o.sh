#!/bin/bash
# round 3, GPU job O: the one-call fit with a land mask (masked in-place layout from the first pass)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03o; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" | tee $O/summary.txt; tail -15 $O/parity.log
timeout 900 python tools/nan_probe.py > $O/nan_probe.txt 2>&1; grep -v amdgpu.ids $O/nan_probe.txt | cut -c1-200
for seed in 4 5; do timeout 900 python tools/fuzz_fit.py $seed 60 2>&1 | grep -v amdgpu.ids | tail -6; done | tee $O/fuzz_fit.txt
