#!/bin/bash
# round 3, GPU job N: fuzz sweeps over the paths built this round (one-call fit, masks, lean complex) + the older fuzzers
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03n; mkdir -p $O
for seed in 1 2 3; do timeout 900 python tools/fuzz_fit.py $seed 60 2>&1 | grep -v amdgpu.ids | tail -8; done | tee $O/fuzz_fit.txt
FUZZ_LAYOUT=inplace timeout 900 python tools/fuzz_parity.py 11 60 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/fuzz_parity_inplace.txt
timeout 900 python tools/fuzz_hilbert.py 5 40 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/fuzz_hilbert.txt
timeout 900 python tools/fuzz_mca.py 7 30 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/fuzz_mca.txt
timeout 900 python tools/fuzz_rotator.py 3 20 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/fuzz_rotator.txt
