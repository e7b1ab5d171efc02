#!/bin/bash
# round 3, GPU job Y: long fuzz sweeps on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03y; mkdir -p $O
for seed in 21 22 23 24 25; do timeout 1200 python tools/fuzz_fit.py $seed 100 2>&1 | grep -v amdgpu.ids | tail -6; done | tee $O/fuzz_fit.txt
for seed in 31 32; do FUZZ_LAYOUT=inplace timeout 1200 python tools/fuzz_parity.py $seed 100 2>&1 | grep -v amdgpu.ids | tail -4; done | tee $O/fuzz_parity.txt
timeout 1200 python tools/fuzz_parity.py 33 100 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/fuzz_parity.txt
for seed in 41 42; do timeout 1200 python tools/fuzz_complex.py $seed 60 2>&1 | grep -v amdgpu.ids | tail -4; done | tee $O/fuzz_complex.txt
timeout 1200 python tools/fuzz_mca.py 51 60 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/fuzz_mca.txt
timeout 1200 python tools/fuzz_hilbert.py 61 80 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/fuzz_hilbert.txt
timeout 1200 python tools/fuzz_rotator.py 71 30 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/fuzz_rotator.txt
timeout 1200 python tools/fuzz_cpcca.py 81 60 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/fuzz_cpcca.txt
