#!/bin/bash
# round 3, GPU job R: masked in-place passes that skip fully masked slabs (wave-level in atb, active pair list in axb)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03r; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bootstrap.py tests/test_gpu_models.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -5 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "masked" > $O/masked_fullsize.log 2>&1; echo "masked fullsize rc=$?" | tee -a $O/summary.txt; tail -3 $O/masked_fullsize.log
MASK=blobs timeout 900 python tools/nan_probe.py > $O/nan_probe_blobs.txt 2>&1; grep -v amdgpu.ids $O/nan_probe_blobs.txt | cut -c1-200
timeout 900 python tools/nan_probe.py > $O/nan_probe_random.txt 2>&1; grep -v amdgpu.ids $O/nan_probe_random.txt | cut -c1-200
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic > $O/full.json 2> $O/full.err; python -c "
import json;d=json.loads(open('$O/full.json').read().strip().splitlines()[-1]);print('full', d['ms_per_step'], d['roofline']['by_kernel'])"
for seed in 6 7; do timeout 900 python tools/fuzz_fit.py $seed 60 2>&1 | grep -v amdgpu.ids | tail -4; done | tee $O/fuzz_fit.txt
