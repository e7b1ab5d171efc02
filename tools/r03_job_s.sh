#!/bin/bash
# round 3, GPU job S: Gram kernel with one partial per workgroup: probe, parity + pca + cpcca tests (wide Gram panels), bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
python tools/small_kernel_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/small_kernel_probe.txt
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pca.py tests/test_gpu_cpcca.py tests/test_gpu_complex.py tests/test_gpu_rotation.py tests/test_gpu_golden.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -4 $O/tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic > $O/full.json 2> $O/full.err; python -c "
import json;d=json.loads(open('$O/full.json').read().strip().splitlines()[-1]);print('full', d['ms_per_step'], d['roofline']['by_kernel'])"
