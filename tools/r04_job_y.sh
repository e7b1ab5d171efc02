#!/bin/bash
# round 4: after the host-side trims (eigen-solver, small matrix products): tests and the short fits
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04y; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_sharded_native.py tests/test_gpu_models.py -x -q > $O/pytest.txt 2>&1; grep -h "passed\|failed\|^E " $O/pytest.txt | tail -4
for i in 1 2 3; do
  python bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eighth', d['ms_per_step'])"
  python bench.py --nsamples 5000 --nlat 360 --nlon 720 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config2', d['ms_per_step'])"
done | tee $O/short_fits.txt
