#!/bin/bash
# round 4: chol_rinv_kernel with the blocked inverse / straight-line panels: tests that lean on it, then the eighth share and config 2
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04s; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_pca.py tests/test_gpu_gram.py tests/test_gpu_sharded_native.py tests/test_gpu_complex.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python tools/rinv_probe.py > $O/rinv_probe.txt 2>&1; tail -12 $O/rinv_probe.txt
python bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 > $O/eighth.json 2> $O/eighth.err
python -c "
import json;d=json.load(open('$O/eighth.json'));print('eighth', d['ms_per_step'], {k:v['mean_launch_ms'] for k,v in d['roofline']['by_kernel'].items()})"
python bench.py --no-traffic --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python -c "
import json;d=json.load(open('$O/bench.json'));print('c4', d['ms_per_step'], {k:v['mean_launch_ms'] for k,v in d['roofline']['by_kernel'].items()}); c=d['config']['configs'] if 'configs' in d['config'] else d['configs']; print('c2', c['config2']['ms'], 'c3', c['config3']['ms'], 'c5', c['config5']['ms'])"
