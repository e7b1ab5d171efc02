cd $GRAFT_REPO_ROOT; O=gpurun_out/j12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_complex.py tests/test_gpu_hilbert_operator.py tests/test_gpu_pca.py -x -q > $O/pytest.txt 2>&1
grep -E "passed|failed|error|Error" $O/pytest.txt | tail -5
BULK=1 python tools/fuzz_complex.py 2 60 2>&1 | grep -v amdgpu | tail -4
python tools/fuzz_complex.py 1 60 2>&1 | grep -v amdgpu | tail -3
python bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eighth', d['ms_per_step'], {k:v['mean_launch_ms'] for k,v in d['roofline']['by_kernel'].items()})"
