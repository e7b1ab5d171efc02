cd $GRAFT_REPO_ROOT; O=gpurun_out/j11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pca.py tests/test_gpu_cpcca.py tests/test_gpu_parity.py tests/test_gpu_rotation.py tests/test_gpu_bootstrap.py -x -q > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
MODEL=MCA REPS=5 python tools/cca_trace_probe.py 2>&1 | grep fit
python tools/cca_probe.py 2>&1 | grep fit
