cd $GRAFT_REPO_ROOT; O=gpurun_out/j14; mkdir -p $O
python tools/null_mode_probe.py 2>&1 | grep -v amdgpu | tee $O/null_mode_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_models.py -x -q > $O/pytest.txt 2>&1
grep -E "passed|failed|rror" $O/pytest.txt | tail -5
