cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j3; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cpcca.py tests/test_gpu_pca.py -x -q 2>&1 | tail -5 > $O/pytest.txt
REPS=8 python tools/cca_trace_probe.py > $O/cca_plain.txt 2>&1
python tools/cca_probe.py > $O/cca_probe.txt 2>&1
python tools/cca_profile.py > $O/cca_cprofile.txt 2>&1
cat $O/pytest.txt $O/cca_plain.txt $O/cca_probe.txt
