cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j6; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_complex.py tests/test_gpu_hilbert_operator.py tests/test_gpu_models.py tests/test_gpu_golden.py -x -q 2>&1 | tail -5 > $O/pytest.txt
python tools/c5_operator_probe.py > $O/c5.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/c5 -o p --output-format csv -- python $R/tools/c5_operator_probe.py > $O/c5_probe_rocprof.txt 2>&1
cd $R
python tools/trace_gaps.py $O/c5 colstats_tr_kernel > $O/c5_timeline.txt 2>&1
rm -rf $O/c5
cat $O/pytest.txt; grep rep $O/c5.txt; grep colargminmax $O/c5_timeline.txt
