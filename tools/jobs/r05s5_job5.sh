cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j5; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_hilbert_operator.py -x -q 2>&1 | tail -5 > $O/pytest.txt
OVERLAP=0 python tools/c5_operator_probe.py > $O/c5_serial.txt 2>&1
for cfg in "16 16" "4 16" "64 16" "16 32" "16 0" "8 64"; do set -- $cfg
  EOFX_SUMSQ_PAIRS=$1 EOFX_SUMSQ_SPARE=$2 python tools/c5_operator_probe.py > $O/c5_overlap_$1_$2.txt 2>&1
  echo "pairs $1 spare $2: $(grep rep4 $O/c5_overlap_$1_$2.txt)"
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/c5 -o p --output-format csv -- python $R/tools/c5_operator_probe.py > $O/c5_probe_rocprof.txt 2>&1
cd $R
python tools/trace_tail.py $O/c5 160 100 > $O/c5_timeline.txt 2>&1
rm -rf $O/c5
cat $O/pytest.txt; grep rep4 $O/c5_serial.txt
