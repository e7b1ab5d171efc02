cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j1; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/c5 -o p --output-format csv -- python $R/tools/c5_operator_probe.py > $O/c5_probe.txt 2>&1
cd $R
python tools/prof_summary.py $O/c5 > $O/c5_summary.txt 2>&1
python tools/trace_gaps.py $O/c5 colstats_tr_kernel > $O/c5_timeline.txt 2>&1
rm -rf $O/c5
tail -5 $O/c5_probe.txt
