cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j7; mkdir -p $O; export TMPDIR=/tmp
MODEL=MCA REPS=5 python tools/cca_trace_probe.py > $O/mca_plain.txt 2>&1
cd /tmp
MODEL=MCA REPS=4 timeout 600 rocprofv3 --kernel-trace --stats -d $O/mca -o p --output-format csv -- python $R/tools/cca_trace_probe.py > $O/mca_probe.txt 2>&1
cd $R
python tools/prof_summary.py $O/mca > $O/mca_summary.txt 2>&1
python tools/trace_tail.py $O/mca 175 150 > $O/mca_timeline.txt 2>&1
rm -rf $O/mca
EOFX_PCA_TRACE=1 MODEL=MCA REPS=3 python tools/cca_trace_probe.py > $O/mca_pca_trace.txt 2>&1
grep fit $O/mca_plain.txt
