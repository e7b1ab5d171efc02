cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j2; mkdir -p $O; export TMPDIR=/tmp
python tools/cca_trace_probe.py > $O/cca_plain.txt 2>&1
python tools/cca_profile.py > $O/cca_cprofile.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/cca -o p --output-format csv -- python $R/tools/cca_trace_probe.py > $O/cca_probe.txt 2>&1
cd $R
python tools/prof_summary.py $O/cca > $O/cca_summary.txt 2>&1
python tools/trace_tail.py $O/cca 330 30 > $O/cca_timeline.txt 2>&1
rm -rf $O/cca
cat $O/cca_plain.txt; grep "fit" $O/cca_probe.txt
