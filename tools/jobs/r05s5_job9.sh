cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j9; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
MODEL=MCA REPS=2 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -d $O/pmc -o p --output-format csv -- python $R/tools/cca_trace_probe.py > $O/pmc_run.txt 2>&1
cd $R
python tools/prof_summary.py $O/pmc > $O/pmc_full.txt 2>&1
ls $O/pmc | head
rm -rf $O/pmc
grep -c . $O/pmc_full.txt
