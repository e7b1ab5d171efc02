#!/bin/bash
# round 6, job e: kernel timeline of one converge call at config-5 size
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06_c5conv -o p --output-format csv -- python $R/tools/c5_converge_probe.py 2>&1 | grep "^rule"
cd $R
python tools/prof_summary.py gpurun_out/r06_c5conv > gpurun_out/r06_c5_converge_kernel_trace_summary.txt 2>&1
head -40 gpurun_out/r06_c5_converge_kernel_trace_summary.txt | cut -c1-200
python tools/trace_gaps.py gpurun_out/r06_c5conv panel_import_kernel > gpurun_out/r06_c5_converge_timeline.txt 2>&1
tail -3 gpurun_out/r06_c5_converge_timeline.txt
rm -rf gpurun_out/r06_c5conv
