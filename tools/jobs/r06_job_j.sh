#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06_cca -o p --output-format csv -- python $R/tools/cca_profile2.py 2>&1 | grep "^CCA"
cd $R
python tools/prof_summary.py gpurun_out/r06_cca > gpurun_out/r06_cca_kernel_trace_summary.txt 2>&1
head -45 gpurun_out/r06_cca_kernel_trace_summary.txt | cut -c1-190
rm -rf gpurun_out/r06_cca
