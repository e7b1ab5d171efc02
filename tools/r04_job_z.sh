#!/bin/bash
# round 4: the whitened cross models at config-3 size after the wide matmul route: timings and the kernel table of one run
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04z; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python tools/cca_probe.py > $O/cca_probe.txt 2>&1; grep "fit " $O/cca_probe.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/tr -o p --output-format csv -- python $R/tools/cca_probe.py > /dev/null 2> $O/tr.err
cd $R
python tools/prof_summary.py $O/tr > $O/cca_kernels.txt 2>&1
rm -rf $O/tr
head -32 $O/cca_kernels.txt | cut -c1-170
