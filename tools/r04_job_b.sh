#!/bin/bash
# round 4 job B: kernel timeline of one config-3 MCA call with the TSC (Gram route), both layouts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04b
for lay in copy inplace; do
  TSC=1 LAYOUT=$lay timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/r04b/tr_$lay -o p --output-format csv -- python $R/tools/mca_timeline.py > $R/gpurun_out/r04b/log_$lay.txt 2>&1
  python $R/tools/trace_gaps.py $R/gpurun_out/r04b/tr_$lay panel_import_kernel > $R/gpurun_out/r04b/mca_timeline_$lay.txt 2>&1
  rm -rf $R/gpurun_out/r04b/tr_$lay
done
cd $R
awk '{k=$4; for(i=5;i<=NF;i++)k=k" "$i; d[k]+=$2; c[k]++} END{for(k in d) printf "%10.1f us %4d x  %s\n", d[k], c[k], k}' gpurun_out/r04b/mca_timeline_copy.txt | sort -rn | head -25
tail -1 gpurun_out/r04b/mca_timeline_copy.txt
awk '{k=$4; for(i=5;i<=NF;i++)k=k" "$i; d[k]+=$2; c[k]++} END{for(k in d) printf "%10.1f us %4d x  %s\n", d[k], c[k], k}' gpurun_out/r04b/mca_timeline_inplace.txt | sort -rn | head -25
tail -1 gpurun_out/r04b/mca_timeline_inplace.txt
