#!/bin/bash
# round 4 job L: kernel timeline of one fit at one eighth of the grid (the rank's share of an 8-GPU run)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04l
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/r04l/tr -o p --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --nlon 180 > $R/gpurun_out/r04l/log.txt 2>&1
python $R/tools/trace_gaps.py $R/gpurun_out/r04l/tr fit_probe_kernel > $R/gpurun_out/r04l/eighth_timeline.txt 2>&1
rm -rf $R/gpurun_out/r04l/tr
cd $R
awk 'NR>2 {k=$4; for(i=5;i<=NF;i++)k=k" "$i; d[k]+=$2; c[k]++; g[k]+=$3} END{for(k in d) printf "%10.1f us %4d x  gaps %8.1f  %s\n", d[k], c[k], g[k], k}' gpurun_out/r04l/eighth_timeline.txt | sort -rn | head -30
tail -1 gpurun_out/r04l/eighth_timeline.txt
