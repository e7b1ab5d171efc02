#!/bin/bash
# round 3, GPU job J: timeline of one native fit at one rank's share of the 8-GPU run (p / 8 features) and at config 2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03j; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p --output-format csv -- python $R/bench.py --no-traffic --gpus 1 --steps 3 --warmup 1 --nlon 180 --no-cpu-baseline --no-configs > $R/$O/prof.log 2>&1
cd $R
python tools/trace_gaps.py $O/prof > $O/native_eighth_timeline.txt 2>&1; cat $O/native_eighth_timeline.txt | cut -c1-130
rm -rf $O/prof
