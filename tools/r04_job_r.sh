#!/bin/bash
# round 4: where the fixed per-fit cost sits at one eighth of the grid (bench.py --nlon 180), and the kernels of one ResidentPCA fit
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04r; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 > $O/eighth.json 2> $O/eighth.err
python -c "
import json;d=json.load(open('$O/eighth.json'));print('eighth', d['ms_per_step'], d['roofline']['by_kernel'])"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o p --output-format csv -- python $R/bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 4 --warmup 2 > $O/eighth_rocprof.json 2> $O/trace.err
cd $R
python tools/trace_gaps.py $O/trace > $O/eighth_timeline.txt 2>&1
python tools/prof_summary.py $O/trace > $O/eighth_summary.txt 2>&1
rm -rf $O/trace
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/pca -o p --output-format csv -- python $R/tools/pca_probe.py > $O/pca_probe.txt 2>&1
cd $R
python tools/prof_summary.py $O/pca > $O/pca_summary.txt 2>&1
rm -rf $O/pca
tail -50 $O/eighth_timeline.txt | cut -c1-120
head -40 $O/pca_summary.txt | cut -c1-150
