#!/bin/bash
# round 4: where the fixed per-fit cost sits at one eighth of the grid (bench.py --nlon 180): timeline of one fit
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04r; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 > $O/eighth.json 2> $O/eighth.err
python -c "
import json;d=json.load(open('$O/eighth.json'));print('eighth', d['ms_per_step'])"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o p --output-format csv -- python $R/bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 4 --warmup 2 > $O/eighth_rocprof.json 2> $O/trace.err
cd $R
python tools/trace_gaps.py $O/trace > $O/eighth_timeline.txt 2>&1
rm -rf $O/trace
grep -v "atb_f16_kernel\|axb_f16\|splitk_reduce\|gram_mfma\|f64_reduce\|chol_rinv\|panel_matmul" $O/eighth_timeline.txt | cut -c1-110 | tail -45
