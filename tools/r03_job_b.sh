#!/bin/bash
# round 3, GPU job B: kernel trace of the fused bench + one-fit timeline
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o p --output-format csv -- python $R/bench.py --no-traffic --no-cpu-baseline --steps 4 --warmup 2 > $O/bench_under_rocprof.json 2> $O/bench.err
cd $R
python tools/prof_summary.py $O/trace > $O/kernel_trace_summary.txt 2>&1
python tools/trace_gaps.py $O/trace > $O/one_fit_timeline.txt 2>&1
head -30 $O/kernel_trace_summary.txt
tail -5 $O/one_fit_timeline.txt
rm -rf $O/trace/*/*.db 2>/dev/null
du -sh $O
