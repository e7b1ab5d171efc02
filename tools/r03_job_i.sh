#!/bin/bash
# round 3, GPU job I: one rank's share of the 8-GPU strong-scaling run (p / 8 features), native single-rank entry next to
# the sharded (Python-orchestrated + all-reduce) driver at world 1, with a timeline of one sharded fit
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 --nlon 180 --no-cpu-baseline --no-configs > $O/eighth.json 2> $O/eighth.err; tail -c 600 $O/eighth.json
python bench.py --gpus 1 --steps 20 --warmup 5 --nlon 180 --no-cpu-baseline --no-configs --force-sharded > $O/eighth_sharded.json 2> $O/eighth_sharded.err; tail -c 900 $O/eighth_sharded.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-traffic --gpus 1 --steps 3 --warmup 1 --nlon 180 --no-cpu-baseline --no-configs --force-sharded > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py $O/prof > $O/sharded_eighth_timeline.txt 2>&1; head -120 $O/sharded_eighth_timeline.txt | cut -c1-150
rm -rf $O/prof
