# rocprofv3 --kernel-trace --stats over the default bench (no CPU baseline leg): per-kernel table under gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02trace}
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG -o p --output-format csv -- python $R/bench.py --no-traffic --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/$TAG.json 2> $R/gpurun_out/$TAG.err
cd $R
python tools/prof_summary.py gpurun_out/$TAG > gpurun_out/${TAG}_summary.txt 2>&1
head -45 gpurun_out/${TAG}_summary.txt
