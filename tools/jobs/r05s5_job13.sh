cd $GRAFT_REPO_ROOT; O=gpurun_out/j13; mkdir -p $O
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], {k[:3]:v['mean_launch_ms'] for k,v in d['roofline']['by_kernel'].items()})"; }
for rep in 1 2; do
  for lib in new trunc; do
    if [ $lib = trunc ]; then export EOFX_LIB=$PWD/build/libeofx_trunc.so; else unset EOFX_LIB; fi
    python bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 30 --warmup 5 2>/dev/null | show "eighth $lib"
    python bench.py --no-traffic --no-cpu-baseline --no-configs --steps 10 --warmup 3 2>/dev/null | show "headline $lib"
  done
done
