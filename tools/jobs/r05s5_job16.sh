cd $GRAFT_REPO_ROOT
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], {k[:3]:v['mean_launch_ms'] for k,v in d['roofline']['by_kernel'].items()})"; }
for rep in 1 2 3; do
  for lib in new lotrunc trunc; do
    if [ $lib = new ]; then unset EOFX_LIB; else export EOFX_LIB=$PWD/build/libeofx_$lib.so; fi
    python bench.py --no-traffic --no-cpu-baseline --no-configs --steps 10 --warmup 3 2>/dev/null | show "headline $lib"
  done
done
