# HBM traffic of the bench kernels: separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py, kernel trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/r02bpmc_$c -o p --output-format csv -- python $R/bench.py --no-traffic --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/r02bpmc_$c.json 2> $R/gpurun_out/r02bpmc_$c.err
done
cd $R
for c in FETCH_SIZE WRITE_SIZE; do python tools/prof_summary.py gpurun_out/r02bpmc_$c | grep -A 3 "^void eofx::atb_f16\|^void eofx::axb_f16\|^void eofx::apply_kernel\|^eofx::colstats_kernel"; done
