# HBM / L2 counters of the in-place product's debug variants (tools/probes/axb_probe.hip): one --pmc pass per counter
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE TCC_EA0_RDREQ_sum "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/axbpmc_$tag -o p --output-format csv -- $R/build/axb_probe > $R/gpurun_out/axbpmc_$tag.txt 2>&1
done
cd $R
for c in FETCH_SIZE TCC_EA0_RDREQ_sum TCC_HIT_sum_TCC_MISS_sum; do python tools/prof_summary.py gpurun_out/axbpmc_$c 2>&1 | head -60; done
