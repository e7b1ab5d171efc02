# PMC passes over tools/fused_probe.py (variant = $1, the EOFX_FUSED_TIMING value; default 0)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-0}
i=0
for c in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  EOFX_FUSED_TIMING=$V timeout 200 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/fpmc_v${V}_$i -o p --output-format csv -- python $R/tools/fused_probe.py 10000 400000 1 > $R/gpurun_out/fpmc_v${V}_$i.log 2>&1
done
cd $R
for i in 1 2; do python tools/prof_summary.py gpurun_out/fpmc_v${V}_$i | grep -A 8 "^void eofx::fused2_kernel" ; done
