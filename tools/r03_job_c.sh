#!/bin/bash
# round 3, GPU job C: SQ counters of the three streaming kernels (separate --pmc passes over a short bench, kernel trace only)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/sq_counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $O/pmc$i -o p --output-format csv -- python $R/bench.py --no-traffic --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc$i.json 2> $O/pmc$i.err
  (cd $R && python tools/prof_summary.py $O/pmc$i | awk '/^# PMC/{p=1} p' | grep -A 9 "atb_f16_fit_kernel\|atb_f16_kernel<2, true\|axb_f16_kernel<4" > $O/pmc${i}_summary.txt 2>&1)
  rm -rf $O/pmc$i
done
cat $O/pmc*_summary.txt
