#!/bin/bash
# round 3: everything that goes under profiles/r03_* in one GPU job (rocprofv3 kernel trace + stats of the default bench,
# the separate PMC passes -- HBM traffic and SQ counters --, a one-fit timeline, and the probe outputs DESIGN.md quotes)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 --no-configs > $O/bench_n1.json 2> $O/bench_n1.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o p --output-format csv -- python $R/bench.py --no-traffic --no-cpu-baseline --steps 4 --warmup 2 > $O/bench_under_rocprof.json 2> $O/trace.err
cd $R
python tools/prof_summary.py $O/trace > $O/bench_kernel_trace_summary.txt 2>&1
python tools/trace_gaps.py $O/trace > $O/one_fit_timeline.txt 2>&1
cp $O/trace/*/p_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null || cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rm -rf $O/trace
cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $O/pmc$i -o p --output-format csv -- python $R/bench.py --no-traffic --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc$i.json 2> $O/pmc$i.err
  (cd $R && python tools/prof_summary.py $O/pmc$i > $O/pmc${i}_full.txt 2>&1; awk '/^# PMC/{p=1} p' $O/pmc${i}_full.txt | grep -A 9 "atb_f16_fit_kernel\|atb_f16_kernel<2, true\|axb_f16_kernel<4\|^# PMC" > $O/pmc${i}_summary.txt)
  rm -rf $O/pmc$i $O/pmc${i}_full.txt
done
cd $R
timeout 600 python tools/nan_probe.py > $O/nan_probe.txt 2>&1
ENGINE_ONLY=1 timeout 600 python tools/complex_probe.py 8000 720 1440 20 > $O/complex_k20.txt 2>&1
ENGINE_ONLY=1 timeout 600 python tools/complex_probe.py 8000 720 1440 40 > $O/complex_k40.txt 2>&1
timeout 300 python tools/sketch_probe.py > $O/sketch_probe.txt 2>&1
[ -x build/inflight_probe ] && ./build/inflight_probe > $O/inflight_probe.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/btrace -o p --output-format csv -- python $R/tools/bootstrap_probe.py > $O/bootstrap_probe.txt 2> $O/btrace.err
cd $R
python tools/prof_summary.py $O/btrace > $O/bootstrap_probe_kernel_trace_summary.txt 2>&1
rm -rf $O/btrace
ls -la $O; du -sh $O
