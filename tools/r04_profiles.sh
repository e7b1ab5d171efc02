#!/bin/bash
# round 4 evidence set on the final code: the driver command, rocprofv3 kernel trace + stats of the bench, PMC passes (HBM
# traffic, SQ counters) of the streaming kernels, and the config-3 call (Gram route): timeline + PMC of gram_nt_kernel
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04final; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee $O/summary.txt; grep -h "passed\|failed" $O/pytest_gpu.txt | tail -2
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full_driver_command.json 2> $O/bench_full.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s" | tee -a $O/summary.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o p --output-format csv -- python $R/bench.py --no-traffic --no-cpu-baseline --no-configs --steps 4 --warmup 2 > $O/bench_under_rocprof.json 2> $O/trace.err
cd $R
python tools/prof_summary.py $O/trace > $O/bench_kernel_trace_summary.txt 2>&1
python tools/trace_gaps.py $O/trace > $O/one_fit_timeline.txt 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rm -rf $O/trace
cd /tmp
# config 3 with the total squared covariance, in place: timeline of one call
TSC=1 LAYOUT=inplace timeout 600 rocprofv3 --kernel-trace --stats -d $O/mca -o p --output-format csv -- python $R/tools/mca_timeline.py > $O/mca.log 2>&1
(cd $R && python tools/trace_gaps.py $O/mca panel_import_kernel > $O/mca_timeline.txt 2>&1; python tools/prof_summary.py $O/mca > $O/mca_kernel_trace_summary.txt 2>&1)
cp $(find $O/mca -name "*kernel_stats.csv" | head -1) $O/mca_kernel_stats.csv
rm -rf $O/mca
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  TSC=1 LAYOUT=inplace timeout 600 rocprofv3 --kernel-trace --pmc $set -d $O/gpmc$i -o p --output-format csv -- python $R/tools/mca_timeline.py > $O/gpmc$i.log 2> $O/gpmc$i.err
  (cd $R && python tools/prof_summary.py $O/gpmc$i > $O/gpmc${i}_full.txt 2>&1; awk '/^# PMC/{p=1} p' $O/gpmc${i}_full.txt | grep -A 9 "gram_nt_kernel\|planes_split\|^# PMC" > $O/gram_pmc${i}_summary.txt)
  rm -rf $O/gpmc$i $O/gpmc${i}_full.txt
done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $O/pmc$i -o p --output-format csv -- python $R/bench.py --no-traffic --no-cpu-baseline --no-configs --steps 2 --warmup 1 > $O/pmc$i.json 2> $O/pmc$i.err
  (cd $R && python tools/prof_summary.py $O/pmc$i > $O/pmc${i}_full.txt 2>&1; awk '/^# PMC/{p=1} p' $O/pmc${i}_full.txt | grep -A 9 "atb_f16_fit_kernel\|atb_f16_kernel<2, true\|axb_f16_kernel<4\|axb_f16_dma_kernel\|axb_bsplit\|^# PMC" > $O/pmc${i}_summary.txt)
  rm -rf $O/pmc$i $O/pmc${i}_full.txt
done
python $R/tools/pca_probe.py > $O/pca_probe.txt 2>&1
python $R/bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 > $O/eighth.json 2> $O/eighth.err
EOFX_AXB_DMA=0 python $R/bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 > $O/eighth_register_path.json 2>/dev/null
EOFX_AXB_DMA=0 python $R/bench.py --no-traffic --no-cpu-baseline --no-configs --steps 10 --warmup 3 > $O/bench_register_path.json 2>/dev/null
python $R/bench.py --no-traffic --no-cpu-baseline --no-configs --steps 10 --warmup 3 > $O/bench_dma_path.json 2>/dev/null
$R/build/rinv_phase_probe > $O/rinv_phases_now.txt 2>&1
ls -la $O
