#!/bin/bash
# round 5 evidence set on the final code: GPU suite, the driver command, rocprofv3 kernel trace + stats of the bench, PMC passes
# (HBM traffic, SQ counters) of the streaming kernels, the complex branch against the reference solver (probe + two fuzz sweeps),
# config 5 phase by phase, one rank's eighth share, the default-argument MCA, the Hilbert kernel's counters.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05final; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee $O/summary.txt; grep -h "passed\|failed" $O/pytest_gpu.txt | tail -2
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full_driver_command.json 2> $O/bench_full.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s" | tee -a $O/summary.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o p --output-format csv -- python $R/bench.py --no-traffic --no-cpu-baseline --no-configs --steps 4 --warmup 2 > $O/bench_under_rocprof.json 2> $O/trace.err
cd $R
python tools/prof_summary.py $O/trace > $O/bench_kernel_trace_summary.txt 2>&1
python tools/trace_gaps.py $O/trace > $O/one_fit_timeline.txt 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rm -rf $O/trace
cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $O/pmc$i -o p --output-format csv -- python $R/bench.py --no-traffic --no-cpu-baseline --no-configs --steps 2 --warmup 1 > $O/pmc$i.json 2> $O/pmc$i.err
  (cd $R && python tools/prof_summary.py $O/pmc$i > $O/pmc${i}_full.txt 2>&1; awk '/^# PMC/{p=1} p' $O/pmc${i}_full.txt | grep -A 9 "atb_f16_fit_kernel\|atb_f16_kernel<2, true\|axb_f16_kernel<4\|axb_f16_dma_kernel\|axb_bsplit\|^# PMC" > $O/pmc${i}_summary.txt)
  rm -rf $O/pmc$i $O/pmc${i}_full.txt
done
# config 5 under the kernel trace: every kernel of one call -- the operator route (the one HilbertEOF and the bench take) ...
timeout 600 rocprofv3 --kernel-trace --stats -d $O/c5o -o p --output-format csv -- python $R/tools/c5_operator_probe.py > $O/c5_operator_probe_under_rocprof.txt 2>&1
(cd $R && python tools/prof_summary.py $O/c5o > $O/c5_operator_kernel_trace_summary.txt 2>&1; python tools/trace_gaps.py $O/c5o colstats_tr_kernel > $O/c5_operator_timeline.txt 2>&1)
rm -rf $O/c5o
# ... and the two-part route (Im written, both parts streamed)
ENGINE_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/c5 -o p --output-format csv -- python $R/tools/complex_probe.py 8000 720 1440 20 > $O/complex_probe_under_rocprof.txt 2>&1
(cd $R && python tools/prof_summary.py $O/c5 > $O/complex_probe_kernel_trace_summary.txt 2>&1)
rm -rf $O/c5
cd $R
python tools/c5_operator_probe.py > $O/c5_operator_probe.txt 2>&1
python tools/fuzz_operator.py 3 60 > $O/fuzz_operator.txt 2>&1
python tools/cca_probe.py > $O/cca_probe.txt 2>&1
python tools/wide_small_probe.py > $O/wide_small_probe.txt 2>&1
ENGINE_ONLY=1 EOFX_C64_TRACE=1 python tools/complex_probe.py 8000 720 1440 20 > $O/complex_probe.txt 2>&1
NO_RAWT=1 ENGINE_ONLY=1 python tools/complex_probe.py 8000 720 1440 20 > $O/complex_probe_without_raw_layout.txt 2>&1
EOFX_C64_KRYLOV=0 ENGINE_ONLY=1 python tools/complex_probe.py 8000 720 1440 20 > $O/complex_probe_subspace_iteration.txt 2>&1
python tools/krylov_probe.py 2000 40 80 20 1000 260 500 20 600 30 40 12 > $O/krylov_probe.txt 2>&1
python tools/fuzz_complex.py 1 60 > $O/fuzz_complex_auto.txt 2>&1
BULK=1 python tools/fuzz_complex.py 2 60 > $O/fuzz_complex_bulk_converge.txt 2>&1
python tools/fuzz_parity.py 5 60 > $O/fuzz_parity.txt 2>&1
python tools/fuzz_fit.py 5 60 > $O/fuzz_fit.txt 2>&1
python tools/fuzz_cpcca.py 5 30 > $O/fuzz_cpcca.txt 2>&1
(for t in null_mode_probe scale_probe model_scale_probe edge_shape_probe model_edge_probe; do echo "## tools/$t.py"; python tools/$t.py 2>&1 | grep -v amdgpu; done) > $O/robustness_probes.txt
python tools/mca_default_probe.py > $O/mca_default_probe.txt 2>&1
python tools/hosteig_probe.py > $O/hosteig_probe.txt 2>&1
python bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 > $O/eighth.json 2> $O/eighth.err
python bench.py --force-sharded --no-traffic --no-cpu-baseline --no-configs --steps 5 --warmup 2 > $O/force_sharded_world1.json 2> $O/force_sharded.err
ls -la $O
