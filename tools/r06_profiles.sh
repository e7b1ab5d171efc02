#!/bin/bash
# round 6 evidence set on the final code: GPU suite, the driver command, rocprofv3 kernel trace + stats of the bench, PMC passes
# (HBM traffic, SQ counters) of the streaming kernels, one fit's timeline, config 5 on the convergent rule kernel by kernel, one
# rank's eighth share, the sharded entries on world-1 RCCL and on two ranks sharing the GPU, the robustness probes.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06final; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee $O/summary.txt; grep -h "passed\|failed" $O/pytest_gpu.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full_driver_command.json 2> $O/bench_full.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s" | tee -a $O/summary.txt
S=$(date +%s); python bench.py > $O/bench_default_flags.json 2> $O/bench_default.err; echo "bench (no flags) rc=$? wall=$(( $(date +%s) - S ))s" | tee -a $O/summary.txt
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
# config 5 on the convergent rule under the kernel trace
timeout 600 rocprofv3 --kernel-trace --stats -d $O/c5 -o p --output-format csv -- python $R/tools/c5_converge_probe.py > $O/c5_converge_probe_under_rocprof.txt 2>&1
(cd $R && python tools/prof_summary.py $O/c5 > $O/c5_converge_kernel_trace_summary.txt 2>&1; python tools/trace_gaps.py $O/c5 panel_import_kernel > $O/c5_converge_timeline.txt 2>&1)
rm -rf $O/c5
cd $R
python bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 > $O/eighth.json 2> $O/eighth.err
python bench.py --force-sharded --no-traffic --no-cpu-baseline --steps 5 --warmup 2 > $O/force_sharded_world1.json 2> $O/force_sharded.err
for c in 3 5; do
  python bench.py --config $c --steps 3 --warmup 1 > $O/bench_config${c}_1rank.json 2> $O/cfg.err
  python bench.py --config $c --steps 3 --warmup 1 --force-sharded > $O/bench_config${c}_w1.json 2>> $O/cfg.err
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2958$c bench.py --gpus 2 --backend gloo --same-gpu --config $c --steps 2 --warmup 1 2>> $O/cfg.err | tail -1 > $O/bench_config${c}_2rank.json
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29590 bench.py --gpus 2 --backend gloo --same-gpu --steps 3 --warmup 1 > $O/selflaunch_2ranks_same_gpu.json 2> $O/selflaunch.err
(for t in null_mode_probe scale_probe model_scale_probe edge_shape_probe model_edge_probe; do echo "## tools/$t.py"; timeout 300 python tools/$t.py 2>&1 | grep -v amdgpu; done) > $O/robustness_probes.txt
python tools/cca_probe.py > $O/cca_probe.txt 2>&1
for L in none inplace gram; do LOAD=$L timeout 300 python tools/thread_probe5.py 2>&1 | grep "^LOAD"; done > $O/thread_probe5_eigh_victim.txt
ls -la $O
cat $O/summary.txt
