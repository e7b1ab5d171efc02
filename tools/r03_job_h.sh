#!/bin/bash
# round 3, GPU job H: the lean complex layout (Re in place, Im^T only): complex tests, model tests, config-5 full size,
# complex probe in both layouts, rotation probe trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_complex.py -x -q -m gpu > $O/complex_tests.log 2>&1; echo "complex tests rc=$?" | tee $O/summary.txt; tail -15 $O/complex_tests.log
timeout 1500 python -m pytest tests/test_gpu_complex_cross.py tests/test_gpu_models.py tests/test_gpu_rotation.py -x -q -m gpu > $O/model_tests.log 2>&1; echo "model tests rc=$?" | tee -a $O/summary.txt; tail -5 $O/model_tests.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5" > $O/config5_test.log 2>&1; echo "config5 test rc=$?" | tee -a $O/summary.txt; tail -5 $O/config5_test.log
ENGINE_ONLY=1 timeout 600 python tools/complex_probe.py 8000 720 1440 20 > $O/complex_k20_lean.txt 2>&1; tail -2 $O/complex_k20_lean.txt
LAYOUT=written ENGINE_ONLY=1 timeout 600 python tools/complex_probe.py 8000 720 1440 20 > $O/complex_k20_written.txt 2>&1; tail -2 $O/complex_k20_written.txt
ENGINE_ONLY=1 timeout 600 python tools/complex_probe.py 8000 720 1440 40 > $O/complex_k40_lean.txt 2>&1; tail -2 $O/complex_k40_lean.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $O/rot_prof -o rot --output-format csv -- python tools/rot_probe.py 1036800 200 steps > $O/rot_prof.log 2>&1
python tools/prof_summary.py $O/rot_prof > $O/rot_probe_kernel_trace_summary.txt 2>&1; head -8 $O/rot_probe_kernel_trace_summary.txt | cut -c1-160
rm -rf $O/rot_prof
