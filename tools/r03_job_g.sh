#!/bin/bash
# round 3, GPU job G: the column-blocked rotation step (128 / 256-wide panels), the non-default-stream test, the
# config-5 full-size test through eofx_rsvd_c64, rotation probe with and without rocprofv3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_rotation.py -x -q -m gpu > $O/rotation_tests.log 2>&1; echo "rotation tests rc=$?" | tee $O/summary.txt; tail -5 $O/rotation_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "non_default_stream" > $O/stream_test.log 2>&1; echo "stream test rc=$?" | tee -a $O/summary.txt; tail -3 $O/stream_test.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5" > $O/config5_test.log 2>&1; echo "config5 test rc=$?" | tee -a $O/summary.txt; tail -3 $O/config5_test.log
for m in 100 200; do
  timeout 600 python tools/rot_probe.py 1036800 $m steps > $O/rot_probe_m$m.txt 2>&1; tail -3 $O/rot_probe_m$m.txt
done
timeout 900 rocprofv3 --kernel-trace --stats -d $O/rot_prof -o rot --output-format csv -- python tools/rot_probe.py 1036800 200 steps > $O/rot_prof.log 2>&1
python tools/prof_summary.py $O/rot_prof > $O/rot_probe_kernel_trace_summary.txt 2>&1; head -12 $O/rot_probe_kernel_trace_summary.txt
