#!/bin/bash
# round 3, GPU job L: complex cross diagnostics / inverse_transform, complex models after the lean-layout gating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_complex_cross.py tests/test_gpu_complex.py tests/test_gpu_models.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -25 $O/tests.log
