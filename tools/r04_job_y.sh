#!/bin/bash
# round 4: the lean Hilbert stage, copy chunks and transforms on two streams: tests, then config 5 (probe) with and without
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04y; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_complex.py tests/test_gpu_complex_cross.py tests/test_gpu_golden.py -x -q > $O/pytest.txt 2>&1; grep -h "passed\|failed\|^E " $O/pytest.txt | tail -6
python tools/complex_probe.py 8000 720 1440 20 > $O/cplx.txt 2>&1; grep "^rep1" $O/cplx.txt | cut -c1-200
EOFX_HILBERT_UNPIPED=1 python tools/complex_probe.py 8000 720 1440 20 > $O/cplx_unpiped.txt 2>&1; grep "^rep1" $O/cplx_unpiped.txt | cut -c1-200
