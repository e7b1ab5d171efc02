#!/bin/bash
# round 4: the first field's Rayleigh-Ritz eigh on a second stream under the second field's PCA kernels: tests, CCA-family probe
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04y; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_pca.py tests/test_gpu_cpcca.py tests/test_gpu_models.py tests/test_gpu_complex_cross.py tests/test_gpu_golden.py tests/test_gpu_rotation.py -x -q > $O/pytest.txt 2>&1; grep -h "passed\|failed\|^E " $O/pytest.txt | tail -6
python tools/cca_probe.py > $O/cca_probe.txt 2>&1; grep "fit " $O/cca_probe.txt
python tools/cca_probe.py > $O/cca_probe2.txt 2>&1; grep "fit " $O/cca_probe2.txt
