#!/bin/bash
# round 4: panel_matmul's wide route on the matrix cores: tests, then the PCA probe and the default-argument MCA
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04y; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_gram.py tests/test_gpu_pca.py tests/test_gpu_cpcca.py tests/test_gpu_models.py -x -q > $O/pytest.txt 2>&1; grep -h "passed\|failed\|^E " $O/pytest.txt | tail -6
python tools/pca_probe.py > $O/pca_probe.txt 2>&1; grep -A18 "^rep 1" $O/pca_probe.txt
EOFX_NO_MATMUL_NT=1 python tools/pca_probe.py > $O/pca_probe_old.txt 2>&1; grep -A4 "^rep 1" $O/pca_probe_old.txt | grep "fit\|TOTAL"
grep "TOTAL" $O/pca_probe_old.txt | tail -1
