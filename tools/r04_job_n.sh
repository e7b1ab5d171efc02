#!/bin/bash
# round 4 job N: the randomized PCA route (blocked device Cholesky-QR) -- tests, then the default-argument MCA at config 3
mkdir -p gpurun_out/r04n
python -m pytest tests/test_gpu_pca.py tests/test_gpu_golden.py tests/test_gpu_cpcca.py tests/test_gpu_models.py tests/test_gpu_complex_cross.py -x -q -m gpu > gpurun_out/r04n/tests.txt 2>&1
tail -4 gpurun_out/r04n/tests.txt; grep -n "^E " gpurun_out/r04n/tests.txt | head
CPU=0 python tools/pca_probe.py > gpurun_out/r04n/pca_probe.txt 2>&1
grep -v Warning gpurun_out/r04n/pca_probe.txt | tail -16
