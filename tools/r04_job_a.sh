#!/bin/bash
# round 4 job A: the Gram route of the cross-covariance path -- parity tests, then timing at config-3 size
mkdir -p gpurun_out/r04a
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_cpcca.py tests/test_gpu_models.py -x -q -m gpu -k "cross or mca or cpcca or g5 or MCA" > gpurun_out/r04a/tests_cross.txt 2>&1
tail -5 gpurun_out/r04a/tests_cross.txt
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config3 or mca" > gpurun_out/r04a/tests_full.txt 2>&1
tail -5 gpurun_out/r04a/tests_full.txt
python tools/mca_probe.py > gpurun_out/r04a/mca_probe.txt 2>&1
tail -8 gpurun_out/r04a/mca_probe.txt
