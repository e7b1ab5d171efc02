#!/bin/bash
# round 4 job C: Gram route with the lazily joined sketch -- cross tests, then config-3 timing in place
mkdir -p gpurun_out/r04c
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_cpcca.py tests/test_gpu_models.py tests/test_gpu_fullsize.py -x -q -m gpu -k "cross or mca or cpcca or g5 or MCA or config3 or rsvd_vs_oracle or peaked" > gpurun_out/r04c/tests_cross.txt 2>&1
tail -4 gpurun_out/r04c/tests_cross.txt
python tools/mca_probe.py > gpurun_out/r04c/mca_probe_inplace.txt 2>&1
tail -6 gpurun_out/r04c/mca_probe_inplace.txt
