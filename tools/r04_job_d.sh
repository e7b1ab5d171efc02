#!/bin/bash
# round 4 job D: the whole GPU suite after the Gram route / in-place cross models / mask-aware PCA
mkdir -p gpurun_out/r04d
python -m pytest tests -x -q -m gpu > gpurun_out/r04d/tests_gpu.txt 2>&1
tail -15 gpurun_out/r04d/tests_gpu.txt
