#!/bin/bash
# round 4 job O: device-side whitening / resident scores -- tests of the cross family, then CCA / RDA at config-3 size
# (build/r03 = a worktree of the round-3 head with its own library, when present: the "before" column)
mkdir -p gpurun_out/r04o
python -m pytest tests/test_gpu_pca.py tests/test_gpu_cpcca.py -x -q -m gpu > gpurun_out/r04o/tests.txt 2>&1
tail -3 gpurun_out/r04o/tests.txt; grep -n "^E " gpurun_out/r04o/tests.txt | head
python tools/cca_probe.py > gpurun_out/r04o/cca_probe.txt 2>&1
tail -4 gpurun_out/r04o/cca_probe.txt

