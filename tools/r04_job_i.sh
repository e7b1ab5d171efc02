#!/bin/bash
# round 4 job I: convergence trace of the complex rSVD at config-5 size
mkdir -p gpurun_out/r04i
EOFX_C64_TRACE=1 python tools/complex_probe.py 8000 720 1440 20 > gpurun_out/r04i/complex_probe.txt 2>&1
grep -v "^$" gpurun_out/r04i/complex_probe.txt | tail -40
