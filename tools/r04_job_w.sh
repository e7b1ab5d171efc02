#!/bin/bash
# round 4: randomised parity sweeps with the LDS-DMA X Y kernel forced on (EOFX_AXB_DMA=1): small random shapes, masks, weights
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04w; rm -rf $O; mkdir -p $O
export EOFX_AXB_DMA=1
for s in 11 12; do timeout 900 python tools/fuzz_fit.py $s 120 > $O/fuzz_fit_$s.txt 2>&1; tail -3 $O/fuzz_fit_$s.txt; done
timeout 900 python tools/fuzz_parity.py 21 80 > $O/fuzz_parity.txt 2>&1; tail -3 $O/fuzz_parity.txt
timeout 900 python tools/fuzz_mca.py 31 60 > $O/fuzz_mca.txt 2>&1; tail -3 $O/fuzz_mca.txt
timeout 900 python tools/fuzz_complex.py 41 40 > $O/fuzz_complex.txt 2>&1; tail -3 $O/fuzz_complex.txt
