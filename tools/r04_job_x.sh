#!/bin/bash
# round 4: the randomised sweeps on the final tree (default kernel selection)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04x; rm -rf $O; mkdir -p $O
timeout 900 python tools/fuzz_fit.py 111 100 > $O/fuzz_fit.txt 2>&1; tail -1 $O/fuzz_fit.txt
timeout 900 python tools/fuzz_parity.py 121 60 > $O/fuzz_parity.txt 2>&1; tail -1 $O/fuzz_parity.txt
timeout 900 python tools/fuzz_mca.py 131 50 > $O/fuzz_mca.txt 2>&1; tail -1 $O/fuzz_mca.txt
timeout 900 python tools/fuzz_cpcca.py 151 40 > $O/fuzz_cpcca.txt 2>&1; tail -1 $O/fuzz_cpcca.txt
timeout 900 python tools/fuzz_complex.py 141 30 > $O/fuzz_complex.txt 2>&1; tail -1 $O/fuzz_complex.txt
timeout 900 python tools/fuzz_rotator.py 161 30 > $O/fuzz_rotator.txt 2>&1; tail -1 $O/fuzz_rotator.txt
timeout 900 python tools/fuzz_hilbert.py 171 40 > $O/fuzz_hilbert.txt 2>&1; tail -1 $O/fuzz_hilbert.txt
timeout 900 bash tools/fuzz_sharded.sh > $O/fuzz_sharded.txt 2>&1; cut -c1-150 $O/fuzz_sharded.txt | tail -6
