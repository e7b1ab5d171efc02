#!/bin/bash
# round 4: the remaining randomised sweeps on the final tree (CPCCA family, rotators, Hilbert stage, large shapes, two-rank sharded)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04x; rm -rf $O; mkdir -p $O
timeout 900 python tools/fuzz_cpcca.py 51 40 > $O/fuzz_cpcca.txt 2>&1; tail -2 $O/fuzz_cpcca.txt
timeout 900 python tools/fuzz_rotator.py 61 40 > $O/fuzz_rotator.txt 2>&1; tail -2 $O/fuzz_rotator.txt
timeout 900 python tools/fuzz_hilbert.py 71 60 > $O/fuzz_hilbert.txt 2>&1; tail -2 $O/fuzz_hilbert.txt
timeout 1200 python tools/fuzz_large.py 81 12 > $O/fuzz_large.txt 2>&1; tail -2 $O/fuzz_large.txt
timeout 900 bash tools/fuzz_sharded.sh > $O/fuzz_sharded.txt 2>&1; tail -4 $O/fuzz_sharded.txt
