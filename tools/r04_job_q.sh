#!/bin/bash
# round 4: what does the B staging of axb_f16 cost -- its arithmetic or its loads / stores?  (one box, three repetitions)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04q; mkdir -p $O
for i in 1 2 3; do ./build/axb_probe | grep "DBG  0\|DBG 16\|DBG 20\|DBG 32\|DBG  7"; echo; done > $O/axb_bcost.txt 2>&1
cat $O/axb_bcost.txt
