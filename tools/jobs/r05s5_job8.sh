cd $GRAFT_REPO_ROOT; O=gpurun_out/j8; mkdir -p $O
timeout 900 python tools/fuzz_operator.py 1 80 > $O/fuzz_operator_1.txt 2>&1
timeout 900 python tools/fuzz_operator.py 2 80 > $O/fuzz_operator_2.txt 2>&1
tail -12 $O/fuzz_operator_1.txt; tail -12 $O/fuzz_operator_2.txt
