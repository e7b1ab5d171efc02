cd $GRAFT_REPO_ROOT; O=gpurun_out/j17; mkdir -p $O
for t in "fuzz_parity 11 80" "fuzz_fit 12 80" "fuzz_cpcca 13 40" "fuzz_complex 14 60" "fuzz_operator 15 80" "fuzz_hilbert 16 50" "fuzz_mca 17 40" "fuzz_rotator 18 30" "fuzz_large 19 6"; do set -- $t
  timeout 600 python tools/$1.py $2 $3 > $O/$1.txt 2>&1; echo "$1 $2 $3: $(grep -i "cases\|bad" $O/$1.txt | tail -1)"; grep -i "MISMATCH\|EXC" $O/$1.txt | head -3
done
BULK=1 timeout 600 python tools/fuzz_complex.py 21 60 > $O/fuzz_complex_bulk.txt 2>&1; echo "fuzz_complex BULK 21 60: $(grep -i cases $O/fuzz_complex_bulk.txt | tail -1)"; grep -i "MISMATCH\|EXC" $O/fuzz_complex_bulk.txt | head -3
