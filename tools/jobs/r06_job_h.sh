#!/bin/bash
# round 6, job h: the fuzz sweeps on the final tree with a third set of seeds (the complex models now default to the convergent rule)
cd "$(dirname "$0")/../.."; O=gpurun_out/r06_fuzz; mkdir -p $O
for t in "fuzz_parity 31 80" "fuzz_fit 32 80" "fuzz_cpcca 33 40" "fuzz_complex 34 60" "fuzz_operator 35 80" "fuzz_hilbert 36 50" "fuzz_mca 37 40" "fuzz_rotator 38 30" "fuzz_large 39 6"; do set -- $t
  timeout 900 python tools/$1.py $2 $3 > $O/$1.txt 2>&1; echo "$1 $2 $3: $(grep -i "cases\|bad" $O/$1.txt | tail -1)"; grep -i "MISMATCH\|EXC" $O/$1.txt | head -3
done 2>&1 | tee gpurun_out/r06_fuzz.txt
BULK=1 timeout 900 python tools/fuzz_complex.py 41 60 > $O/fuzz_complex_bulk.txt 2>&1; echo "fuzz_complex BULK=1 (n_iter=converge) 41 60: $(grep -i cases $O/fuzz_complex_bulk.txt | tail -1)" | tee -a gpurun_out/r06_fuzz.txt; grep -i "MISMATCH\|EXC" $O/fuzz_complex_bulk.txt | head -5 | tee -a gpurun_out/r06_fuzz.txt
tail -4 $O/fuzz_complex_bulk.txt | cut -c1-300
