#!/bin/bash
# correctness sweep of the fused Hilbert kernel over all plan variants, then timing at config-5 size
B=build/hfft_probe
for n in 33 100 240 512 513 777 1000 1024 2000 2048 4000 4096 8000 8192; do
  for pad in 1 0; do
    timeout 120 $B $n 1001 $pad 1 1 | tail -1 | sed "s/^/n=$n pad=$pad: /"
  done
done
timeout 60 $B 500 3 1 0 1 | tail -1
timeout 60 $B 500 1 1 1 1 | tail -1
timeout 300 $B 8000 1036800 1 0 3 | tail -2
timeout 300 $B 4000 1036800 1 0 3 | tail -2
timeout 300 $B 1000 1036800 1 0 3 | tail -2
