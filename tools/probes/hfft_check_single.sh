#!/bin/bash
# the one-feature-per-workgroup mode of the fused Hilbert kernel (half-length complex transform): every plan, then timing
B=build/hfft_probe
for n in 513 777 1024 1500 2048 3001 4096 5000 8191 8192 8193 10000 12345 16383 16384; do
  for pad in 1 0; do
    timeout 300 $B $n 37 $pad 1 1 1 | tail -1 | cut -c1-120 | sed "s/^/n=$n pad=$pad: /"
  done
done
timeout 300 $B 10000 1036800 1 0 2 1 | tail -2
timeout 300 $B 16384 400000 1 0 2 1 | tail -2
timeout 300 $B 8000 1036800 1 0 2 1 | tail -2
