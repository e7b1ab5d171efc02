#!/bin/bash
# 2-rank (same GPU, gloo) runs of tools/sharded_mca_worker.py on a few random shapes; prints the comparison against
# the single-GPU drivers.  Usage: bash tools/fuzz_sharded.sh
cd "$(dirname "$0")/.."
port=29530
for args in "--nsamples 130 --p1 777 --p2 1300" "--nsamples 513 --p1 9001 --p2 640 --nan" "--nsamples 257 --p1 3333 --p2 2111 --pca" \
            "--nsamples 90 --p1 70001 --p2 300" "--nsamples 1025 --p1 515 --p2 20000 --nan" "--nsamples 64 --p1 129 --p2 131 --modes 5"; do
  port=$((port + 1))
  out=$(MASTER_ADDR=127.0.0.1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port $port tools/sharded_mca_worker.py --backend gloo --same-gpu $args 2>/dev/null | grep '^{' | tail -1)
  echo "$args -> $out"
done
