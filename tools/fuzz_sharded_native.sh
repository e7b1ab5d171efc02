#!/bin/bash
# 2-rank (same GPU, gloo host-callback binding) runs of tools/sharded_native_worker.py on a few shapes: the engine-owned sharded
# entries (EOF -- also land-masked --, MCA, Hilbert EOF on both routes) and the panel-level operator route against the single-GPU
# entries.  Slice widths divisible by 4 keep the EOF on eofx_fit_sharded_f32 ("native": [true, ...]); odd ones vote it to the
# panel-level driver.  Usage: bash tools/fuzz_sharded_native.sh
cd "$(dirname "$0")/.."
port=29600
for args in "--nsamples 130 --p1 776 --p2 1304" "--nsamples 513 --p1 9000 --p2 640 --mask" "--nsamples 257 --p1 3336 --p2 2112 --lowrank --modes 7" \
            "--nsamples 90 --p1 70000 --p2 304 --modes 6" "--nsamples 1025 --p1 2056 --p2 20000 --mask --modes 12" "--nsamples 64 --p1 136 --p2 136 --modes 5" \
            "--nsamples 2049 --p1 16384 --p2 8192 --modes 20" "--nsamples 777 --p1 4001 --p2 1303 --mask"; do
  port=$((port + 1))
  out=$(MASTER_ADDR=127.0.0.1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port $port tools/sharded_native_worker.py --backend gloo --same-gpu $args 2>/tmp/native_fuzz.err | grep '^{' | tail -1)
  [ -z "$out" ] && out="FAILED: $(grep -v '^W0\|Gloo\|^$' /tmp/native_fuzz.err | grep -i 'error\|Traceback' -A3 | tail -8 | tr '\n' ' ')"
  echo "$args -> $out"
done
