#!/bin/bash
# round 6, job f: soak of default-argument cross models (bitwise), sharded fuzz on the panel-level and the engine-owned entries
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1200 python tools/soak_cross_default.py 400 2>&1 | grep -v Warning | tail -12 | tee gpurun_out/r06_soak_cross_default.txt
bash tools/fuzz_sharded.sh 2>&1 | tail -8 | tee gpurun_out/r06_fuzz_sharded.txt
port=29560
for args in "--nsamples 130 --p1 777 --p2 1300" "--nsamples 513 --p1 9001 --p2 640 --mask" "--nsamples 257 --p1 3333 --p2 2111 --lowrank --modes 7" \
            "--nsamples 90 --p1 70001 --p2 300 --modes 6" "--nsamples 1025 --p1 2051 --p2 20000 --mask --modes 12" "--nsamples 64 --p1 129 --p2 131 --modes 5"; do
  port=$((port + 1))
  out=$(MASTER_ADDR=127.0.0.1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port $port tools/sharded_native_worker.py --backend gloo --same-gpu $args 2>gpurun_out/native_fuzz.err | grep '^{' | tail -1)
  [ -z "$out" ] && out="FAILED: $(grep -v '^W0\|Gloo\|^$' gpurun_out/native_fuzz.err | grep -i 'error\|Traceback' -A3 | tail -8 | tr '\n' ' ')"
  echo "$args -> $out"
done 2>&1 | tee gpurun_out/r06_fuzz_sharded_native.txt | cut -c1-1500
