#!/bin/bash
# round 4 job K: the engine's own sharded entry -- world-1 RCCL / callback bitwise tests, 2 ranks on one GPU through bench.py,
# then bench.py --force-sharded (one rank, RCCL really called) against the plain single-GPU line
mkdir -p gpurun_out/r04k
python -m pytest tests/test_gpu_sharded_native.py tests/test_gpu_fullsize.py -x -q -m gpu -k "native or world1 or two_rank_sharded_path or needs_a_comm or vote" > gpurun_out/r04k/tests.txt 2>&1
tail -5 gpurun_out/r04k/tests.txt
for mode in "" "--force-sharded" "--force-sharded --no-native"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic $mode > gpurun_out/r04k/b.json 2> gpurun_out/r04k/b.err || tail -5 gpurun_out/r04k/b.err
  python - "$mode" <<PY
import json, sys
d = json.loads(open("gpurun_out/r04k/b.json").read().strip().splitlines()[-1])
print(repr(sys.argv[1]), d["ms_per_step"], d["config"]["entry"][:60], d.get("comm"), d["parity"]["s_head"])
PY
done
for mode in "" "--force-sharded" "--force-sharded --no-native"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --nlon 180 $mode > gpurun_out/r04k/b.json 2> gpurun_out/r04k/b.err || tail -5 gpurun_out/r04k/b.err
  python - "$mode" <<PY
import json, sys
d = json.loads(open("gpurun_out/r04k/b.json").read().strip().splitlines()[-1])
print("eighth", repr(sys.argv[1]), d["ms_per_step"], d["config"]["entry"][:60], d.get("comm"))
PY
done
