#!/bin/bash
mkdir -p gpurun_out/r04k
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "two_rank_sharded_path" > gpurun_out/r04k/tests2.txt 2>&1
tail -4 gpurun_out/r04k/tests2.txt
grep -n "^E " gpurun_out/r04k/tests2.txt | head
for mode in "" "--force-sharded" "--force-sharded --no-native"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --nlon 180 $mode > gpurun_out/r04k/b.json 2> gpurun_out/r04k/b.err || tail -5 gpurun_out/r04k/b.err
  python - "$mode" <<PY
import json, sys
d = json.loads(open("gpurun_out/r04k/b.json").read().strip().splitlines()[-1])
print("eighth", repr(sys.argv[1]), d["ms_per_step"], d["config"]["entry"][:40], d.get("comm"))
PY
done
