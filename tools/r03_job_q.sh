#!/bin/bash
# round 3, GPU job Q: the multi-rank code paths after this round's changes -- bench.py with 2 ranks on one GPU (gloo carries the
# collectives, the kernels are the real ones), the sharded MCA / EOF workers, the 2-rank tests of the full-size suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
MASTER_ADDR=127.0.0.1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
   bench.py --gpus 2 --same-gpu --backend gloo --steps 3 --warmup 1 --nlon 360 --no-cpu-baseline --no-configs > $O/bench_2rank_same_gpu.json 2> $O/bench_2rank.err
echo "2-rank bench rc=$?" | tee $O/summary.txt
python -c "
import json;d=json.loads(open('$O/bench_2rank_same_gpu.json').read().strip().splitlines()[-1]);print(d['n_gpus'], d['ms_per_step'], d['config'].get('entry'), d['comm'], d['parity'])" 2>&1 | cut -c1-600
timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 --nlon 360 --no-cpu-baseline --no-configs --no-traffic > $O/bench_1rank_quarter.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/bench_1rank_quarter.json').read().strip().splitlines()[-1]);print(d['n_gpus'], d['ms_per_step'], d['parity'])" 2>&1 | cut -c1-400
bash tools/fuzz_sharded.sh 2>&1 | tee $O/fuzz_sharded.txt | cut -c1-250
