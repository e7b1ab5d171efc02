#!/bin/bash
# round 6, job a: new sharded entries (tests, bench legs on 1 / 2 ranks of one GPU), the fence, the PCA A/B test, R9 evidence
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_sharded_native.py "tests/test_gpu_pca.py::test_basis_only_fast_path_against_the_eigh_route" -x -q 2>&1 | tail -25 > gpurun_out/r06_t2.txt
cat gpurun_out/r06_t2.txt
for c in 3 5; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2955$c bench.py --gpus 2 --backend gloo --same-gpu --config $c --steps 2 --warmup 1 2>gpurun_out/r06_cfg${c}_2rank.err | tail -1 > gpurun_out/r06_cfg${c}_2rank.json
  tail -3 gpurun_out/r06_cfg${c}_2rank.err; cut -c1-1500 gpurun_out/r06_cfg${c}_2rank.json
  python bench.py --config $c --steps 3 --warmup 1 2>gpurun_out/r06_cfg${c}_1rank.err | tail -1 > gpurun_out/r06_cfg${c}_1rank.json
  tail -3 gpurun_out/r06_cfg${c}_1rank.err; cut -c1-1500 gpurun_out/r06_cfg${c}_1rank.json
  python bench.py --config $c --steps 3 --warmup 1 --force-sharded 2>gpurun_out/r06_cfg${c}_w1.err | tail -1 > gpurun_out/r06_cfg${c}_w1.json
  tail -3 gpurun_out/r06_cfg${c}_w1.err; cut -c1-1500 gpurun_out/r06_cfg${c}_w1.json
done
timeout 1500 python tools/r9_evidence.py --json gpurun_out/r06_r9_evidence.json > gpurun_out/r06_r9_evidence.txt 2> gpurun_out/r06_r9_evidence.err
tail -5 gpurun_out/r06_r9_evidence.err; cat gpurun_out/r06_r9_evidence.txt
