#!/bin/bash
# round 4: the LDS-DMA variant of the in-place X Y product (EOFX_AXB_DMA=1) -- tests over it, then the fit with and without, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04u; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
EOFX_AXB_DMA=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_models.py tests/test_gpu_complex.py tests/test_gpu_fullsize.py tests/test_gpu_sharded_native.py -x -q > $O/pytest_dma.txt 2>&1
grep -h "passed\|failed" $O/pytest_dma.txt | tail -2
for i in 1 2 3; do
  for m in 0 1; do
    EOFX_AXB_DMA=$m python bench.py --no-traffic --no-cpu-baseline --no-configs --steps 10 --warmup 3 > $O/bench_${m}_$i.json 2> $O/bench_${m}_$i.err
    python - <<PY
import json
d=json.load(open("$O/bench_${m}_$i.json"))
print("dma=$m rep $i", d['ms_per_step'], {k.split()[0]:round(v['mean_launch_ms'],3) for k,v in d['roofline']['by_kernel'].items()})
PY
  done
done | tee $O/ab.txt
EOFX_AXB_DMA=1 python bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 > $O/eighth_1.json 2>/dev/null
EOFX_AXB_DMA=0 python bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 > $O/eighth_0.json 2>/dev/null
python -c "
import json
for m in (0,1): print('eighth dma=%d'%m, json.load(open('$O/eighth_%d.json'%m))['ms_per_step'])" | tee -a $O/ab.txt
