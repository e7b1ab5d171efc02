#!/bin/bash
# round 4: the LDS-DMA X Y pass at one rank's share of an 8-GPU run (--nlon 180) and at config 2 size, against the register path, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04v; rm -rf $O; mkdir -p $O
for i in 1 2 3; do
  for m in 0 1; do
    EOFX_AXB_DMA=$m python bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 > $O/e_${m}_$i.json 2>/dev/null
    EOFX_AXB_DMA=$m python bench.py --nsamples 5000 --nlat 360 --nlon 720 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 > $O/c2_${m}_$i.json 2>/dev/null
    python - <<PY
import json
for f in ("e","c2"):
    d=json.load(open("$O/%s_${m}_$i.json"%f))
    print(f, "dma=$m rep $i", d['ms_per_step'], {k.split()[0]:round(v['mean_launch_ms'],4) for k,v in d['roofline']['by_kernel'].items()})
PY
  done
done | tee $O/ab_small.txt
