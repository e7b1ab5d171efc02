#!/bin/bash
# round 4: sketch workers pinned to the caller's L3 domain: the generator alone, then the fits that wait for it
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04y; rm -rf $O; mkdir -p $O
for pin in 0 1; do EOFX_SKETCH_PIN=$pin python tools/sketch_pin_probe.py 2>/dev/null | grep "default"; done | tee $O/sketch_pin.txt
for i in 1 2; do for pin in 0 1; do
  EOFX_SKETCH_PIN=$pin python bench.py --nlon 180 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eighth pin=$pin', d['ms_per_step'])"
  EOFX_SKETCH_PIN=$pin python bench.py --nsamples 5000 --nlat 360 --nlon 720 --no-traffic --no-cpu-baseline --no-configs --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config2 pin=$pin', d['ms_per_step'])"
  EOFX_SKETCH_PIN=$pin python bench.py --no-traffic --no-cpu-baseline --no-configs --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config4 pin=$pin', d['ms_per_step'])"
done; done | tee -a $O/sketch_pin.txt
