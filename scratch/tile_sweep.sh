#!/bin/bash
for S in 88 48 84 44; do
  FDTD_TILE_SHAPE=$S python bench.py --steps 10 --warmup 2 --no-cpu-baseline --kernel tile 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('NS tile $S', d['value'], d['roofline']['ms_update_E'])"
done
for S in 88 48 44; do
  FDTD_TILE_SHAPE=$S python bench.py --steps 6 --warmup 2 --no-cpu-baseline --kernel tile --workload C3 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 tile $S', d['value'], d['roofline']['ms_update_E'])"
done
