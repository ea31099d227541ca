#!/bin/bash
# us per GD iteration under the launch-planning knobs at shapes other than the two the defaults were tuned on.   bash tools/plan_sweep.sh
for shape in "166667 6" "166667 32" "166667 48" "100000 2" "100000 8" "1000000 6" "1000000 16" "1000000 32"; do
  for sp in 1 0; do for v in "X=0" "PCL_G=1"; do
    echo -n "$v: "; env $v python tools/iter_latency.py $shape 6 $sp 2>&1 | tail -1 | sed 's/| PCL_BLOCKS=[A-Za-z0-9]* FUSE=[A-Za-z0-9]* |//'
  done; done
done
