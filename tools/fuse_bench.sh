#!/bin/bash
# one launch per GD iteration (fused prologue) against the two-launch form at the latency-bound shapes:  bash tools/fuse_bench.sh
for f in 1024 0; do
  for w in shipped cfg1; do
    PCL_GD_FUSE_BLOCKS=$f python bench.py --workload $w --images-per-launch 1 --steps 8 --no-cpu-baseline --no-also --no-single-image 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse<=$f $w: %.1f cand-poses/s, %.2f us per iteration, loss launch %.2f us' % (d['value'], d['ms_per_step']*10, d['roofline']['avg_launch_ms']*1e3))"
  done
done
