#!/bin/bash
# us per GD iteration of a 32-candidate refinement, fp16-level against RGBA8 texels, starting poses all over the room (what make_input
# hands over: spread 1) — the data behind ops.refine_texels.   bash tools/refine_fmt_sweep.sh
for cfg in "1024x2048 400000 700000 1000000 2000000" "2048x4096 2000000 3000000 4500000 6000000 8000000"; do
  set -- $cfg; hw=$1; shift
  for n in "$@"; do for f in f16 u8; do
    echo -n "$hw fmt $f: "; PCL_TOOL_HW=$hw PCL_PANO_FMT=$f python tools/iter_latency.py $n 32 6 1 2>&1 | tail -1 | sed 's/PCL_BLOCKS.*fused [01] |//'
  done; done
done
