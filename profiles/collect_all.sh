#!/bin/bash
# Everything profiles/rNN/ is built from, in one go on the GPU box:  bash profiles/collect_all.sh r02
R=${1:-r02}
PMC="--steps 8 --warmup 2 --no-cpu-baseline"
ROOF_KEY=cfg2/poses256/f16 POINT_POSES=256e6 ROOF_SOURCE=$R/a_default_cfg2_8images ROOF_CMD="bench.py (default)" \
  bash profiles/collect.sh ${R}_a --workload cfg2 $PMC
ROOF_KEY=cfg2/poses32/f16 POINT_POSES=32e6 ROOF_SOURCE=$R/b_cfg2_single_image ROOF_CMD="bench.py --images-per-launch 1" \
  bash profiles/collect.sh ${R}_b --workload cfg2 --images-per-launch 1 $PMC
ROOF_KEY=cfg5/poses64/f16 POINT_POSES=640e6 ROOF_SOURCE=$R/c2_cfg5_2images ROOF_CMD="bench.py --workload cfg5 --steps 4" \
  bash profiles/collect.sh ${R}_c --workload cfg5 --steps 4 --warmup 1 --no-cpu-baseline
ROOF_KEY=cfg3/poses256/f16 POINT_POSES=256e6 ROOF_SOURCE=$R/d_cfg3 ROOF_CMD="bench.py --workload cfg3 --steps 2" \
  bash profiles/collect.sh ${R}_d --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline
ROOF_KEY=shipped/poses6/u8 POINT_POSES=1000002 ROOF_SOURCE=$R/h_shipped_1image ROOF_CMD="bench.py --workload shipped --images-per-launch 1" \
  bash profiles/collect.sh ${R}_h --workload shipped --images-per-launch 1 $PMC
ROOF_KEY=shipped/poses48/u8 POINT_POSES=8000016 ROOF_SOURCE=$R/i_shipped_8images ROOF_CMD="bench.py --workload shipped --images-per-launch 8" \
  bash profiles/collect.sh ${R}_i --workload shipped --images-per-launch 8 $PMC
PIPELINE_TAG=pipeline_cfg2 ROOF_SOURCE=$R/e_init_stage ROOF_CMD="tools/init_bench.py" bash profiles/collect.sh ${R}_e --script tools/init_bench.py
bash profiles/collect.sh ${R}_f --script tools/chain_bench.py 167000 6 1
PIPELINE_TAG=pipeline_shipped ROOF_SOURCE=$R/p_pipeline_shipped ROOF_CMD="tools/pipeline_trace.py 8" bash profiles/collect.sh ${R}_p --script tools/pipeline_trace.py 8
PIPELINE_TAG=depth_mask_cfg2 ROOF_SOURCE=$R/z_depth_mask_cfg2 ROOF_CMD="tools/dgd_profile.py 32 20" bash profiles/collect.sh ${R}_z --script tools/dgd_profile.py 32 20
PIPELINE_TAG=depth_mask_cfg2_every_point ROOF_SOURCE=$R/z1_depth_mask_cfg2_stride1 ROOF_CMD="tools/dgd_profile.py 32 20 1" bash profiles/collect.sh ${R}_z1 --script tools/dgd_profile.py 32 20 1
# the driver's own command line (--steps 20 --warmup 5): 4 launch groups of 5 images = 160 poses per launch
ROOF_KEY=cfg2/poses160/f16 POINT_POSES=160e6 ROOF_SOURCE=$R/g_driver_cfg2_5images ROOF_CMD="bench.py --steps 20 --warmup 5" \
  bash profiles/collect.sh ${R}_g --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline
# the trim launch alone, with the counters: plain (chunk, slot) order against the row-sorted work list (round 6), cfg-2 size and the shipped shape
bash profiles/collect.sh ${R}_t1 --script tools/trim_pmc.py 1000000 plain
bash profiles/collect.sh ${R}_t2 --script tools/trim_pmc.py 1000000 order
bash profiles/collect.sh ${R}_t3 --script tools/trim_pmc.py 166667 plain
bash profiles/collect.sh ${R}_t4 --script tools/trim_pmc.py 166667 order
# issue cost of the VALU instruction classes, measured in the same collection (tools/roof_mix.py prices the loss kernel's mix with it)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o gpurun_out/valu_rate tools/micro/valu_rate.hip && ./gpurun_out/valu_rate > gpurun_out/valu_rate.txt 2>&1; rm -f gpurun_out/valu_rate
python3 profiles/merge_roofs.py gpurun_out/prof_${R}_*/summary/roofs.json gpurun_out/prof_${R}_*/summary/pipeline_roofs.json
