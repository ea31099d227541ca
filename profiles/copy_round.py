#!/usr/bin/env python3
"""After `bash profiles/collect_all.sh rNN` on the GPU box: copy the summaries gpurun merged back under gpurun_out/ into profiles/rNN/ under
their long names, install the merged roofs files, and price the loss kernel's instruction mix (tools/roof_mix.py).   python3 profiles/copy_round.py r06"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
NAMES = {"a": "a_default_cfg2_8images", "b": "b_cfg2_single_image", "c": "c2_cfg5_2images", "d": "d_cfg3_256poses", "e": "e_init_stage",
         "f": "f_shipped_167k_6cand_chain_bench", "g": "g_driver_cfg2_5images", "h": "h_shipped_1image_fused", "i": "i_shipped_8images",
         "p": "p_pipeline_shipped", "z": "z_depth_mask_cfg2", "z1": "z1_depth_mask_cfg2_stride1", "t1": "t1_trim_1M_plain_order",
         "t2": "t2_trim_1M_work_list", "t3": "t3_trim_167k_plain_order", "t4": "t4_trim_167k_work_list"}
RAW_COUNTERS = ("a", "b", "g", "e", "z", "z1", "t1", "t2", "t3", "t4")     # the raw counter dumps of the headline, pipeline, depth and trim runs only
dst = os.path.join(HERE, R)
os.makedirs(dst, exist_ok=True)
for tag, name in NAMES.items():
    src = os.path.join(REPO, "gpurun_out", "prof_%s_%s" % (R, tag))
    for f, suffix in (("summary/kernel_roofs.json", "kernel_roofs.json"), ("summary/kernel_stats.txt", "kernel_stats.txt"), ("summary/pmc.json", "pmc.json"),
                      ("kt.log", "command_output.txt")):
        p = os.path.join(src, f)
        if os.path.exists(p) and (suffix != "pmc.json" or tag in RAW_COUNTERS):
            shutil.copy(p, os.path.join(dst, "%s_%s" % (name, suffix)))
for f in ("roofs.json", "pipeline_roofs.json"):
    shutil.copy(os.path.join(REPO, "gpurun_out", f), os.path.join(HERE, f))
shutil.copy(os.path.join(REPO, "gpurun_out", "valu_rate.txt"), os.path.join(dst, "valu_rate.txt"))
subprocess.check_call([sys.executable, os.path.join(REPO, "tools", "roof_mix.py"), "--table", os.path.join(dst, "valu_rate.txt"), "--out", os.path.join(dst, "roof_mix.json")])
