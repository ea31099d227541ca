#!/usr/bin/env python3
"""Static instruction mix of one kernel of a hipcc -S listing: whole kernel and the blocks the assembler marks as loop
bodies.   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o loss.s piccolo_amd/csrc/pcl_loss.hip
          python tools/isa_stats.py loss.s _Z15pcl_loss_kernelILi2ELb1ELb0ELi2EEv11PclLossArgs [evals_per_loop_body]"""
import collections
import re
import sys

path, kernel = sys.argv[1], sys.argv[2]
per = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
in_loop = False
all_ops, loop_ops = collections.Counter(), collections.Counter()
for l in body:
    if re.match(r"^\.LBB\d+_\d+:", l):
        in_loop = "in Loop" in l or "Loop Header" in l
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    all_ops[op] += 1
    if in_loop:
        loop_ops[op] += 1


def cls(c):
    valu = sum(n for o, n in c.items() if o.startswith("v_") and not o.startswith("v_readlane") and not o.startswith("v_writelane"))
    return dict(valu=valu, readlane=sum(n for o, n in c.items() if o.startswith("v_readlane") or o.startswith("v_writelane")),
                trans=sum(n for o, n in c.items() if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", o)),
                pk=sum(n for o, n in c.items() if o.startswith("v_pk_")), salu=sum(n for o, n in c.items() if o.startswith("s_") and o != "s_nop" and not o.startswith("s_waitcnt")),
                nop=c.get("s_nop", 0), vmem=sum(n for o, n in c.items() if o.startswith("buffer_") or o.startswith("global_")),
                mov=sum(n for o, n in c.items() if o.startswith("v_mov")))


meta = {}
for key in ("next_free_vgpr", "next_free_sgpr", "private_segment_fixed_size"):
    m = re.search(r"\.amdhsa_kernel " + re.escape(kernel) + r".*?\.amdhsa_" + key + r" (\d+)", "\n".join(lines), re.S)
    meta[key] = int(m.group(1)) if m else None
print("kernel", kernel, meta)
a, lp = cls(all_ops), cls(loop_ops)
print("whole kernel:", a)
print("loop blocks :", lp, " -> VALU+lane ops per pose evaluation (loop body = %g): %.1f" % (per, (lp["valu"] + lp["readlane"]) / per))
print("loop mix:", ", ".join("%s %d" % (o, n) for o, n in loop_ops.most_common(40)))
