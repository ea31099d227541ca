#!/usr/bin/env python3
"""What `roofline.frac` is a fraction OF: the VALU "peak" of bench.py prices every wave64 instruction at 4
cycles, while the SIMD issues a v_mul / v_add in 2.4, a v_fma_f32 in 3.1, every packed / VOP3 op in 4.3 and a v_rcp / v_rsq in 8.2 cycles
(tools/micro/valu_rate.hip, measured on the box).  This script prices the loss kernel's OWN instruction mix:

  1. compiles csrc/pcl_loss.hip to ISA (hipcc -S, device only: no GPU needed) and counts the VALU opcodes of the point loop of the
     shipped instances pcl_loss_kernel<G = 2, GRAD, VIS = 0, FMT = f16 | u8>;
  2. reads the issue cost of every opcode class from the microbenchmark's table (profiles/r06/valu_rate.txt: `valu_rate` run in the same
     collection as the counters) — an opcode the table does not hold takes its class's cost (table below);
  3. -> mix_ceiling_cycles_per_instr = sum(count x cycles) / sum(count): the issue cycles the kernel's own stream needs per instruction;
        frac_of_mix_ceiling = frac x mix / 4 (the same achieved instruction rate against 1024 SIMDs x 2.4 GHz / mix);
        fp32_flops_per_point_pose (an fma = 2, a packed op twice that) for `fp32_flop_frac` against the 157.3 TFLOP/s vector peak;
  4. writes profiles/r06/roof_mix.json and stamps the two per-instance figures into the entries of profiles/roofs.json (bench.py reports
     them in `roofline`).

   python tools/roof_mix.py [--table profiles/r06/valu_rate.txt] [--roofs profiles/roofs.json] [--out profiles/r06/roof_mix.json]"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLOCK_GHZ, SIMDS, FP32_PEAK_TFLOPS = 2.4, 1024, 157.3

# table row name -> (regex of the opcodes it prices)
TABLE_ROWS = [
    ("v_rcp_f32", r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_"), ("v_pk_fma_f32", r"v_pk_"), ("v_fma_mix_f32", r"v_fma_mix"),
    ("v_fma_f32", r"v_(fma|fmac)_f32"), ("v_mul_f32", r"v_mul_f32"), ("v_add_f32", r"v_(add|sub|subrev)_f32"), ("v_mov_b32", r"v_mov_b32"),
    ("v_cndmask_b32_e64 .., vcc", r"v_cndmask_b32"), ("v_med3_f32", r"v_(med3|max3|min3)_"), ("v_bfi_b32", r"v_(bfi|bfe|perm|alignbit|and_or|lshl_or|or3|xad)_"),
    ("v_cvt_i32_f32", r"v_cvt_"), ("v_fract_f32", r"v_(fract|floor|trunc|rndne|ceil)_"), ("v_cmp_gt_f32 -> vcc", r"v_cmp"), ("v_mul_u32_u24", r"v_(mul_u32_u24|mul_i32_i24|mad_u32_u24|mul_lo|mul_hi|mad_)"),
    ("v_add_lshl_u32", r"v_(add_lshl|lshl_add|add3|add_u32|add_co|sub_u32|sub_co|subrev|lshlrev|lshrrev|ashrrev|and_b32|or_b32|xor_b32|not_b32)"),
    ("v_min_f32", r"v_(min|max)_"),
]
DEFAULT_ROW = "v_med3_f32"      # any other VOP1 / VOP2 / VOP3 op: the 4.3-cycle class every non-trivial op measured in
FLOPS = [(r"v_pk_fma_f32", 4), (r"v_pk_(mul|add)_f32", 2), (r"v_(fma|fmac)_f32", 2), (r"v_fma_mix", 2), (r"v_(mul|add|sub|subrev)_f32", 1),
         (r"v_pk_(fma)_f16", 4), (r"v_pk_(mul|add)_f16", 2)]


def read_table(path):
    """{row name: cycles per wave64 instruction per SIMD} from valu_rate's output"""
    out = {}
    for ln in open(path):
        m = re.match(r"^(.*?)\s+[\d.]+ ms\s+([\d.]+) ns per wave-instruction per SIMD", ln)
        if m:
            out[m.group(1).strip()] = float(m.group(2)) * CLOCK_GHZ
    return out


def loop_ops(listing, kernel):
    lines = listing.split("\n")
    start = next(i for i, ln in enumerate(lines) if ln.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    ops, in_loop = collections.Counter(), False
    for ln in lines[start:end + 1]:
        if re.match(r"^\.LBB\d+_\d+:", ln):
            in_loop = "in Loop" in ln or "Loop Header" in ln
            continue
        t = ln.strip()
        if in_loop and t and not t.startswith(";") and not t.startswith("."):
            ops[t.split()[0]] += 1
    return ops


def price(ops, table):
    valu = {o: c for o, c in ops.items() if o.startswith("v_") and not re.match(r"v_(readlane|writelane|readfirstlane)", o)}
    rows, cycles, flops, unpriced = collections.Counter(), 0.0, 0, collections.Counter()
    for o, c in valu.items():
        row = next((name for name, rx in TABLE_ROWS if re.match(rx, o)), None)
        if row is None:
            row = DEFAULT_ROW
            unpriced[o] += c
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", o)
        cyc = table.get(base, table[row])                      # an exact row of the table wins
        rows[row] += c
        cycles += c * cyc
        flops += c * next((f for rx, f in FLOPS if re.match(rx, o)), 0)
    n = sum(valu.values())
    return {"valu_instructions_in_loop": n, "mix_ceiling_cycles_per_instr": cycles / n, "fp32_flops_per_instr": flops / n,
            "by_class": {k: {"count": v, "cycles_each": round(table[k], 3)} for k, v in rows.most_common()},
            "priced_by_class_default": dict(unpriced), "s_nop_in_loop": ops.get("s_nop", 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", default=os.path.join(REPO, "profiles", "r06", "valu_rate.txt"))
    ap.add_argument("--roofs", default=os.path.join(REPO, "profiles", "roofs.json"))
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r06", "roof_mix.json"))
    args = ap.parse_args()
    table = read_table(args.table)
    need = {name for name, _ in TABLE_ROWS}
    assert need <= set(table), sorted(need - set(table))
    with tempfile.TemporaryDirectory() as tmp:
        s = os.path.join(tmp, "loss.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S",
                               os.path.join(REPO, "piccolo_amd", "csrc", "pcl_loss.hip"), "-o", s], stderr=subprocess.DEVNULL)
        listing = open(s).read()
    sys.path.insert(0, REPO)
    from piccolo_amd import build
    result = {"table": os.path.relpath(args.table, REPO), "clock_GHz": CLOCK_GHZ, "loss_kernel_source_hash": build.loss_kernel_source_hash(),
              "class_costs_cycles": {k: round(v, 3) for k, v in sorted(table.items())}, "instances": {}}
    for fmt_name, fmt_code in (("f16", 2), ("u8", 1), ("f32", 0)):
        kernel = "_Z15pcl_loss_kernelILi2ELb1ELi0ELi%dEEv11PclLossArgs" % fmt_code
        if kernel + ":" not in listing:
            continue
        result["instances"][fmt_name] = dict(kernel=kernel, **price(loop_ops(listing, kernel), table))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(result, open(args.out, "w"), indent=1, sort_keys=True)
    for k, v in result["instances"].items():
        print("%s: %d VALU instructions in the loop, mix ceiling %.3f cycles per instruction (bench.py's convention: 4), %.3f fp32 flops per instruction"
              % (k, v["valu_instructions_in_loop"], v["mix_ceiling_cycles_per_instr"], v["fp32_flops_per_instr"]))
    # stamp the roofs entries of the SAME loss-kernel sources
    if os.path.exists(args.roofs):
        roofs = json.load(open(args.roofs))
        n = 0
        for key, e in roofs.items():
            inst = result["instances"].get(key.rsplit("/", 1)[-1])
            if inst and e.get("source_hash") == result["loss_kernel_source_hash"]:
                e["mix_ceiling_cycles_per_instr"] = inst["mix_ceiling_cycles_per_instr"]
                e["fp32_flops_per_point_pose"] = inst["fp32_flops_per_instr"] * e["valu_instr_per_point_pose"]
                e["mix_source"] = os.path.relpath(args.out, REPO)
                n += 1
        json.dump(roofs, open(args.roofs, "w"), indent=1, sort_keys=True)
        print("stamped %d of %d entries of %s" % (n, len(roofs), os.path.relpath(args.roofs, REPO)))


if __name__ == "__main__":
    main()
