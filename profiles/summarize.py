#!/usr/bin/env python3
"""Condense rocprofv3 output of profiles/collect.sh into small text/JSON summaries.

  <dir>/kt/**/*kernel_stats.csv          -> summary/kernel_stats.txt   (per-kernel calls / total / average)
  <dir>/pmc*/**/*counter_collection.csv  -> summary/pmc.json           (per kernel: mean counter value per dispatch)
  + summary/kernel_roofs.json: per pcl_* kernel the memory-side bytes per launch, corrected as MI355X_MICROARCH.md
    prescribes (FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B, i.e. it
    reports half the bytes of a wide stream -> doubled), L2 hit rate, VALU-busy fraction
    (4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): SQ_* count quad-cycles) and the wave-cycle split
  + summary/roofs.json (if ROOF_KEY is set): the loss kernel's entry under its launch-shape key, the format of
    profiles/roofs.json that bench.py reads (ROOF_KEY = workload/posesN/texels, POINT_POSES = points x poses per launch).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def library_hash(fn="pcl_source_hash"):
    """pcl_source_hash() (loss-kernel sources) or pcl_library_hash() (every source) of the library the profiled command loaded
    (host-only calls): bench.py scores an entry only for a library with the same sources."""
    import ctypes
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "piccolo_amd", "lib", "libpiccolo_hip.so")
    lib = ctypes.CDLL(os.environ.get("PCL_SO", so))
    getattr(lib, fn).restype = ctypes.c_char_p
    return getattr(lib, fn)().decode()


def short(name):
    return name.split("(")[0].replace("void ", "")[:70]


def main(d):
    out = os.path.join(d, "summary")
    os.makedirs(out, exist_ok=True)
    stats = glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True)
    lines = []
    avg_ns = {}                                            # kernel -> average duration (ns) in the kernel-trace run
    from_summary = not stats and os.path.exists(os.path.join(out, "pmc.json"))
    if from_summary:
        # the raw rocprofv3 trees are deleted on the GPU box (tens of MB per pass): recompute the roofs from what travelled back
        for ln in open(os.path.join(out, "kernel_stats.txt")):
            mm = re.match(r"^(\S.*?)\s+(\d+)\s+(\d+)\s+(\d+)\s+[0-9.eE+-]+\s*$", ln)
            if mm and not ln.startswith("kernel "):
                avg_ns[mm.group(1).strip()] = float(mm.group(4))
    for f in stats:
        rows = list(csv.DictReader(open(f)))
        for r in rows:
            avg_ns[short(r["Name"])] = float(r["AverageNs"])
        lines.append("# %s" % os.path.relpath(f, d))
        lines.append("%-72s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        for r in rows:
            lines.append("%-72s %8s %14s %12.0f %8s" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"],
                                                       float(r["AverageNs"]), r["Percentage"]))
    if not from_summary:
        open(os.path.join(out, "kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
        print("\n".join(lines[:14]))

    if from_summary:
        summ = json.load(open(os.path.join(out, "pmc.json")))
    else:
        pmc = defaultdict(lambda: defaultdict(list))
        for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        summ = {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()} for k, cs in pmc.items()}
        json.dump(summ, open(os.path.join(out, "pmc.json"), "w"), indent=1, sort_keys=True)
    # per kernel: memory-side traffic (FETCH_SIZE doubled, MI355X_MICROARCH.md) and the VALU roof
    roofs, point_poses, key = {}, float(os.environ.get("POINT_POSES", "0") or 0), os.environ.get("ROOF_KEY", "")
    for k, c in summ.items():
        if not k.startswith("pcl_"):
            continue
        m = {n: v["mean"] for n, v in c.items()}
        r = {"kernel": k, "dispatches_sampled": max(v["n"] for v in c.values())}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            fetch, write = m["FETCH_SIZE"] * 1024, m["WRITE_SIZE"] * 1024
            r.update(fetch_size_bytes_raw=fetch, write_size_bytes=write, hbm_bytes_per_launch=2 * fetch + write,
                     traffic_note="2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; gfx950 reports half the bytes of wide reads); "
                                  "memory-side request counters, Infinity-Cache hits included")
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
            r["l2_hit_frac"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
            for n in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
                if n in m:
                    r[n.lower() + "_per_wave_cycle"] = m[n] / m["SQ_WAVE_CYCLES"]
        # ---- the roof that BINDS this kernel (round 5): every candidate roof as achieved / peak, the largest one named.
        # Cycles of the dispatch: GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs; it agrees with the kernel-trace duration x 2.4 GHz
        # to 1 % on every kernel checked: loss 258 865 vs 256 994, resolve 1 157 154 vs 1 154 724), else duration x 2.4 GHz.
        # (SQ_BUSY_CYCLES / 32 reads 5-15 % low — a shader engine that has run out of work stops counting — and gave ratios above 1.)
        fr = {}
        dur = avg_ns.get(k)
        cycles = m["GRBM_GUI_ACTIVE"] / 8.0 if m.get("GRBM_GUI_ACTIVE", 0) > 0 else (dur * 2.4 if dur else 0.0)
        if dur:
            r["avg_duration_us_kernel_trace"] = dur / 1e3
        if cycles > 0:
            r["dispatch_cycles"] = cycles
            if "SQ_INSTS_VALU" in m:
                # issue slots under the 4-cycle model (one wave64 VALU instruction per SIMD per 4 cycles: what the loss kernel's packed /
                # FMA mix costs, tools/micro/valu_rate.hip: 1.75 ns).  Plain v_mov / v_add / v_mul / compares issue in 2.4 cycles
                # (1.00 ns, same tool), so a kernel made of integer and select work can pass 1.0 of THIS peak: it is then reported
                # against the 2.4-cycle peak as well, and that is what the binding decision uses.
                f4 = 4.0 * m["SQ_INSTS_VALU"] / (1024.0 * cycles)
                fr["valu_issue"] = f4
                if f4 > 1.0:
                    r["valu_issue_vs_4_cycle_peak"] = f4
                    fr["valu_issue"] = 2.4 * m["SQ_INSTS_VALU"] / (1024.0 * cycles)
                    r["valu_issue_note"] = ("above the 4-cycle peak: this kernel's VALU work is mostly plain integer / move / compare instructions, "
                                            "which issue every 2.4 cycles (tools/micro/valu_rate.hip: 1.00 ns against 1.75 ns for packed / FMA ops); "
                                            "valu_issue is against that 2.4-cycle peak")
            if "SQ_ACTIVE_INST_VALU" in m:
                r["valu_busy_frac"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles)
                r.pop("valu_busy_frac_raw", None); r.pop("valu_busy_note", None)
                if r["valu_busy_frac"] > 1.0:
                    r["valu_busy_note"] = ("SQ_ACTIVE_INST_VALU sums the waves' in-flight VALU time: transcendental and plain pipes overlap across "
                                           "waves of a SIMD, so the sum can pass the SIMD's cycles; use valu_issue")
            if "TCP_TOTAL_CACHE_ACCESSES_sum" in m:
                # a RATE, not a fraction: L1 tag lookups per cycle per CU (the trim launch reaches 1.0-1.2; the peak is not documented).
                # The texture path's roof is the texture-address unit's busy fraction below.
                r["l1_line_lookups_per_cycle_per_cu"] = m["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256.0 * cycles)
            if m.get("TA_BUSY_avr", 0) > 1.5:
                fr["texture_unit_busy"] = m["TA_BUSY_avr"] / cycles
            if "SQ_LDS_IDX_ACTIVE" in m:
                fr["lds_busy"] = m["SQ_LDS_IDX_ACTIVE"] / (256.0 * cycles)
        if dur:
            if "hbm_bytes_per_launch" in r:
                fr["hbm"] = r["hbm_bytes_per_launch"] / (dur * 1e-9) / 8.0e12
            if "TCC_EA0_ATOMIC_sum" in m:
                r["memory_side_atomic_requests"] = m["TCC_EA0_ATOMIC_sum"]
                fr["memory_side_atomics"] = m["TCC_EA0_ATOMIC_sum"] * 64.0 / (dur * 1e-9) / 1.3e12   # 64-byte requests against 1.3 TB/s
        if fr:
            r["roof_fractions"] = fr
            r["binding_roof"] = max(fr, key=fr.get)
            r["binding_frac"] = fr[r["binding_roof"]]
            r["roofs_are"] = ("valu_issue: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x dispatch cycles) (x 2.4 cycles where the 4-cycle figure passes 1); "
                              "texture_unit_busy: TA_BUSY_avr / cycles (the texture-address units' busy fraction: the gathers' roof); lds_busy: "
                              "SQ_LDS_IDX_ACTIVE / (256 x cycles); hbm: memory-side bytes / duration / 8 TB/s; memory_side_atomics: TCC_EA0_ATOMIC x "
                              "64 B / duration / 1.3 TB/s (MI355X_MICROARCH.md); cycles = GRBM_GUI_ACTIVE / 8")
        if "SQ_INSTS_VALU" in m:
            r["valu_insts_per_launch"] = m["SQ_INSTS_VALU"]
            if point_poses and ((k.startswith("pcl_loss_kernel") and "true" in k.split("<")[1].split(",")[1]) or k.startswith("pcl_loss_fused_kernel")):
                r["valu_instr_per_point_pose"] = m["SQ_INSTS_VALU"] / (point_poses / 64.0)
        roofs[k] = r
        print(k, {n: (round(v, 4) if isinstance(v, float) else v) for n, v in r.items() if n not in ("traffic_note", "kernel")})
    json.dump(roofs, open(os.path.join(out, "kernel_roofs.json"), "w"), indent=1, sort_keys=True)
    ptag = os.environ.get("PIPELINE_TAG", "")
    if ptag:
        # profiles/pipeline_roofs.json format (bench.py attaches it to also.pipeline*): per pipeline shape, per kernel, the binding roof
        keep = {k: {n: v for n, v in r.items() if n in ("avg_duration_us_kernel_trace", "binding_roof", "binding_frac", "roof_fractions", "hbm_bytes_per_launch", "l1_line_lookups_per_cycle_per_cu",
                                                         "valu_issue_vs_4_cycle_peak", "valu_issue_note",
                                                         "l2_hit_frac", "memory_side_atomic_requests", "valu_insts_per_launch", "dispatches_sampled")}
                for k, r in roofs.items() if "binding_roof" in r and r.get("avg_duration_us_kernel_trace", 0) >= float(os.environ.get("PIPELINE_MIN_US", "4"))
                and re.search(os.environ.get("PIPELINE_KERNELS", "zpass|zcache|pcl_loss|epilogue|fill_u32|pcl_trim|pcl_bin|resolve|pcl_select|pcl_hist|pcl_depth"), k)}
        json.dump({ptag: {"library_hash": library_hash(fn="pcl_library_hash"), "source": "profiles/%s" % os.environ.get("ROOF_SOURCE", os.path.basename(d.rstrip("/"))),
                          "command": os.environ.get("ROOF_CMD", ""), "kernels": keep}},
                  open(os.path.join(out, "pipeline_roofs.json"), "w"), indent=1, sort_keys=True)
    if key:
        # the GRAD variant of the loss kernel is the one bench.py's roofline is about
        # (a fused chain runs 1 plain + 99 fused launches per refinement: the entry is the kernel with the most dispatches)
        cand = [r for k, r in roofs.items() if (k.startswith("pcl_loss_kernel") and ", true," in k) or k.startswith("pcl_loss_fused_kernel")]
        if cand:
            best = dict(max(cand, key=lambda r: (r.get("dispatches_sampled", 0), r.get("valu_insts_per_launch", 0))))
            best["source_hash"] = library_hash()
            best["source"] = "profiles/%s (rocprofv3 --pmc passes of: %s)" % (os.environ.get("ROOF_SOURCE", os.path.basename(d.rstrip("/"))), os.environ.get("ROOF_CMD", "bench.py"))
            json.dump({key: best}, open(os.path.join(out, "roofs.json"), "w"), indent=1, sort_keys=True)
            print("roofs.json:", key, {n: best.get(n) for n in ("hbm_bytes_per_launch", "valu_instr_per_point_pose", "valu_busy_frac")})


if __name__ == "__main__":
    main(sys.argv[1])
