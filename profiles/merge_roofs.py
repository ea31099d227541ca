#!/usr/bin/env python3
"""Merge the per-run summary/roofs.json files of profiles/collect.sh into profiles/roofs.json (what bench.py reads) and, on the GPU
box, into gpurun_out/roofs.json so that the merged file travels back.   python3 profiles/merge_roofs.py gpurun_out/prof_*/summary/roofs.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def merge(paths, name):
    dst = os.path.join(HERE, name)
    try:
        roofs = json.load(open(dst))
    except Exception:
        roofs = {}
    for p in paths:
        try:
            roofs.update(json.load(open(p)))
        except Exception as exc:
            print("skipped", p, exc)
    for out in (dst, os.path.join(os.path.dirname(HERE), "gpurun_out", name)):
        try:
            os.makedirs(os.path.dirname(out), exist_ok=True)
            json.dump(roofs, open(out, "w"), indent=1, sort_keys=True)
        except OSError as exc:
            print("cannot write", out, exc)
    return roofs


def main(paths):
    pipe = [p for p in paths if os.path.basename(p) == "pipeline_roofs.json"]
    paths = [p for p in paths if os.path.basename(p) != "pipeline_roofs.json"]
    if pipe:
        pr = merge(pipe, "pipeline_roofs.json")
        print("pipeline_roofs.json:", {k: (v.get("library_hash"), {kn: (kr.get("binding_roof"), round(kr.get("binding_frac", 0), 3)) for kn, kr in v.get("kernels", {}).items()})
                                       for k, v in sorted(pr.items())})
    if not paths:
        return
    dst = os.path.join(HERE, "roofs.json")
    try:
        roofs = json.load(open(dst))
    except Exception:
        roofs = {}
    for p in paths:
        try:
            roofs.update(json.load(open(p)))
        except Exception as exc:
            print("skipped", p, exc)
    for out in (dst, os.path.join(os.path.dirname(HERE), "gpurun_out", "roofs.json")):
        try:
            os.makedirs(os.path.dirname(out), exist_ok=True)
            json.dump(roofs, open(out, "w"), indent=1, sort_keys=True)
        except OSError as exc:
            print("cannot write", out, exc)
    print("roofs.json:", {k: (v.get("source_hash"), round(v.get("valu_instr_per_point_pose", 0), 3)) for k, v in sorted(roofs.items())})


if __name__ == "__main__":
    main(sys.argv[1:])
