#!/usr/bin/env python3
"""Reads bench.py's JSON line on stdin and prints the headline fields on one short line."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1] if len(sys.argv) > 1 else "", d["config"]["workload"][:5], "value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3),
      "kernel_us", round(r["avg_launch_ms"] * 1e3, 1), "frac", round(r["frac"], 3), "timed", r["launches_timed"],
      "t_err", round(d["median_t_err_m"], 4), "r_err", round(d["median_r_err_deg"], 3))
