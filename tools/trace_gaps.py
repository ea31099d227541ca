#!/usr/bin/env python3
"""Reads a rocprofv3 kernel_trace.csv and prints, for the GD loop, the average duration of the loss and epilogue kernels and
the idle gaps between consecutive kernels.   python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].replace("void ", "", 1).split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
loss, epi, gap_le, gap_el = [], [], [], []
for (n0, s0, e0), (n1, s1, e1) in zip(seq, seq[1:]):
    if n0.startswith("pcl_loss_kernel") and n1.startswith("pcl_gd_epilogue"):
        loss.append(e0 - s0); gap_le.append(s1 - e0)
    if n0.startswith("pcl_gd_epilogue") and n1.startswith("pcl_loss_kernel"):
        epi.append(e0 - s0); gap_el.append(s1 - e0)
avg = lambda v: sum(v) / max(len(v), 1) / 1e3  # noqa: E731
print("pairs %d: loss %.2f us | gap %.2f us | epilogue %.2f us | gap %.2f us  => %.2f us per iteration" % (
    len(loss), avg(loss), avg(gap_le), avg(epi), avg(gap_el), avg(loss) + avg(gap_le) + avg(epi) + avg(gap_el)))
