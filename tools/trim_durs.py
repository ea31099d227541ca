import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "pcl_trim_kernel" in r["Kernel_Name"]]
for i in range(0, len(d), 10):
    seg = d[i:i + 10]
    print("calls %d-%d: median %.1f us  all: %s" % (i, i + len(seg) - 1, sorted(seg)[len(seg) // 2], " ".join("%.0f" % v for v in seg)))
