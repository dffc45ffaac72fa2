"""per-kernel durations of the variability-nudge kernels from a rocprofv3 --kernel-trace CSV directory (tools/vn_probe.py)"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "vnudge" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = collections.OrderedDict()
half = len(rows) // 2
for i, r in enumerate(rows):
    key = ("synthetic" if i < half else "idle", r["Kernel_Name"].split("::")[1].split("(")[0])
    d.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print("%-10s %-20s n=%d  min %.1f us  median %.1f us" % (k[0], k[1], len(v), min(v), sorted(v)[len(v) // 2]))
