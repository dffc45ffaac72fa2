#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per-kernel averages like --stats, next to the median and the average WITHOUT the single slowest
launch (one 31-ms launch -- a stall of the box, not the first launch -- among 187 moves an average by 18 %).
usage: tools/trace_stats_warm.py <dir> [out.csv]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv")
acc = collections.OrderedDict()
f.sort(key=lambda x: __import__("os").path.getmtime(x))          # several runs merged into one directory: the newest trace
for r in csv.DictReader(open(f[-1])):
    name = r["Kernel_Name"]
    if not any(k in name for k in ("k_forward", "k_backward", "k_vnudge", "k_diag", "k_interp", "k_exner", "k_searchsorted", "k_rms")):
        continue
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc.setdefault(name, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows = [("Name", "Calls", "AverageNs", "MedianNs", "AverageNs_without_the_slowest_launch", "MinNs", "MaxNs")]
for name, v in acc.items():
    d = sorted(x[1] for x in v)
    rest = d[:-1] or d
    rows.append((name, len(d), "%.1f" % (sum(d) / len(d)), d[len(d) // 2], "%.1f" % (sum(rest) / len(rest)), d[0], d[-1]))
w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
w.writerows(rows)
