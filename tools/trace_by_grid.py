#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per (kernel, grid size) call count and average duration.  `--stats` averages per
kernel NAME; bench.py launches the same K1 / K3 instantiation on config 3 (35 718 columns) and, for `scaling_anchor`, on
config 4 (348 528 columns): this splits them.  usage: tools/trace_by_grid.py <dir with *_kernel_trace.csv> [out.csv]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv")
acc = collections.OrderedDict()
f.sort(key=lambda x: __import__("os").path.getmtime(x))          # several runs merged into one directory: the newest trace
for r in csv.DictReader(open(f[-1])):
    name = r["Kernel_Name"]
    if "k_forward" not in name and "k_backward" not in name and "k_vnudge" not in name and "k_diag" not in name:
        continue
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    key = (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"]))
    a = acc.setdefault(key, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = [("Name", "Workgroups", "WorkgroupSize", "Calls", "AverageNs")] + [(k[0], k[1], k[2], v[0], "%.1f" % (v[1] / v[0])) for k, v in acc.items()]
w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
w.writerows(rows)
