#!/usr/bin/env python3
"""Summary of tools/k7_profile.sh: per K7 kernel the rocprofv3 average duration, HBM bytes per launch from the PMC passes
(FETCH_SIZE / WRITE_SIZE in KiB; FETCH calibrated on the 256 MiB copies of the same run, as the microarch guide prescribes)
against the algorithmic bytes, and the SQ counters that say where the waves' cycles go.  usage: k7_summary.py <dir> [n_rows]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 35718
nG, nL = 91, 160
ALG = {"k_interp fwd": n * (2 * nG + nL) * 8, "k_interp bwd": n * (nG + nL + nG) * 8, "k_searchsorted": n * (nG + 1) * 16,
       "k_exner": n * 2 * nG * 8, "k_interp_c": n * (nG + 1 + 2 * nL + nG) * 8, "k_rms": n * (nL + 1) * 8}


def short(name):
    import re
    m = re.search(r"(?:^|::|\s)(k_interp_c|k_interp|k_searchsorted|k_exner|k_rms|k_copy16|k_copy8)\b", name)
    return m.group(1) if m else None


def counters(sub):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sorted(glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k:
                out[k][r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    return out


def split_interp(vals):
    """k_interp is launched in two shapes (GCM->LES first, then LES->GCM): first half / second half of its dispatches"""
    vals = sorted(vals)
    h = len(vals) // 2
    return [v for _, v in vals[:h]], [v for _, v in vals[h:]]


def avg(v):
    v = v[len(v) // 3:]
    return sum(v) / max(1, len(v))


# durations
dur = collections.defaultdict(list)
for path in glob.glob(os.path.join(d, "stats", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k:
            dur[k].append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows = {}
for k, v in dur.items():
    if k == "k_interp":
        a, b = split_interp(v)
        rows["k_interp fwd"], rows["k_interp bwd"] = avg(a), avg(b)
    else:
        rows[k] = avg([x for _, x in sorted(v)])
fetch, write, sq1, sq2 = counters("fetch"), counters("write"), counters("sq1"), counters("sq2")


def per(kernel, table, counter):
    v = table.get(kernel.split()[0], {}).get(counter)
    if not v:
        return None
    if kernel.startswith("k_interp "):
        a, b = split_interp(v)
        return avg(a if kernel.endswith("fwd") else b)
    return avg([x for _, x in sorted(v)])


cal_bytes = 1 << 28
cal = {}
for k in ("k_copy16", "k_copy8"):
    f, w = per(k, fetch, "FETCH_SIZE"), per(k, write, "WRITE_SIZE")
    if f:
        cal[k] = (f * 1024 / cal_bytes, (w or 0) * 1024 / cal_bytes)
        print("calibration %s: FETCH_SIZE reports %.3f x, WRITE_SIZE %.3f x the true bytes of a 256 MiB copy" % (k, cal[k][0], cal[k][1]))
fcal = cal.get("k_copy8", (0.5, 1.0))
print("%-16s %9s %9s %7s | %10s %10s %7s | wait_any wait_inst active | VALU/wave LDS/wave conflicts | waves  vmem_rd vmem_wr" % (
    "kernel", "avg us", "alg MB", "frac", "fetch MB", "write MB", "x alg"))
for k in ("k_interp fwd", "k_interp bwd", "k_searchsorted", "k_exner", "k_interp_c", "k_rms"):
    if k not in rows:
        continue
    t, alg = rows[k], ALG[k]
    f, w = per(k, fetch, "FETCH_SIZE"), per(k, write, "WRITE_SIZE")
    fb = f * 1024 / fcal[0] if f else None
    wb = w * 1024 / (fcal[1] or 1.0) if w else None
    wc = per(k, sq1, "SQ_WAVE_CYCLES")
    g = lambda tab, c: per(k, tab, c) or 0.0          # noqa: E731
    waves = g(sq2, "SQ_WAVES")
    print("%-16s %9.1f %9.1f %7.3f | %10s %10s %7s | %7.0f%% %8.0f%% %5.0f%% | %9.0f %8.0f %8.1f%% | %6.0f %7.0f %7.0f" % (
        k, t, alg / 1e6, alg / (t * 1e-6) / 8e12, "%.1f" % (fb / 1e6) if fb else "-", "%.1f" % (wb / 1e6) if wb else "-",
        "%.3f" % ((fb + wb) / alg) if fb and wb else "-",
        100 * g(sq1, "SQ_WAIT_ANY") / wc if wc else 0, 100 * g(sq1, "SQ_WAIT_INST_ANY") / wc if wc else 0,
        100 * g(sq1, "SQ_ACTIVE_INST_ANY") / wc if wc else 0,
        g(sq1, "SQ_INSTS_VALU") / waves if waves else 0, g(sq2, "SQ_INSTS_LDS") / waves if waves else 0,
        100 * g(sq1, "SQ_LDS_BANK_CONFLICT") / max(1.0, g(sq1, "SQ_LDS_IDX_ACTIVE")), waves,
        g(sq2, "SQ_INSTS_VMEM_RD") / waves if waves else 0, g(sq2, "SQ_INSTS_VMEM_WR") / waves if waves else 0))
