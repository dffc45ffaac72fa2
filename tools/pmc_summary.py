#!/usr/bin/env python3
"""Parse rocprofv3 --pmc CSV outputs (FETCH_SIZE pass, WRITE_SIZE pass) into per-kernel HBM bytes per
launch, calibrated on the known-size copy kernels of tools/pmc_run.py. Writes profiles/traffic.json.
usage: pmc_summary.py <fetch_dir> <write_dir> <n_cols> <copy_bytes> [out.json]"""
import csv
import glob
import json
import os
import sys


def collect(d, counter):
    rows = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                rows.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return rows


def short(name):
    for k in ("k_forward", "k_backward_cons", "k_backward", "k_copy16", "k_copy8", "k_cloud_idx", "k_diag"):
        if k in name:
            return k
    return None


def summarise(fdir, wdir, n_cols, copy_bytes, tag=None):
    """the two --pmc passes' CSVs -> dict (per-kernel HBM bytes per launch, calibrated on the known-size copies)"""
    res = {"n_cols": n_cols, "copy_bytes": copy_bytes, "tag": tag or os.environ.get("PMC_TAG", "round 2"), "units": "rocprofv3 FETCH_SIZE / WRITE_SIZE are KiB", "kernels": {}}
    for counter, d in (("FETCH_SIZE", fdir), ("WRITE_SIZE", wdir)):
        for name, vals in collect(d, counter).items():
            k = short(name)
            if k is None:
                continue
            vals = vals[len(vals) // 3:]                       # drop the first (cold / warm-up) launches
            res["kernels"].setdefault(k, {})[counter + "_KiB_avg"] = sum(vals) / len(vals)
    ks = res["kernels"]
    # calibration: bytes the counter SHOULD report for the copies = copy_bytes (read) and copy_bytes (written)
    for k in ("k_copy16", "k_copy8"):
        if k in ks and "FETCH_SIZE_KiB_avg" in ks[k]:
            ks[k]["fetch_counter_over_true"] = ks[k]["FETCH_SIZE_KiB_avg"] * 1024 / copy_bytes
            ks[k]["write_counter_over_true"] = ks[k].get("WRITE_SIZE_KiB_avg", 0) * 1024 / copy_bytes
    res["calibration"] = {}
    # both first-generation kernels access 8 B per lane (PMC_CAL_K1=k_copy16 when a 16-B second-generation K1 is forced)
    for k, calk in (("k_forward", os.environ.get("PMC_CAL_K1", "k_copy8")), ("k_backward", "k_copy8"), ("k_backward_cons", "k_copy8")):
        cal = ks.get(calk, {})
        fr, wr = cal.get("fetch_counter_over_true", 0.5), cal.get("write_counter_over_true", 1.0)
        res["calibration"][k] = {"pattern": calk, "fetch_counter_over_true": fr, "write_counter_over_true": wr}
        if k in ks:
            f = ks[k].get("FETCH_SIZE_KiB_avg", 0) * 1024 / (fr or 1)
            w = ks[k].get("WRITE_SIZE_KiB_avg", 0) * 1024 / (wr or 1)
            ks[k]["hbm_read_bytes_per_launch"] = f
            ks[k]["hbm_write_bytes_per_launch"] = w
            ks[k]["hbm_bytes_per_launch"] = f + w
            ks[k]["hbm_bytes_per_column"] = (f + w) / n_cols
    res["k_forward_bytes_per_launch"] = ks.get("k_forward", {}).get("hbm_bytes_per_launch")
    res["k_backward_bytes_per_launch"] = ks.get("k_backward", {}).get("hbm_bytes_per_launch")
    if "k_backward_cons" in ks:
        res["k_backward_cons_bytes_per_launch"] = ks["k_backward_cons"].get("hbm_bytes_per_launch")
    return res


def main():
    fdir, wdir, n_cols, copy_bytes = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    out = sys.argv[5] if len(sys.argv) > 5 else "profiles/traffic.json"
    res = summarise(fdir, wdir, n_cols, copy_bytes)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
