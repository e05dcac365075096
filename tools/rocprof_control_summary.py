#!/usr/bin/env python3
"""Condense tools/profile_control.sh output: mean FETCH_SIZE / WRITE_SIZE of the segsum launches of the uniform-source
control -> profiles/<tag>_control_pmc.json and the `control_uniform_bytes_per_launch` key of profiles/pmc_traffic.json.
usage: tools/rocprof_control_summary.py gpurun_out/prof_<tag>_control <tag> [fetch_scale]"""
import collections
import csv
import json
import os
import shutil
import sys


def per_kernel(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in d.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.992
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles")
    shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(out, f"{tag}_control_kernel_stats.csv"))
    f = per_kernel(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"))
    w = per_kernel(os.path.join(src, "pmc_write", "write_counter_collection.csv"))
    seg = [k for k in f if "segsum_kernel" in k]
    fix = []                                     # (rounds 1-3 had a second, fix-up launch)
    res = {"note": "bench.py --control-only under rocprofv3, FETCH_SIZE and WRITE_SIZE in separate --pmc runs (KB, raw); "
                   "FETCH_SIZE x fetch_scale = gfx950 calibration of tools/pmc_calibrate.py",
           "fetch_scale": scale, "kernels": {k: {"launches": f[k][0], "FETCH_SIZE_KB_raw": f[k][1], "WRITE_SIZE_KB": w.get(k, (0, 0))[1]}
                                             for k in seg + fix}}
    b = sum((f[k][1] * scale + w.get(k, (0, 0))[1]) * 1024 for k in seg + fix)
    res["control_uniform_bytes_per_launch"] = b
    json.dump(res, open(os.path.join(out, f"{tag}_control_pmc.json"), "w"), indent=1)
    p = os.path.join(out, "pmc_traffic.json")
    t = json.load(open(p)) if os.path.exists(p) else {}
    t["control_uniform_bytes_per_launch"] = b
    t["control_from"] = f"profiles/{tag}_control_pmc.json"
    json.dump(t, open(p, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
