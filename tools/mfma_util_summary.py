#!/usr/bin/env python3
"""MFMA utilisation of the projection kernels from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass (counters in a run
of their own, with --kernel-trace only): busy matrix-pipe cycles summed over the chip's 1,024 SIMDs / (elapsed shader cycles x 1,024).
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  usage: tools/mfma_util_summary.py <counter_collection.csv> <out.json>"""
import collections, csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "gemm_" not in k:
        continue
    key = (k.split("(")[0], int(r["Grid_Size"]))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[key]["ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for (name, grid), c in sorted(agg.items()):
    n = len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    if n < 5:
        continue
    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / n
    cyc = sum(c["GRBM_GUI_ACTIVE"]) / n / 8.0
    ns = sum(c["ns"]) / len(c["ns"])
    out[f"{name} grid={grid}"] = {"launches": n, "mfma_busy_cycles_all_simds": busy, "elapsed_cycles": cyc, "avg_ns_under_pmc": ns,
                                  "clock_ghz": cyc / ns, "mfma_util": busy / (cyc * 1024)}
json.dump({"what": "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1,024 SIMDs); tools/gemm_pair.py under "
                   "rocprofv3 --pmc (durations under the counters are longer than in a plain run)", "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(f"{k}: util {v['mfma_util']:.3f}  clock {v['clock_ghz']:.2f} GHz  {v['avg_ns_under_pmc'] / 1e3:.0f} us  ({v['launches']} launches)")
