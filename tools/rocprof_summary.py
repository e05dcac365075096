#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/profile_bench.sh) into the files kept under profiles/:
  <tag>_kernel_stats.csv   the --kernel-trace --stats summary, as rocprofv3 wrote it
  <tag>_pmc_summary.json   per-kernel mean FETCH_SIZE / WRITE_SIZE (KB, raw) from the separate --pmc passes
usage: tools/rocprof_summary.py gpurun_out/prof_<tag> <tag> [fetch_scale] [sage|gat|gcn|c5gat|bf16]
fetch_scale = calibration factor for FETCH_SIZE in the segsum access pattern (tools/pmc_calibrate.py).
The last argument says which constants of profiles/pmc_traffic.json the run provides: the headline SAGE launch (default),
the two GATConv aggregation kernels (bench.py --conv gat), or GCNConv's weighted launch (--conv gcn)."""
import collections
import csv
import json
import os
import shutil
import sys


def per_kernel(path):
    d = collections.defaultdict(list)
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "mean_KB": sum(v) / len(v)} for k, v in d.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else None
    kind = sys.argv[4] if len(sys.argv) > 4 else "sage"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles")
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats.csv"))
    fetch = per_kernel(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"))
    write = per_kernel(os.path.join(src, "pmc_write", "write_counter_collection.csv"))
    summ = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, {}), write.get(k, {})
        if max(f.get("mean_KB", 0), w.get("mean_KB", 0)) < 1000:
            continue
        summ[k.split("(")[0]] = {"FETCH_SIZE_KB_raw": f.get("mean_KB"), "WRITE_SIZE_KB": w.get("mean_KB"),
                                 "launches": f.get("launches") or w.get("launches")}
    res = {"note": "raw rocprofv3 counters, separate --pmc passes; FETCH_SIZE on gfx950 under-reports wide "
                   "coalesced reads (MI355X_MICROARCH.md, HBM): multiply by fetch_scale for the segsum pattern",
           "fetch_scale_segsum": scale, "kernels": summ}
    json.dump(res, open(os.path.join(out, f"{tag}_pmc_summary.json"), "w"), indent=1)
    def kb(pred):
        """bytes per launch of the kernels whose name satisfies pred, averaged over the instantiations that match"""
        vs = [v for k, v in summ.items() if pred(k)]
        return sum((v["FETCH_SIZE_KB_raw"] * scale + v["WRITE_SIZE_KB"]) * 1024 for v in vs) / len(vs) if vs else 0.0

    def mode(k):      # WMODE of a segsum instantiation name "...segsum_kernel<float, 4, 1, <mode>, true>"
        try:
            return int(k.split("<")[1].split(",")[3])
        except Exception:
            return -1
    if scale:
        sys.path.insert(0, root)
        from bench import kernel_source_sha
        p = os.path.join(out, "pmc_traffic.json")
        t = json.load(open(p)) if os.path.exists(p) else {}
        main = lambda m: (lambda k: "segsum_kernel<float" in k and mode(k) == m)
        t.pop("includes_fixup_kernel", None)        # (rounds 1-3: a second launch; cut rows are finished inside the launch now)
        if kind == "sage":
            # per aggregation launch (ONE kernel), averaged over the forward and the backward launch
            # (forward: WMODE 0, the unweighted mean; backward, aggregate-first since round 6: WMODE 1, per-entry weights)
            both = lambda k: "segsum_kernel<float" in k and mode(k) in (0, 1)
            t.update({"segsum_kernel_bytes_per_launch": kb(both), "fetch_scale": scale, "from": f"profiles/{tag}_pmc_summary.json"})
        elif kind == "gat":
            # forward: W_GAT_DST_FUSED (10: the statistics inside the launch, round 5); W_GAT_DST_PRE (6) under Schedule(gat_fused_stats=False)
            t.update({"gat_fwd_aggregate_bytes_per_launch": kb(main(10)) or kb(main(6)), "gat_bwd_fused_bytes_per_launch": kb(main(5)),
                      "gat_from": f"profiles/{tag}_pmc_summary.json"})
        elif kind == "c5gat":      # the same two kernels on the C5 graph (bench.py --conv gat --nodes 4000000 --edges 100000000 --graph-seed 2)
            t.update({"c5_gat_fwd_aggregate_bytes_per_launch": kb(main(10)) or kb(main(6)), "c5_gat_bwd_fused_bytes_per_launch": kb(main(5)),
                      "c5_from": f"profiles/{tag}_pmc_summary.json"})
        elif kind == "gcn":
            t.update({"gcn_segsum_bytes_per_launch": kb(main(1)), "gcn_from": f"profiles/{tag}_pmc_summary.json"})
        elif kind == "bf16":       # the headline layer with bf16 storage (bench.py --storage bf16): segsum_kernel<unsigned short, 4, 1, 0, true>
            bf = lambda k: "segsum_kernel<unsigned short" in k and mode(k) == 0
            t.update({"bf16_segsum_bytes_per_launch": kb(bf), "bf16_from": f"profiles/{tag}_pmc_summary.json"})
        t["source_sha16"] = kernel_source_sha()
        json.dump(t, open(p, "w"), indent=1)
    print(json.dumps(res, indent=1)[:1500])


if __name__ == "__main__":
    main()
