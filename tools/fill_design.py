#!/usr/bin/env python3
"""DESIGN.md = tools/doc/DESIGN.in.md with every @KEY@ replaced by the number of ONE bench line
(profiles/<tag>_bench.json) and its rocprofv3 kernel stats: one current number per claim, no hand-copied figures.
usage: python tools/fill_design.py r04a [tag of the rocprofv3 kernel stats, default the same]"""
import csv, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
stats_tag = sys.argv[2] if len(sys.argv) > 2 else tag
P = lambda n: os.path.join(root, "profiles", f"{stats_tag if n.endswith('.csv') else tag}_{n}")
b = json.load(open(P("bench.json")))
c, r = b["configs"], b["roofline"]


def stats(name):
    out = {}
    if not os.path.exists(P(name)):
        return out
    for row in csv.DictReader(open(P(name))):
        out[row["Name"]] = row
    return out


def avg_ms(st, *needles):
    for k, row in st.items():
        if all(n in k for n in needles):
            return float(row["AverageNs"]) / 1e6
    return float("nan")


def minmax_ms(st, *needles):
    for k, row in st.items():
        if all(n in k for n in needles):
            return float(row["MinNs"]) / 1e6, float(row["MaxNs"]) / 1e6
    return float("nan"), float("nan")


ks = stats("kernel_stats.csv")
seg_min, seg_max = minmax_ms(ks, "segsum_kernel<float, 4, 1, 0")
v = c["C4_w8_virtual"]
g, g5 = c["gat_c4"]["roofline"], c["C5_1gpu"]["roofline"]
gcn = c["gcn_c4"]["roofline"]
rng = lambda xs: "%.2f–%.2f" % (min(xs), max(xs))
GBs = lambda x: "%.2f" % (x / 1e9)
f2 = lambda x: "%.2f" % x
pg = b["projection"]["per_gemm_ms"]
val = {
    "STEP": f2(b["ms_per_step"]), "GEPS": "%.2f" % (b["value"] / 1e9), "REP": " / ".join(f2(x) for x in b["ms_per_step_repeats"]),
    "SEG": f2(r["avg_launch_ms"]), "SEGF": f2(seg_min), "PAIR": f2(seg_max),
    "FRACA": f2(r["frac_algorithmic"]), "FRACT": f2(r["frac_traffic"]), "PMC": GBs(r["traffic"]),
    "CSEG": f2(r["control_uniform"]["avg_launch_ms"]), "CFRAC": f2(r["control_uniform"]["frac_traffic"]),
    "CPMC": GBs(r["control_uniform"]["traffic"]),
    "GF": f2(pg["fwd"]), "GB": f2(pg["bwd_data"]),
    "CPU": "%.2f" % (b["cpu_baseline"]["value"] / 1e6),
    "GCN": f2(c["gcn_c4"]["ms_per_step"]), "GAT": f2(c["gat_c4"]["ms_per_step"]),
    "GATH": " / ".join(f2(x) for x in c["gat_c4"]["ms_per_step_by_heads"].values()),
    "C5": "%.1f" % c["C5_1gpu"]["ms_per_step"], "C5R": "%.2f" % (c["C5_1gpu"]["edge_layers_per_s"] / 1e9),
    "C5L": "%.1f" % c["C5_1gpu"].get("ms_per_step_with_mse_loss", float("nan")),
    "C1": f2(c["C1"]["ms_per_step"]), "C1G": "%.3f" % (c["C1"].get("ms_per_step_graph") or float("nan")),
    "C3G": "%.3f" % (c["C3"].get("ms_per_step_graph") or float("nan")),
    "PAR": "%.1e" % b["parity_max_err"] if b.get("parity_max_err") is not None else "n/a",
    "VSR": f2(v["hubs_sage"]["rank0_ms_graph_replay"]) if v["hubs_sage"].get("rank0_ms_graph_replay") else "n/a", "C2": f2(c["C2"]["ms_per_step"]), "C2F": f2(c["C2"]["ms_per_step_f32"]),
    "C2G": "%.3f" % (c["C2"].get("ms_per_step_graph") or float("nan")), "C2GF": "%.3f" % (c["C2"].get("ms_per_step_graph_f32") or float("nan")),
    "C3": f2(c["C3"]["ms_per_step"]), "C4B": f2(c["C4_bf16_storage"]["ms_per_step"]),
    "R": f2(c["R_net1_step"]["ms_per_step"]), "RE": f2(c["R_net1_step"]["ms_per_step_eager"]),
    "GATF": f2(g["gat_fwd_aggregate"]["avg_launch_ms"]), "GATFA": f2(g["gat_fwd_aggregate"]["frac_algorithmic"]),
    "GATFP": GBs(g["gat_fwd_aggregate"]["traffic"]), "GATFT": f2(g["gat_fwd_aggregate"]["frac_traffic"]),
    "GATB": f2(g["gat_bwd_fused"]["avg_launch_ms"]), "GATBA": f2(g["gat_bwd_fused"]["frac_algorithmic"]),
    "GATBP": GBs(g["gat_bwd_fused"]["traffic"]), "GATBT": f2(g["gat_bwd_fused"]["frac_traffic"]),
    "GCNS": f2(gcn["avg_launch_ms"]), "GCNA": f2(gcn["frac_algorithmic"]), "GCNP": GBs(gcn["traffic"]), "GCNT": f2(gcn["frac_traffic"]),
    "C5F": "%.1f" % g5["gat_fwd_aggregate"]["avg_launch_ms"], "C5B": "%.1f" % g5["gat_bwd_fused"]["avg_launch_ms"],
    "C5FA": f2(g5["gat_fwd_aggregate"]["frac_algorithmic"]), "C5BA": f2(g5["gat_bwd_fused"]["frac_algorithmic"]),
    "C5FP": "%.1f" % (g5["gat_fwd_aggregate"]["traffic"] / 1e9), "C5BP": "%.1f" % (g5["gat_bwd_fused"]["traffic"] / 1e9),
    "C5FT": f2(g5["gat_fwd_aggregate"]["frac_traffic"]), "C5BT": f2(g5["gat_bwd_fused"]["frac_traffic"]),
    "VR": rng(v["hubs_sage"]["per_rank_ms"]), "VBAL": f2(v["hubs_sage"]["balance"]), "VC": f2(v["hubs_sage"]["compute_ceiling"]),
    "V2": f2(v["hubs_sage_by_world"]["2"]["compute_ceiling"]), "V4": f2(v["hubs_sage_by_world"]["4"]["compute_ceiling"]),
    "VG": rng(v["hubs_gat"]["per_rank_ms"]), "VGC": f2(v["hubs_gat"]["compute_ceiling"]),
    "VGR": f2(v["hubs_gat"]["rank0_ms_graph_replay"]) if v["hubs_gat"].get("rank0_ms_graph_replay") else "n/a",
    "VGRC": f2(v["hubs_gat"]["compute_ceiling_graph_replay"]) if v["hubs_gat"].get("compute_ceiling_graph_replay") else "n/a",
    "VROWS": rng(v["rows_sage"]["per_rank_ms"]), "VROWSC": f2(v["rows_sage"]["compute_ceiling"]),
    "VEDG": rng(v["edges_sage"]["per_rank_ms"]), "VEDGC": f2(v["edges_sage"]["compute_ceiling"]),
    "V5": "%.1f–%.1f" % (min(c["C5_1gpu"]["w8_virtual"]["per_rank_ms"]), max(c["C5_1gpu"]["w8_virtual"]["per_rank_ms"])),
    "V5C": f2(c["C5_1gpu"]["w8_virtual"]["compute_ceiling"]),
    "TAG": tag,
}
for key, emu in (("E", v["hubs_sage"].get("emulated_wire", {}).get("by_wire_GBps", {})),
                 ("G", c["C5_1gpu"]["w8_virtual"].get("emulated_wire_one_layer", {}).get("by_wire_GBps", {}))):
    for bw in ("800", "400", "200"):
        e = emu.get(bw) or emu.get(bw + ".0") or {}
        val[key + bw] = f2(e["rank0_ms"]) if e else "n/a"
        val[key + bw + "S"] = "%.1f" % e["speedup_estimate"] if e else "n/a"
src = open(os.path.join(root, "tools", "doc", "DESIGN.in.md")).read()
missing = sorted(set(re.findall(r"@([A-Z0-9]+)@", src)) - set(val))
if missing:
    sys.exit("no value for: " + " ".join(missing))
out = re.sub(r"@([A-Z0-9]+)@", lambda m: val[m.group(1)], src)
open(os.path.join(root, "DESIGN.md"), "w").write(out)
print("DESIGN.md written from", tag, "(%d lines)" % out.count("\n"))
