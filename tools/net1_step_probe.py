#!/usr/bin/env python3
"""One captured Net_1 training step (NPInter2 fold 0, first batch of 200 keys) replayed R times between two marker
kernels, for a per-step kernel list.

  run     (under rocprofv3 --kernel-trace):   python3 tools/net1_step_probe.py run [R]
  report  (on the trace csv):                 python3 tools/net1_step_probe.py report <kernel_trace.csv> [R]

`run` also prints the wall time per replayed step and per eager step.  `report` cuts the trace at the markers
(`erfinv` element-wise kernels, used nowhere else) and prints calls per step / mean duration / share per kernel name,
the sum of kernel durations per step and the span per step (span - sum = launch gaps inside the graph)."""
import csv
import os
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def run(R):
    import torch
    import torch.nn.functional as F
    from npi_gnn_amd import net1
    from train_npinter2 import load_fold
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ig, train_keys, train_y, _, _, F_in, _ = load_fold(dev, 0)
    loader = net1.KeyLoader(ig, train_keys[:800], train_y[:800], 200).shuffle(torch.Generator().manual_seed(0))
    model = net1.Net_1(F_in, 2).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=torch.tensor(1e-3, device=dev), weight_decay=1e-3, capturable=True, fused=True)
    ep = net1.GraphedEpoch(model, loader, opt, dev)
    ep()                       # eager epoch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ep.epochs_done = 0
        ep()
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / (5 * len(ep.batches))
    ep.epochs_done = 1
    ep()                       # captures + replays
    torch.cuda.synchronize()
    g = ep.graphs[0]
    marker = torch.rand(1024, device=dev)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    torch.erfinv(marker)
    t0 = time.perf_counter()
    for _ in range(R):
        g.replay()
    torch.cuda.synchronize()
    t_rep = (time.perf_counter() - t0) / R
    torch.erfinv(marker)
    torch.cuda.synchronize()
    d = ep.batches[0]
    print(f"batch 0: {d.x.size(0)} nodes, {d.edge_index.size(1)} directed edges, {d.num_graphs} graphs")
    print(f"training step: eager {t_eager * 1e3:.3f} ms, replayed from the HIP graph {t_rep * 1e3:.3f} ms")
    _ = F


def short(name):
    name = name.replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
    return name[:110]


def report(path, R):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "erfinv" in r["Kernel_Name"].lower()]
    assert len(marks) >= 2, f"markers found: {len(marks)}"
    seg = rows[marks[-2] + 1:marks[-1]]
    per = defaultdict(lambda: [0, 0])
    for r in seg:
        p = per[short(r["Kernel_Name"])]
        p[0] += 1
        p[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy = sum(p[1] for p in per.values())
    span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    print(f"{len(seg) / R:.1f} kernels per step, kernel time {busy / R / 1e3:.1f} us per step, span {span / R / 1e3:.1f} us per step")
    print(f"{'calls/step':>10} {'us/call':>8} {'us/step':>8}  kernel")
    for name, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"{c / R:10.2f} {t / c / 1e3:8.2f} {t / R / 1e3:8.1f}  {name}")
    n = len(seg) // R
    print(f"\n-- the last step in launch order ({n} kernels) --")
    for r in seg[-n:]:
        print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.2f}  {short(r['Kernel_Name'])[:100]}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 100)
    else:
        report(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 100)
