#!/usr/bin/env python3
"""HIP-event time of the forward projection kernel alone (prepared weight copies, 30 launches after 10), both arithmetics, at one
shape -- for same-box A/B runs of variant libraries (NPI_GNN_LIB).  usage: tools/gemm_time.py [rows [K [N [epi]]]]   (epi: also the row-dot and rank-2 epilogue variants; allocations included)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npi_gnn_amd import functional as NF
dev = torch.device("cuda:0")
M, K, N = (int(v) for v in (sys.argv[1:4] + ["1000000", "256", "256"][len(sys.argv) - 1:]))
g = torch.Generator(device=dev).manual_seed(1)
a = torch.randn(M, K, device=dev, generator=g)
w = torch.randn(K, N, device=dev, generator=g) / K ** 0.5
out = torch.empty(M, N, device=dev)
sc = NF.row_scales(a)
ws3, _ = NF.prepare_weight(w, backward=False)
ws2, _ = NF.prepare_weight(w, backward=False, f16=True)
res = []
for name, fn in (("bf16x3", lambda: NF.linear_fwd(a, w, ws=ws3, out=out)), ("fp16x2", lambda: NF.linear_fwd(a, w, ws=ws2, a_scales=sc, out=out))):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record()
    torch.cuda.synchronize()
    res.append(f"{name} {e0.elapsed_time(e1) / 30:.4f} ms")
# the two epilogue variants of GATConv (fp16 x 2): x W with the row dots of the scores, dC W^T with the rank-2 term
if len(sys.argv) > 4 and sys.argv[4] == "epi" and K == N:
    att = torch.randn(2 * N, device=dev, generator=g) * 0.1
    r0, r1 = torch.randn(M, device=dev, generator=g), torch.randn(M, device=dev, generator=g)
    c0, c1 = torch.randn(K, device=dev, generator=g), torch.randn(K, device=dev, generator=g)
    for name, fn in (("scores", lambda: NF.linear_fwd_scores(a, w, att, a_scales=sc)),
                     ("rank2", lambda: NF.linear_bwd_data_rank2(a, w, r0, r1, c0, c1, dc_scales=sc)),
                     ("scores_bf16x3", lambda: NF.linear_fwd_scores(a, w, att)),
                     ("rank2_bf16x3", lambda: NF.linear_bwd_data_rank2(a, w, r0, r1, c0, c1))):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(f"{name} {e0.elapsed_time(e1) / 20:.4f} ms")
    h, d0, d1 = NF.linear_fwd_scores(a, w, att, a_scales=sc)
    hr = a[:4096].double() @ w.double()
    e_h = float((h[:4096] - hr.float()).abs().max() / hr.abs().max())
    e_d = float((d0[:4096, 0].double() - hr @ att[:N].double()).abs().max() / (hr @ att[:N].double()).abs().max())
    dx = NF.linear_bwd_data_rank2(a, w, r0, r1, c0, c1, dc_scales=sc)
    xr = a[:4096].double() @ w.double().t() + r0[:4096, None].double() * c0.double() + r1[:4096, None].double() * c1.double()
    e_x = float((dx[:4096] - xr.float()).abs().max() / xr.abs().max())
    res.append(f"err scores h {e_h:.1e} dot {e_d:.1e} rank2 {e_x:.1e}")
ref = (a[:2048].double() @ w.double()).float()
err = float((out[:2048] - ref).abs().max() / ref.abs().max())
print(os.path.basename(os.environ.get("NPI_GNN_LIB", "default")), M, K, N, "|", "  ".join(res), f"| fp16x2 max err / max {err:.2e}")
