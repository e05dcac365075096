#!/usr/bin/env python3
"""The projection GEMMs on two fp16 pieces per operand (NPI_GEMM_SPLIT_F16X2) against the default six bf16 products: error against
fp64 and kernel time by HIP events, forward and backward-data, at the C4 shape (1M x 256 x 256) and a few others.
usage: tools/f16x2_probe.py [rows]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npi_gnn_amd import functional as NF

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def err(c, a, w, rows=4096):
    ref = a[:rows].double() @ w.double()
    return float((c[:rows].double() - ref).abs().max() / ref.abs().max())


g = torch.Generator(device=dev).manual_seed(1)
for (m, K, N) in ((M, 256, 256), (M // 4, 128, 128), (200_000, 512, 256), (100_001, 256, 128)):
    a = torch.randn(m, K, device=dev, generator=g)
    w = torch.randn(K, N, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    sc = NF.row_scales(a)
    ws3, _ = NF.prepare_weight(w, backward=False)
    ws2, _ = NF.prepare_weight(w, backward=False, f16=True)
    c3 = NF.linear_fwd(a, w, ws=ws3)
    c2 = NF.linear_fwd(a, w, ws=ws2, a_scales=sc)
    c2n = NF.linear_fwd(a, w, a_scales=sc)                    # planes prepared inside the call
    assert torch.equal(c2, c2n)
    print(f"fwd {m} x {K} x {N}: err bf16x3 {err(c3, a, w):.2e}  fp16x2 {err(c2, a, w):.2e}  | ms bf16x3 {t(lambda: NF.linear_fwd(a, w, ws=ws3)):.3f}"
          f"  fp16x2 {t(lambda: NF.linear_fwd(a, w, ws=ws2, a_scales=sc)):.3f}  row_scales {t(lambda: NF.row_scales(a)):.3f}", flush=True)
    # epilogue: bias + relu + rowscale
    rs = torch.rand(m, device=dev, generator=g) + 0.5
    e3 = NF.linear_fwd(a, w, b, rowscale=rs, relu=True, ws=ws3)
    e2 = NF.linear_fwd(a, w, b, rowscale=rs, relu=True, ws=ws2, a_scales=sc)
    refe = torch.relu(rs[:4096, None].double() * (a[:4096].double() @ w.double()) + b.double())
    print(f"    epilogue (bias, rowscale, relu): bf16x3 {float((e3[:4096].double() - refe).abs().max() / refe.abs().max()):.2e}  "
          f"fp16x2 {float((e2[:4096].double() - refe).abs().max() / refe.abs().max()):.2e}")
    # backward data: dA = dC W^T
    dc = torch.randn(m, N, device=dev, generator=g)
    scd = NF.row_scales(dc)
    _, wb3 = NF.prepare_weight(w, backward=True)
    _, wb2 = NF.prepare_weight(w, backward=True, f16=True)
    d3 = NF.linear_bwd_data(dc, w, ws=wb3)
    d2 = NF.linear_bwd_data(dc, w, ws=wb2, dc_scales=scd)
    refd = dc[:4096].double() @ w.double().t()
    print(f"    bwd_data: err bf16x3 {float((d3[:4096].double() - refd).abs().max() / refd.abs().max()):.2e}  fp16x2 "
          f"{float((d2[:4096].double() - refd).abs().max() / refd.abs().max()):.2e}  | ms bf16x3 {t(lambda: NF.linear_bwd_data(dc, w, ws=wb3)):.3f}  "
          f"fp16x2 {t(lambda: NF.linear_bwd_data(dc, w, ws=wb2, dc_scales=scd)):.3f}", flush=True)
# dynamic range: rows of very different scale, elements over many decades, zero rows, a tiny and a huge row
m, K, N = 8192, 256, 256
a = torch.randn(m, K, device=dev, generator=g) * torch.pow(10.0, torch.rand(m, 1, device=dev, generator=g) * 30 - 15)
a *= torch.pow(10.0, torch.rand(m, K, device=dev, generator=g) * 8 - 4)
a[5] = 0
a[6] *= 1e-30 / a[6].abs().max()
a[7] *= 1e30 / a[7].abs().max()
w = torch.randn(K, N, device=dev, generator=g) * torch.pow(10.0, torch.rand(1, N, device=dev, generator=g) * 6 - 3)
ref = a.double() @ w.double()
c3 = NF.linear_fwd(a, w)
c2 = NF.linear_fwd(a, w, a_scales=NF.row_scales(a))
den = ref.abs().max(dim=1, keepdim=True).values.clamp(min=1e-300)            # per ROW (the rows differ by 40 decades)
dcol = ref.abs().max(dim=0, keepdim=True).values
print(f"wide dynamic range, error relative to the row's largest |C|: bf16x3 {float(((c3.double() - ref).abs() / den).max()):.2e}  "
      f"fp16x2 {float(((c2.double() - ref).abs() / den).max()):.2e}; zero row exact: {bool((c2[5] == 0).all())}; finite: {bool(torch.isfinite(c2).all())}")
