#!/usr/bin/env python3
"""Error of the two f32 GEMM arithmetics (mode 0 = NPI_GEMM_EXACT_F32, 1 = the default 3-way bf16 split; functional.GEMM_FLAGS, per call)
against an fp64 product, and their speed, on one MI355X.
usage: python tools/gemm_accuracy.py [--rows M] [--hidden F]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npi_gnn_amd import functional as NF  # noqa: E402
from npi_gnn_amd._lib import load  # noqa: E402


def errs(c, ref):
    d = (c.double() - ref).abs()
    return float(d.max() / ref.abs().max()), float(d.norm() / ref.norm())


def timeit(fn, rounds=8):
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts = sorted(ts[1:])
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=200_000 + 77)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--big", type=int, default=1_000_000, help="rows of the timing case")
    a = ap.parse_args()
    lib = load()
    dev = torch.device("cuda:0")
    M, F = a.rows, a.hidden
    g = torch.Generator().manual_seed(0)
    cases = {
        "normal": (torch.randn(M, F, generator=g), torch.randn(F, F, generator=g) / F ** 0.5),
        "wide range (x 2^[-20,20] per element)": (
            torch.randn(M, F, generator=g) * torch.exp2(torch.randint(-20, 21, (M, F), generator=g).float()),
            torch.randn(F, F, generator=g) * torch.exp2(torch.randint(-20, 21, (F, F), generator=g).float())),
        "positive (no cancellation)": (torch.rand(M, F, generator=g), torch.rand(F, F, generator=g)),
    }
    for name, (A, W) in cases.items():
        A, W = A.to(dev), W.to(dev)
        b = torch.zeros(F, device=dev)
        ref_f = A.double() @ W.double()
        ref_b = A.double() @ W.double().t()
        print(f"--- {name}: A [{M},{F}] W [{F},{F}]   (max err / max|ref|,  ||err|| / ||ref||)")
        for mode in (0, 1):
            NF.GEMM_FLAGS = 1 if mode == 0 else 0          # NPI_GEMM_EXACT_F32 / default
            cf = NF.linear_fwd(A, W, b)
            cb = NF.linear_bwd_data(A, W, None)
            torch.cuda.synchronize()
            ef, eb = errs(cf, ref_f), errs(cb, ref_b)
            print(f"  mode {mode}: fwd {ef[0]:.3e} {ef[1]:.3e}   bwd_data {eb[0]:.3e} {eb[1]:.3e}")
        tf = errs((A @ W), ref_f)
        print(f"  torch.matmul f32 (hipBLASLt): fwd {tf[0]:.3e} {tf[1]:.3e}")
        del ref_f, ref_b
    Mb = a.big
    A = torch.randn(Mb, F, generator=g).to(dev)
    W = (torch.randn(F, F, generator=g) / F ** 0.5).to(dev)
    b = torch.randn(F, generator=g).to(dev)
    rs = torch.rand(Mb, generator=g).to(dev)
    fl = 2.0 * Mb * F * F
    for mode in (0, 1):
        NF.GEMM_FLAGS = 1 if mode == 0 else 0
        t1 = timeit(lambda: NF.linear_fwd(A, W, b))
        t2 = timeit(lambda: NF.linear_bwd_data(A, W, rs))
        t3 = timeit(lambda: NF.linear_bwd_weight(A, A, True))
        print(f"mode {mode}: [{Mb},{F}]x[{F},{F}]  fwd {t1:.3f} ms ({fl / t1 / 1e9:.0f} TF/s f32-equivalent)  "
              f"bwd_data {t2:.3f} ms  bwd_weight {t3:.3f} ms", flush=True)
    t = timeit(lambda: torch.addmm(b, A, W))
    print(f"torch.addmm f32: {t:.3f} ms")


if __name__ == "__main__":
    main()
