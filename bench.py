#!/usr/bin/env python3
"""bench.py -- edges/sec per GNN layer (fwd+bwd) on the synthetic ncRNA-protein bipartite graph.

Contract: `python bench.py --gpus N --steps K --warmup W`; rank 0 prints ONE JSON line.  For N > 1 the
driver launches `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`; a plain
`python bench.py --gpus N` (no WORLD_SIZE in the environment) starts exactly that command itself as a
child process BEFORE anything in this process touches the GPU, and exits with the child's code.

A "step" is one pass of the hot path over the whole graph: one SAGEConv layer (gather -> segmented mean ->
MFMA projection) forward AND backward (dX, dW, db), the call pattern of reference src/classes.py:62 +
src/train_with_twoDataset.PY:52-54, with x and the graph already resident in HBM.  Workload =
BASELINE.json configs[3] ("C4"): N = 1M nodes, E = 20M directed edges, hidden = 256, fp32 -- it fits one GPU,
so N=1 runs the full graph; N>1 shards the same graph (strong scaling; npi_gnn_amd/dist.py, --partition).

After the headline measurement (N=1 only) the line also carries
  roofline.control_uniform  the same aggregation kernel on a 1M-node / 20M-edge graph whose sources are uniform
                            over the whole 1 GB table (no cache-resident hub side): the un-assisted HBM fraction;
  configs                   the other BASELINE.json configs on this GPU (C1-C3 with their parity error against
                            committed oracle outputs, GCN / GAT at the C4 shape, C5 on one GPU).
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # HBM3E 8.0 TB/s spec (6.29 TB/s is the guide's measured copy rate)
MFMA_F32_PEAK_TF = 157.3       # v_mfma_f32_32x32x2_f32
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16
MFMA_BF16_RANDOM_TF = 1850.0   # the same pipe on random operands (clock held under load; tools/micro/mfma_bf16_rate.hip: 1.59-1.90 PF/s)
SETUP_STEPS = 40               # untimed steps before the warm-up (lazy initialisation, clock ramp: ~0.3 s of load; reported as config.setup_steps)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=20_000_000)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--graph-seed", type=int, default=20260310,
                    help="seed of the synthetic bipartite graph (2 = the graph of configs.C5_1gpu: its PMC passes)")
    ap.add_argument("--conv", choices=["sage", "gcn", "gat"], default="sage")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-nodes", type=int, default=100_000, help="bounded CPU-baseline sample (1/10 scale)")
    ap.add_argument("--cpu-edges", type=int, default=2_000_000)
    ap.add_argument("--cpu-small", action="store_true", help="CPU baseline on the 1/10-scale sample only")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path (npi_gnn_amd.dist) even with one rank")
    ap.add_argument("--partition", choices=["hubs", "rows", "edges"], default="hubs",
                    help="N>1: replicate the protein side and exchange only hub rows (hubs); a plain destination-row "
                         "split with an all-gather of every row (rows); or the north-star's baseline: a slice of the "
                         "edge list per GPU, x replicated, all-reduce of the partial [N,F] sums (edges)")
    ap.add_argument("--no-control", action="store_true", help="skip roofline.control_uniform")
    ap.add_argument("--control-only", action="store_true",
                    help="run only the uniform-source control launches (for a rocprofv3 --pmc pass)")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config block")
    ap.add_argument("--plain-csr", action="store_true",
                    help="build the graphs with every row's entries in edge-list order (default: in column order, CSRGraph(sort_columns=True))")
    ap.add_argument("--no-autotune", action="store_true",
                    help="N > 1: keep the default Schedule instead of timing its alternatives during set-up")
    ap.add_argument("--skip-c5", action="store_true", help="configs block without the 4M / 100M GAT stack")
    ap.add_argument("--capture", action="store_true",
                    help="N=1: capture the step (both streams) into one HIP graph and time replays of it")
    ap.add_argument("--virtual-world", type=int, default=8,
                    help="N=1: time every rank's local work of a W-rank run on this GPU (configs.C4_w<W>_virtual); 0 = skip")
    ap.add_argument("--rank-check", action="store_true",
                    help="every rank prints {rank, world} and exits before any GPU call (launcher test)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n: int, argv) -> int:
    """Start the N ranks as `python -m torch.distributed.run` -- a CHILD process; this process has made no GPU call
    (importing torch does not initialise HIP) and makes none afterwards."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------
# byte / flop accounting
# ---------------------------------------------------------------------------------------------------------
def algorithmic_bytes(nnz_rows_edges: int, n_rows: int, F: int, s: int = 4) -> int:
    """B = E (F s + 4) + N (F s [self row] + F s [write] + 4 [rowptr]);
    the self loop is an ordinary CSR entry here, so its row read + index are the per-node terms."""
    E, N = nnz_rows_edges, n_rows
    return E * (F * s + 4) + N * (2 * F * s + 4)


def gat_bytes(E: int, N: int, F: int, H: int = 1) -> dict:
    """Algorithmic bytes per launch of the two GATConv aggregation kernels (one head, f32, int32 CSR), SURVEY.md 8(d) plus the
    per-entry / per-node scalars of the attention.  nnz = E + N: the self loop is an ordinary entry.
      forward  (npi_gat_aggregate_scores): per entry a gathered h row 4F + col 4 + its score 4H; per node the output row
               4F + rowptr 4 + (m, s) 8H
      backward (npi_gat_backward_fused_heads): per by-source entry a gathered dOut row 4F + col 4 + rowidx 4 + the target's
               packed scalars 16 + dz written 4; per node its own h row 4F + the d h row written 4F + rowptr 4 + a_src 4"""
    nnz = E + N
    return {"gat_fwd_aggregate": nnz * (4 * F + 4 + 4 * H) + N * (4 * F + 4 + 8 * H),
            "gat_bwd_fused": nnz * (4 * F + 4 + 4 + 16 + 4) + N * (8 * F + 8)}


def agg_roofline(events, alg_bytes, traffic, kernel, traffic_source):
    """roofline block of one aggregation kernel from its live event durations"""
    ms = [a.elapsed_time(b) for a, b in events]
    if not ms:
        return None
    avg = sum(ms) / len(ms)
    ach = alg_bytes / (avg * 1e-3) / 1e9
    ft = (traffic / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None
    return {"bound": "hbm", "kernel": kernel, "avg_launch_ms": avg, "launches_timed": len(ms), "algorithmic_bytes_per_launch": alg_bytes,
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac_algorithmic": ach / HBM_PEAK_GBS, "traffic": traffic,
            "frac_traffic": ft, "frac": ft if ft is not None else ach / HBM_PEAK_GBS, "traffic_source": traffic_source}


def parallelism(args, world):
    if world == 1 and not args.force_sharded:
        return "single GPU"
    if args.partition == "hubs":
        return (f"vertex cut x{world}: ncRNA rows owned in strides, protein rows replicated; per direction one "
                "all-gather of protein rows + one reduce-scatter of partial protein sums over RCCL")
    if args.partition == "edges":
        return (f"edge shards x{world} (contiguous slices of the target-sorted entry stream), x replicated; per "
                "direction one RCCL all-reduce of the partial [N,F] sums (the north-star's baseline split)")
    return f"destination-row shards (strided ownership) x{world}, all-gather of every row over RCCL"


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def mem_available_gb() -> float:
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def cpu_baseline(args, ei_full=None):
    """The oracle (PyG-style torch CPU ops: index_select -> index_add_ -> / -> matmul, autograd backward) timed on this
    box's host cores.  The thread count is chosen on the 1/10-scale sample (the index ops stop scaling long before the
    host's thread count); the reported figure is the metric's OWN configuration -- the full C4 graph, N = 1M, E = 20M --
    whenever the host has the memory for its [E+N, F] message tensors (MemAvailable >= 128 GB), else the 1/10 sample."""
    from npi_gnn_amd.synth import bipartite_edge_index
    from oracle import ref_conv as R
    F = args.hidden

    def data(N, E, ei=None):
        if ei is None:
            ei = bipartite_edge_index(N, E, seed=args.graph_seed)
        g = torch.Generator().manual_seed(1)
        x = torch.randn(N, F, generator=g)
        W = (torch.rand(F, F, generator=g) * 2 - 1) / F ** 0.5
        go = torch.randn(N, F, generator=g)
        return x, ei, W, torch.zeros(F), go

    # The index_add_/index_select ops of this path stop scaling long before the host's thread count
    # (2 x 64-core EPYC 9575F: 32 threads 1.9 M edges/s, 256 threads 0.27 M; a sweep of round 1, EXPERIMENTS.md),
    # so the baseline is the best of a short sweep, not "all threads".
    ncpu = os.cpu_count() or 1
    Ns, Es = args.cpu_nodes, args.cpu_edges
    small = data(Ns, Es)
    budget = time.time() + 20.0
    best, best_threads, runs = None, 1, 0
    for nt in sorted({min(t, ncpu) for t in (16, 32, 64)}):
        torch.set_num_threads(nt)
        for it in range(1 + 2):
            t0 = time.time()
            R.sage_layer_fwd_bwd(*small)
            dt = time.time() - t0
            if it >= 1:
                runs += 1
                if best is None or dt < best:
                    best, best_threads = dt, nt
        if time.time() > budget:
            break
    del small
    mem = mem_available_gb()
    sample_small = (f"oracle/ref_conv.sage_layer_fwd_bwd, 1 SAGE layer fwd+bwd, N={Ns} E={Es} F={F} fp32 (1/10-scale C4), best of "
                    f"{runs} timed runs over 16/32/64 torch threads (1 warm-up each)")
    res = {"value": Es / best, "unit": "edges/s", "cores": best_threads, "kind": "port", "sample": sample_small,
           "ran": "1/10-scale sample", "mem_available_gb": round(mem, 1)}
    full = (args.nodes, args.edges) == (1_000_000, 20_000_000) and not args.cpu_small
    if full and mem >= 128.0:
        try:
            torch.set_num_threads(best_threads)
            big = data(args.nodes, args.edges, ei_full)
            times = []
            for it in range(1 + 2):                             # 1 warm-up, best of 2
                t0 = time.time()
                R.sage_layer_fwd_bwd(*big)
                times.append(time.time() - t0)
            del big
            res.update(value=args.edges / min(times[1:]), ran="full C4 configuration",
                       small_sample_edges_per_s=Es / best,
                       sample=f"oracle/ref_conv.sage_layer_fwd_bwd, 1 SAGE layer fwd+bwd on the metric's own configuration: "
                              f"N={args.nodes} E={args.edges} F={F} fp32 (the full C4 graph of the GPU line), best of 2 timed runs "
                              f"after 1 warm-up ({', '.join(f'{t:.1f}' for t in times)} s) at {best_threads} torch threads -- the "
                              f"best of 16/32/64 on the 1/10-scale sample ({Es / best / 1e6:.2f} M edges/s there)")
        except Exception as e:                                  # e.g. the host ran out of memory after all
            res["full_c4_error"] = f"{type(e).__name__}: {e}"[:200]
    elif full:
        res["ran"] = f"1/10-scale sample (MemAvailable {mem:.0f} GB < 128 GB needed for the full C4 message tensors)"
    res["sample"] += f", os.cpu_count()={os.cpu_count()}, cpu='{cpu_model()}'"
    return res


def kernel_source_sha(files=("segsum.hip", "segsum.h", "gat.hip")) -> dict:
    """sha256[:16] of the kernel sources a PMC constant belongs to (profiles/pmc_traffic.json stores the same)"""
    import hashlib
    out = {}
    for f in files:
        try:
            out[f] = hashlib.sha256(open(os.path.join(ROOT, "npi_gnn_amd", "csrc", f), "rb").read()).hexdigest()[:16]
        except OSError:
            out[f] = None
    return out


def pmc_traffic():
    """HBM-side bytes per aggregation launch from the OFFLINE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate
    runs, gfx950 correction per the guide; tools/profile_bench.sh + tools/rocprof_summary.py write this file).  The file
    records the sha of the kernel sources it was measured on: constants of another source are STALE and are not used."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return {}
    now, then = kernel_source_sha(), t.get("source_sha16") or {}
    t["stale"] = [f for f in ("segsum.hip", "segsum.h") if then.get(f) != now[f]]
    t["stale_gat"] = t["stale"] + [f for f in ("gat.hip",) if then.get(f) != now[f]]
    return t


# ---------------------------------------------------------------------------------------------------------
# roofline.control_uniform: the same kernel with no cache-resident hub table
# ---------------------------------------------------------------------------------------------------------
def control_uniform(dev, N, E, F, launches=10):
    import npi_gnn_amd as npi
    from npi_gnn_amd import functional as NF
    g = torch.Generator(device=dev).manual_seed(20260311)
    ei = torch.randint(0, N, (2, E), generator=g, device=dev)          # sources AND targets uniform over all rows
    n_loops = int((ei[0] == ei[1]).sum())                              # dropped by add_remaining_self_loops
    graph = npi.CSRGraph(ei, N)
    del ei
    x = torch.randn(N, F, generator=g, device=dev)
    for _ in range(2):
        NF.segsum(graph, graph.by_dst, x, mean=True)
    ev = []
    NF._PROFILE = ev
    for _ in range(launches):
        NF.segsum(graph, graph.by_dst, x, mean=True)
    NF._PROFILE = None
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    alg = algorithmic_bytes(E - n_loops, N, F)
    ach = alg / (ms * 1e-3) / 1e9
    t = pmc_traffic()
    traffic = t.get("control_uniform_bytes_per_launch") if not t.get("stale") else None
    res = {"workload": f"N={N} E={E} uniform random sources and targets (every row of the {N * F * 4 / 1e9:.2f} GB table "
                       f"equally likely: no cache-resident hub side), F={F} fp32, aggregation launches only",
           "avg_launch_ms": ms, "launches_timed": len(ev), "algorithmic_bytes_per_launch": alg,
           "achieved": ach, "frac_algorithmic": ach / HBM_PEAK_GBS,
           "traffic": traffic,
           "frac_traffic": (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
           "traffic_source": t.get("control_from") if traffic else
           (f"STALE: {t.get('control_from')} was measured on another {t.get('stale')}" if t.get("stale") else None)}
    del graph, x
    torch.cuda.empty_cache()
    return res


# ---------------------------------------------------------------------------------------------------------
# configs block: the other BASELINE.json configs on this GPU
# ---------------------------------------------------------------------------------------------------------
def _timeit(fn, n, warm, rounds=3):
    """ms per call: the best of `rounds` timed regions of n calls each (the configs block is a set of side measurements on a
    box that other jobs may share: one region of one run measured 74 ms per step between regions of 6.9)"""
    for _ in range(warm):
        fn()
    best = None
    for _ in range(rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        best = dt if best is None or dt < best else best
    return best


def _graph_replay_ms(step, n=200, warm=10):
    """ms per replay of `step` captured in a HIP graph: the GPU's own time for the step's launches.  At the sizes of configs
    1-3 an eager step is bounded by the HOST (about 40 launches of 5-25 us of GPU work each behind ~12 us of Python per
    launch), so the eager figure moves with the host's load from region to region; the replayed one does not.  None when the
    step cannot be captured."""
    try:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            step()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        return _timeit(g.replay, n, warm)
    except Exception as e:                                      # noqa: BLE001 -- a side measurement
        sys.stderr.write(f"graph capture of a config step failed: {type(e).__name__}: {e}\n")
        torch.cuda.synchronize()
        return None


def _stack_step(kind, weights, x, graph, dtype=torch.float32, norm=None, att=None, go=None):
    """forward + backward through a stack of convs with relu between them (full batch).  ``go``: the gradient of the stack's
    output, handed to ``backward`` as the headline's step does; None: a mean-square loss on the output drives it (configs 1-3;
    at the C5 size that loss alone is 8 ms of elementwise kernels over [4M, 256] per step)."""
    import npi_gnn_amd as npi
    from npi_gnn_amd import functional as NF
    dev = x.device
    params = [(W.to(dev).to(dtype).requires_grad_(True), b.to(dev).to(dtype).requires_grad_(True)) for W, b in weights]
    atts = [a.to(dev).requires_grad_(True) for a in att] if att else None
    xin = x.to(dtype).requires_grad_(True)

    def step():
        for W, b in params:
            W.grad = b.grad = None
        xin.grad = None
        h = xin
        for k, (W, b) in enumerate(params):
            if kind == "sage":
                h = npi.sage_conv(h, graph, W, b)
            elif kind == "gcn":
                h = NF.gcn_conv(h, None, W, b, norm=norm)
            else:
                h = npi.gat_conv(h, graph, W, atts[k], b, heads=1, relu=True)      # F.relu(conv(h)), fused
                continue
            h = torch.relu(h)
        if go is None:
            h.float().pow(2).mean().backward()
        else:
            h.backward(go)
        return h
    return step


def _stack_forward(kind, weights, x, graph, dtype=torch.float32, norm=None):
    import npi_gnn_amd as npi
    from npi_gnn_amd import functional as NF
    dev = graph.device
    with torch.no_grad():
        h = x.to(dev).to(dtype)
        for W, b in weights:
            W, b = W.to(dev).to(dtype), b.to(dev).to(dtype)
            h = torch.relu(npi.sage_conv(h, graph, W, b) if kind == "sage" else NF.gcn_conv(h, None, W, b, norm=norm))
    return h.float().cpu()


def emulated_wire(probe_args, t1_ms, timeout=300):
    """tools/virtual_rank_probe.py in a CHILD process: rank 0's step of one sharded layer with the exchanges emulated at 800 /
    400 / 200 GB/s (see virtual.StubCollectives(wire_gbps=)); the small exchanges on their own lane (ShardedGraph(small_group=)),
    as the N > 1 run of this file has them"""
    emu = {"assumptions": {"latency_us_per_exchange": 20.0, "held_cus": 16, "what": "duration = latency + wire bytes per rank / B; "
                           "a no-op kernel holds 16 CUs for it on the communicator's stream; two communicators (small exchanges "
                           "on their own)"}, "by_wire_GBps": {}}
    try:
        cp = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "virtual_rank_probe.py")]
                            + list(probe_args) + ["--two-lanes", "--wire-sweep", "800,400,200"], capture_output=True, text=True,
                            timeout=timeout)
        for l in cp.stdout.splitlines():
            if l.startswith("emulated wire"):
                bw, ms = l.split()[2], float(l.split(":")[1].split("ms/step")[0])
                emu["by_wire_GBps"][bw] = {"rank0_ms": ms, "speedup_estimate": t1_ms / ms}
            elif " events " in l:
                emu["rank0_ms_no_wire"] = float(l.split("events")[1].split("ms/step")[0])
        if not emu["by_wire_GBps"]:
            emu["error"] = (cp.stderr or cp.stdout)[-300:]
    except Exception as e:                                  # noqa: BLE001 -- a side measurement
        emu["error"] = f"{type(e).__name__}: {e}"[:300]
    return emu


def virtual_c5(dev, ei5, N5, F5, weights, att, t1_ms, W, ref=None):
    """configs[4] in its 8-GPU form on ONE GPU: every rank's 3-layer step timed alone (collectives = stand-in copies), and --
    ``ref`` = (x, go, out, dX, per-layer parameter gradients) of the single-GPU stack -- the same 8 ranks run once more in
    exact lock step (npi_gnn_amd.virtual.LockStep: true collective results) and compared with it: ``parity``."""
    from npi_gnn_amd import dist as ND
    from npi_gnn_amd.synth import protein_mask
    hub = protein_mask(N5).to(dev)
    per_rank, nnz, coll = [], [], None
    with stub_collectives(W, dev) as stub:
        for r in range(W):
            sg = ND.ShardedGraph(ei5, N5, r, W, dev, hub_mask=hub)
            layers = [ND.ShardedGATLayer(sg, Wk.to(dev), att[k].to(dev), bk.to(dev)) for k, (Wk, bk) in enumerate(weights)]
            x = torch.randn(sg.n_local, F5, device=dev).requires_grad_(True)
            go = torch.randn(sg.n_local, F5, device=dev)

            def step():                                         # from a given output gradient, as T1 (C5_1gpu.ms_per_step)
                for l in layers:
                    l.zero_grad()
                x.grad = None
                h = x
                for l in layers:
                    h = torch.relu(l(h))
                h.backward(go)
            ms, one = time_virtual_rank(step, stub, steps=2, warm=1)
            per_rank.append(ms)
            nnz.append(int(sg.local_nnz))
            coll = coll or one
            del sg, layers, x, go, step
            torch.cuda.empty_cache()
    res = virtual_summary(W, t1_ms, per_rank, nnz, coll, "3 x GATConv 256 (1 head) on the hub cut, N=4M E=100M, per-rank step of the "
                          f"{W}-rank run timed alone on this GPU (collectives = local copies); T1 = C5_1gpu")
    if ref is not None:
        try:
            from npi_gnn_amd.virtual import gat_stack_reference, sharded_stack_errors, stack_distance
            x5, go5 = ref
            params = [(Wk.to(dev), att[k].to(dev), bk.to(dev)) for k, (Wk, bk) in enumerate(weights)]

            def layers_of(ps):
                return lambda sg: [ND.ShardedGATLayer(sg, W_, a_, b_) for W_, a_, b_ in ps]
            # ONE layer (well conditioned): strict
            r1 = gat_stack_reference(ei5, N5, params[:1], x5, go5, relu=False)
            e1 = sharded_stack_errors(W, ei5, N5, hub, layers_of(params[:1]), x5, go5, *r1, dev, relu_between=False)
            p1 = e1.pop("lockstep_passes")
            del r1
            torch.cuda.empty_cache()
            # the 3-layer stack of the timing, against the single-GPU stack AND against the stack's own fp32 noise floor
            rs = gat_stack_reference(ei5, N5, params, x5, go5, relu=True)
            fls = [stack_distance(gat_stack_reference(ei5, N5, params, x5, go5, relu=True, permute_seed=sd), rs) for sd in (5, 6)]
            floor = {k: max(f[k] for f in fls) for k in fls[0]}
            torch.cuda.empty_cache()
            e3 = sharded_stack_errors(W, ei5, N5, hub, layers_of(params), x5, go5, *rs, dev, relu_between=True)
            p3 = e3.pop("lockstep_passes")
            del rs
            ratio = {k: (v / floor[k] if floor[k] > 0 else None) for k, v in e3.items() if k.endswith(".l2") and not k.startswith("out")}
            res["parity"] = {
                "parity_max_err": max(list(e1.values()) + [e3["out"], e3["out.l2"]]),
                "one_layer": {"by_tensor": e1, "lockstep_passes": p1},
                "stack": {"by_tensor": e3, "fp32_noise_floor": floor, "err_over_floor": ratio,
                          "max_err_over_floor": max(v for v in ratio.values() if v is not None), "lockstep_passes": p3},
                "against": f"the single-GPU GATConv on the whole graph; the {W} ranks in exact lock step on this GPU (true all-gather / "
                           "reduce-scatter / all-reduce results); every rank's rows of out and dX and the all-reduced dW / d att / db, max "
                           "over ranks; <tensor>: max |diff| / max |reference|, <tensor>.l2: ||diff|| / ||reference||.  parity_max_err = "
                           "every tensor of ONE layer and the output of the 3-layer stack.  The stack's GRADIENTS are reported against its "
                           "own fp32 noise floor = the distance between two single-GPU runs that differ only in the order of the edge list (max of two "
                           "such runs; err_over_floor on the L2 figures) "
                           "(the backward of a deep random GAT stack is ill-conditioned: 1e-4 .. 1e-3 on this data whoever computes it)"}
            res["parity_max_err"] = res["parity"]["parity_max_err"]
        except Exception as e:                                  # noqa: BLE001
            res["parity"] = {"parity_max_err": None, "error": f"{type(e).__name__}: {e}"[:300]}
    return res


def run_configs(dev, args, c4):
    """ms per full-batch step (fwd+bwd over the layer stack) and, for C1-C3, the max error against the oracle outputs
    committed under tests/golden/ (made by tests/golden/make_golden.py / make_rpi7317.py from the CPU oracle)."""
    import npi_gnn_amd as npi
    from npi_gnn_amd import functional as NF
    out = {}
    G = os.path.join(ROOT, "tests", "golden")

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as e:                                  # the headline line must survive a failing extra
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()

    fx = torch.load(os.path.join(G, "npinter2_graph.pt"), map_location="cpu", weights_only=False)
    x, ei = fx["x"], fx["edge_index"].long()
    graph = npi.CSRGraph(ei.to(dev), x.size(0))
    _ = graph.by_src
    norm = NF.GCNNorm(graph)
    rows = fx["rows"]
    shape = f"NPInter2 graph N={x.size(0)} E={ei.size(1)}"

    def c1():
        h = _stack_forward("gcn", fx["gcn64"], x, graph, norm=norm)
        return {"workload": f"{shape}, 2 x GCNConv 178->64->64 fp32, full batch",
                "ms_per_step": _timeit(_stack_step("gcn", fx["gcn64"], x.to(dev), graph, norm=norm), 30, 5),
                "ms_per_step_graph": _graph_replay_ms(_stack_step("gcn", fx["gcn64"], x.to(dev), graph, norm=norm)),
                "parity_max_abs_err": float((h[rows] - fx["gcn64_out"]).abs().max()), "parity": "oracle (unpinned: GCNConv)"}

    def c2():
        h = _stack_forward("sage", fx["sage_weights"], x, graph, dtype=torch.bfloat16)
        ref = fx["sage3_out"]
        sb = _stack_step("sage", fx["sage_weights"], x.to(dev), graph, dtype=torch.bfloat16)
        sf = _stack_step("sage", fx["sage_weights"], x.to(dev), graph)
        # eager: host-bound at this size (see _graph_replay_ms) -- the two storage types are timed alternately, best region each
        eb = ef = None
        for _ in range(3):
            tb, tf = _timeit(sb, 50, 5, rounds=1), _timeit(sf, 50, 5, rounds=1)
            eb, ef = (tb if eb is None else min(eb, tb)), (tf if ef is None else min(ef, tf))
        return {"workload": f"{shape}, 3 x SAGEConv 178->128->128->128, bf16 storage / f32 accumulate, full batch",
                "ms_per_step": eb, "ms_per_step_f32": ef,
                "ms_per_step_graph": _graph_replay_ms(sb), "ms_per_step_graph_f32": _graph_replay_ms(sf),
                "note": "ms_per_step*: eager (host-bound: ~40 launches per step); ms_per_step_graph*: the same step replayed from "
                        "a HIP graph = the GPU's time",
                "parity_max_err_rel_to_max": float((h[rows] - ref).abs().max() / ref.abs().max()),
                "parity": "fp32 oracle, bf16 tolerance"}

    def c3():
        p = os.path.join(G, "rpi7317_graph.pt")
        f3 = torch.load(p, map_location="cpu", weights_only=False)
        x3, ei3 = f3["x"], f3["edge_index"].long()
        g3 = npi.CSRGraph(ei3.to(dev), x3.size(0))
        _ = g3.by_src
        n3 = NF.GCNNorm(g3)
        h = _stack_forward("gcn", f3["gcn256"], x3, g3, norm=n3)
        return {"workload": f"RPI7317 graph N={x3.size(0)} E={ei3.size(1)} (7,317 positives + 7,317 seeded negatives), "
                            "3 x GCNConv 178->256->256->256 fp32, full batch",
                "ms_per_step": _timeit(_stack_step("gcn", f3["gcn256"], x3.to(dev), g3, norm=n3), 30, 5),
                "ms_per_step_graph": _graph_replay_ms(_stack_step("gcn", f3["gcn256"], x3.to(dev), g3, norm=n3)),
                "parity_max_abs_err": float((h[f3["rows"]] - f3["gcn256_out"]).abs().max()),
                "parity": "oracle (unpinned: GCNConv)"}

    def r_step():
        # the reference's REAL regime (SURVEY 8(a) "R"): one Net_1 training step -- forward, nll_loss, backward, Adam -- on a
        # batch of 200 enclosing subgraphs of NPInter2 fold 0, extracted on the device; eager and replayed from a HIP graph
        import torch.nn.functional as F_
        from npi_gnn_amd import net1
        from npi_gnn_amd.subgraph import InteractionGraph
        fz = torch.load(os.path.join(G, "npinter2_folds.pt"), map_location="cpu", weights_only=False)
        fb = fz["fold0"]
        pairs, label, Nn = fz["pairs"].long(), fz["label"].long(), fz["num_nodes"]
        test = torch.cat([fb["test_pos"], fb["test_neg"]]).long()
        usable = ~torch.isin(pairs[:, 0] * Nn + pairs[:, 1], test[:, 0] * Nn + test[:, 1])
        feat = torch.cat([fb["node2vec"], fz["kmer"]], dim=1)
        ig = InteractionGraph(pairs.to(dev), usable.to(dev), feat.to(dev), num_nodes=Nn)
        keys, yk = pairs[usable][:800].to(dev), label[usable][:800].to(dev)
        loader = net1.KeyLoader(ig, keys, yk, 200)
        torch.manual_seed(0)
        model = net1.Net_1(feat.size(1) + 1, 2).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=torch.tensor(1e-3, device=dev), weight_decay=1e-3, capturable=True, fused=True)
        ep = net1.GraphedEpoch(model, loader, opt, dev)
        ep()                                                    # eager epoch (4 batches)
        d0 = ep.batches[0]

        def eager():
            opt.zero_grad()
            F_.nll_loss(model(d0), d0.y).backward()
            opt.step()
        ms_eager = _timeit(eager, 100, 10)
        ep()                                                    # captures every batch's step, replays it once
        ms_replay = _timeit(ep.graphs[0].replay, 200, 10)
        return {"workload": f"NPInter2 fold 0, first batch of 200 enclosing subgraphs ({d0.x.size(0)} nodes, "
                            f"{d0.edge_index.size(1)} directed edges, F = {d0.x.size(1)}): one Net_1 training step "
                            "(forward, nll_loss, backward, Adam), fp32",
                "ms_per_step": ms_replay, "ms_per_step_eager": ms_eager,
                "note": "ms_per_step: the step replayed from a HIP graph (net1.GraphedEpoch); the reference logs 1413.5 s for "
                        "its 50-epoch fold = 4,200 such steps + evaluations (examples/train_npinter2.py --capture: 4.8 s)"}

    guarded("C1", c1)
    guarded("C2", c2)
    guarded("C3", c3)
    guarded("R_net1_step", r_step)
    del graph, norm

    # GCN / GAT layer at the C4 shape, on the headline graph
    g4, x4, go4, F = c4["graph"], c4["x"], c4["go"], c4["F"]
    E4 = c4["E"]
    gen = torch.Generator().manual_seed(11)

    pmc = pmc_traffic()
    N4 = x4.size(0)

    def pmc_of(key, stale_key="stale"):
        if pmc.get(stale_key):
            return None, f"STALE: measured on another {pmc.get(stale_key)}"
        return pmc.get(key), (pmc.get("gat_from") if key.startswith("gat") else pmc.get("gcn_from"))

    def gcn_c4():
        conv = npi.GCNConv(F, F).to(dev)
        n4 = NF.GCNNorm(g4)
        xx = x4.detach().requires_grad_(True)

        def step():
            conv.weight.grad = conv.bias.grad = xx.grad = None
            NF.gcn_conv(xx, None, conv.weight, conv.bias, norm=n4).backward(go4)
        ms = _timeit(step, 10, 3)
        ev = []
        NF._PROFILE = ev
        for _ in range(5):
            step()
        NF._PROFILE = None
        torch.cuda.synchronize()
        # SURVEY 8(d) + one f32 weight (the symmetric normalisation) per entry, self loops included
        alg = algorithmic_bytes(E4, N4, F) + (E4 + N4) * 4
        tr, src = pmc_of("gcn_segsum_bytes_per_launch")
        return {"workload": f"C4 graph, 1 x GCNConv {F}->{F} fp32 fwd+bwd", "ms_per_step": ms, "edges_per_s": E4 / ms * 1e3,
                "roofline": agg_roofline(ev, alg, tr, "segsum_kernel<f32, 4, 1, W_ARRAY> (one launch: cut rows are finished inside it), avg of the forward and the "
                                                      "backward launch (the latter co-resident with dW)", src)}

    def gat_c4():
        conv = npi.GATConv(F, F, heads=1).to(dev)
        xx = x4.detach().requires_grad_(True)

        def step():
            for p in conv.parameters():
                p.grad = None
            xx.grad = None
            conv(xx, g4).backward(go4)
        ms = _timeit(step, 10, 3)
        tags = {}
        NF._PROFILE_TAGS = tags
        for _ in range(5):
            step()
        NF._PROFILE_TAGS = None
        torch.cuda.synchronize()
        gb = gat_bytes(E4, N4, F)
        roof = {}
        for tag, kern in (("gat_fwd_aggregate", "segsum_kernel<f32, 4, 1, W_GAT_DST_PRE>: weighted aggregation, scores read back"),
                          ("gat_bwd_fused", "segsum_kernel<f32, 4, 1, W_GAT_SRC_FUSED>: by-source aggregation + SDDMM in one gather pass")):
            tr, src = pmc_of(tag + "_bytes_per_launch", "stale_gat")
            roof[tag] = agg_roofline(tags.get(tag, []), gb[tag], tr, kern, src)
        by_heads = {}
        for Hh in (2, 4, 8):                                   # the same layer width as 2 / 4 / 8 heads of 128 / 64 / 32 channels
            cv = npi.GATConv(F, F // Hh, heads=Hh).to(dev)

            def hstep(cv=cv):
                for p in cv.parameters():
                    p.grad = None
                xx.grad = None
                cv(xx, g4).backward(go4)
            by_heads[str(Hh)] = _timeit(hstep, 5, 2)
            del cv
        return {"workload": f"C4 graph, 1 x GATConv {F}->{F} (1 head) fp32 fwd+bwd", "ms_per_step": ms,
                "edges_per_s": E4 / ms * 1e3, "ms_per_step_by_heads": by_heads, "roofline": roof}

    def c4_bf16():
        """the headline layer with bf16 STORAGE (features, weights, gradients; f32 accumulation inside the kernels, as config C2):
        information only -- the metric's precision is f32 and `value` is the f32 number"""
        bf = torch.bfloat16
        conv = npi.SAGEConv(F, F).to(dev)
        ref = conv(x4, g4).detach()
        convb = npi.SAGEConv(F, F).to(dev)
        convb.load_state_dict(conv.state_dict())
        convb = convb.to(bf)
        xb = x4.detach().to(bf).requires_grad_(True)
        gob = go4.to(bf)

        def step():
            convb.weight.grad = convb.bias.grad = xb.grad = None
            convb(xb, g4).backward(gob)
        ms = _timeit(step, 10, 3)
        ev = []
        NF._PROFILE = ev
        for _ in range(5):
            step()
        NF._PROFILE = None
        torch.cuda.synchronize()
        dev_rel = float((convb(xb, g4).detach().float() - ref).abs().max() / ref.abs().max())
        # SURVEY 8(d) with s = 2 bytes per stored element: per edge F s + 4 = 516 B, per node 2 F s + 4 = 1,028 B
        alg = algorithmic_bytes(E4, N4, F, s=2)
        return {"workload": f"C4 graph, 1 x SAGEConv {F}->{F} fwd+bwd, bf16 storage / f32 accumulate (NOT the metric's precision)",
                "ms_per_step": ms, "edges_per_s": E4 / ms * 1e3, "max_dev_from_f32_output_rel": dev_rel,
                "roofline": agg_roofline(ev, alg, None, "segsum_kernel<bf16, 4, 1, W_NONE>, avg of the forward and the backward "
                                         "launch (the latter co-resident with dW); NOT the metric's precision", "no PMC pass "
                                         "for the bf16 kernels: algorithmic bytes (516 B per edge, 1,028 B per node) only")}

    guarded("gcn_c4", gcn_c4)
    guarded("gat_c4", gat_c4)
    guarded("C4_bf16_storage", c4_bf16)

    if not args.skip_c5:
        def c5():
            from npi_gnn_amd.synth import bipartite_edge_index
            c4.clear()                                             # release the C4 graph and features first
            torch.cuda.empty_cache()
            N5, E5, F5 = 4_000_000, 100_000_000, 256
            ei5 = bipartite_edge_index(N5, E5, seed=2).to(dev)
            g5 = npi.CSRGraph(ei5, N5, sort_columns=not args.plain_csr)
            _ = g5.by_src
            weights = [((torch.randn(F5, F5, generator=gen) / 16), torch.zeros(F5)) for _ in range(3)]
            att = [torch.randn(1, 1, 2 * F5, generator=gen) * 0.1 for _ in range(3)]
            x5 = torch.randn(N5, F5, generator=gen).to(dev)
            go5 = torch.randn(N5, F5, generator=gen).to(dev)
            ms_loss = _timeit(_stack_step("gat", weights, x5, g5, att=att), 3, 1, rounds=2)      # as rounds 1-3 measured it
            st = _stack_step("gat", weights, x5, g5, att=att, go=go5)
            ms = _timeit(st, 3, 1, rounds=2)
            tags = {}
            NF._PROFILE_TAGS = tags
            st()
            NF._PROFILE_TAGS = None
            torch.cuda.synchronize()
            gb = gat_bytes(E5, N5, F5)
            # HBM bytes of the two aggregation launches at THIS size, from their own PMC passes (tools/profile_all.sh: bench.py
            # --conv gat on this very graph); the algorithmic bytes count every gathered row, the hub rows served from L2 included
            roof = {}
            for tag in gb:
                tr = None if pmc.get("stale_gat") else pmc.get("c5_" + tag + "_bytes_per_launch")
                src = pmc.get("c5_from") if tr else (f"STALE: measured on another {pmc.get('stale_gat')}" if pmc.get("stale_gat")
                                                     else "no PMC pass at this size: algorithmic bytes only")
                roof[tag] = agg_roofline(tags.get(tag, []), gb[tag], tr, "as configs.gat_c4.roofline, at the C5 size (3 launches, "
                                         "one per layer)", src)
            res5 = {"workload": f"C5 synthetic bipartite N={N5} E={E5}, 3 x GATConv 256 (1 head) fp32 fwd+bwd, ONE GPU",
                    "ms_per_step": ms, "edge_layers_per_s": 3 * E5 / ms * 1e3, "ms_per_step_with_mse_loss": ms_loss,
                    "step": "forward + backward of the three layers from a given output gradient, as the headline's step "
                            "(ms_per_step_with_mse_loss: with output.pow(2).mean() driving the backward -- 8 ms of elementwise "
                            "kernels over [4M, 256] -- which is how rounds 1-3 timed this config)",
                    "roofline": roof}
            del st, g5, go5
            x5 = x5.detach()
            ref = (x5, torch.randn(N5, F5, generator=gen).to(dev)) if args.virtual_world > 1 else None     # inputs of the parity check
            torch.cuda.empty_cache()
            if args.virtual_world > 1:
                # BASELINE.json configs[4] in its 8-GPU form, rank by rank on this GPU: the same 3-layer GATConv stack on the
                # hub cut, a rank's output rows being the next layer's input rows (collectives = local copies, as C4_w8_virtual),
                # then the same ranks in exact lock step against the single-GPU stack (parity)
                try:
                    res5["w8_virtual"] = virtual_c5(dev, ei5, N5, F5, weights, att, ms, args.virtual_world, ref=ref)
                except Exception as e:
                    res5["w8_virtual"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            del ei5, ref, x5
            torch.cuda.empty_cache()
            if args.virtual_world == 8 and "error" not in res5.get("w8_virtual", {"error": 1}):
                # ONE GATConv layer of this size on rank 0 of 8 with the exchanges emulated (the child builds its own graph of the
                # same shape); the single-GPU figure beside it is a third of the 3-layer step
                res5["w8_virtual"]["emulated_wire_one_layer"] = emulated_wire(
                    ["--conv", "gat", "--nodes", str(N5), "--edges", str(E5), "--steps", "10"], ms / 3, timeout=600)
            return res5
        guarded("C5_1gpu", c5)
    return out


# ---------------------------------------------------------------------------------------------------------
# N > 1: is the sharded layer RIGHT?  (a scaling number without this is a claim about speed only)
# ---------------------------------------------------------------------------------------------------------
def sharded_parity(dev, args, world, sg, layer, x, go, ei, x_full, go_full, W, bias, att=None, samples=256):
    """After the timed region every rank runs the SINGLE-GPU layer (the plain conv of this package, itself held to the
    oracle by the -m gpu tests) over the WHOLE graph on its own GPU and compares the rows it owns of the sharded output
    and of dX, and the all-reduced dW / db; SAGEConv rows are also checked against the formula in fp64 torch ops on
    ``samples`` of the rank's rows.  Errors are relative to the largest reference magnitude; MAX over ranks."""
    import npi_gnn_amd as npi
    N, F = args.nodes, args.hidden
    kind = args.conv
    conv = {"sage": npi.SAGEConv, "gcn": npi.GCNConv, "gat": npi.GATConv}[kind](F, F).to(dev)
    with torch.no_grad():
        conv.weight.copy_(W)
        conv.bias.copy_(bias)
        if att is not None:
            conv.att.copy_(att)
    ei_dev = ei.to(dev)
    graph = npi.CSRGraph(ei_dev, N)
    xr = x_full.to(dev).requires_grad_(True)
    ref = conv(xr, graph)
    ref.backward(go_full.to(dev))
    layer.zero_grad()
    x.grad = None
    out = layer(x)
    out.backward(go)
    torch.cuda.synchronize()
    if args.partition == "edges":
        rows = torch.arange(sg.lo, sg.hi, device=dev)
        dx_ref, dx = xr.grad, x.grad                               # x is replicated: its gradient is complete on every rank
    else:
        rows = sg.own
        dx_ref, dx = xr.grad[rows], x.grad

    def rel(a, b):
        return float((a.detach() - b.detach()).abs().max() / b.detach().abs().max().clamp(min=1e-30))
    err = {"out": rel(out, ref[rows]), "dX": rel(dx, dx_ref), "dW": rel(layer.weight.grad, conv.weight.grad),
           "db": rel(layer.bias.grad, conv.bias.grad)}
    if att is not None:
        err["datt"] = rel(layer.att.grad, conv.att.grad)
    if kind == "sage" and rows.numel():
        g = torch.Generator(device=dev).manual_seed(7 + int(rows[0]))
        pos = torch.randperm(rows.numel(), generator=g, device=dev)[:samples]      # positions in this rank's row order
        pick, order = rows[pos].sort()
        pos = pos[order]
        src, dst = ei_dev[0], ei_dev[1]
        sel = torch.isin(dst, pick) & (src != dst)
        slot = torch.searchsorted(pick, dst[sel])
        acc = xr.detach()[pick].double().index_add_(0, slot, xr.detach()[src[sel]].double())
        cnt = torch.bincount(slot, minlength=pick.numel()).double() + 1.0
        want = (acc / cnt.view(-1, 1)) @ W.to(dev).double() + bias.to(dev).double()
        err["out_rows_fp64_formula"] = float((out.detach()[pos].double() - want).abs().max() / want.abs().max())
    names = sorted(err)
    v = torch.tensor([err[k] for k in names], dtype=torch.float64, device=dev)
    if world > 1 or dist_is_up():
        import torch.distributed as dist
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
    err = {k: float(e) for k, e in zip(names, v.tolist())}
    return {"parity_max_err": max(err.values()), "by_tensor": err,
            "against": "the single-GPU layer of this package run over the whole graph on every rank's GPU (its rows of out and "
                       "dX, the all-reduced dW / db), max over ranks, relative to the largest reference magnitude; "
                       f"out_rows_fp64_formula: {samples} rows per rank against mean(x_j) @ W + b in fp64 torch ops"}


# ---------------------------------------------------------------------------------------------------------
# virtual world: the W shards of the multi-GPU path, one after the other on this ONE GPU
# ---------------------------------------------------------------------------------------------------------
def stub_collectives(W, dev):
    """npi_gnn_amd.virtual.StubCollectives with the stand-in copies on a stream of their own: a collective is issued when its
    input is ready and the compute streams wait for it where they consume its result -- the dependency graph RCCL's stream
    gives the real run (the partial side runs beside the all-gather stand-in, the projection beside the reduce-scatter one)."""
    from npi_gnn_amd.virtual import StubCollectives
    return StubCollectives(W, copy_stream=torch.cuda.Stream(device=dev))


def virtual_summary(W, t1, per_rank, nnz, coll, what):
    worst = max(per_rank)
    wire = sum(v["wire_bytes_per_rank"] for v in coll.values())
    budget = t1 / 6.0 - worst
    return {"what": what, "world": W, "t1_ms": t1, "per_rank_ms": per_rank, "per_rank_entries": nnz,
            "balance": sum(per_rank) / len(per_rank) / worst, "compute_ceiling": t1 / worst,
            "bytes_per_collective": coll, "wire_bytes_per_rank_per_step": wire,
            "exposed_budget_ms_for_6x": budget,
            "implied_bus_GBps": {"all_communication_hidden_under_T1_over_6": wire / (t1 / 6.0 * 1e-3) / 1e9,
                                 "no_overlap_inside_the_exposed_budget": (wire / (budget * 1e-3) / 1e9) if budget > 0 else None}}


def time_virtual_rank(step, stub, steps=5, warm=3):
    for _ in range(warm):
        step()
    stub.log.clear()
    step()                                                  # the collectives of ONE step, by kind
    one = {k: dict(v) for k, v in stub.log.items()}
    best = None
    for _ in range(3):                                      # best of three regions (a shared box: see _timeit)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        best = dt if best is None or dt < best else best
    return best, one


def virtual_world(dev, args, ei_dev, c4, t1_sage_ms, W=8, only=None):
    """SURVEY.md 8(e), what one GPU can measure of the W-GPU run: every rank's LOCAL work (its shard's kernels, host
    launch work included) timed alone on this GPU with the collectives replaced by local copies of the same shapes, for
    the three partitions (SAGEConv) and the sharded GATConv.  From it: the load balance, the compute-side ceiling of the
    speed-up (T1 / max_r T_r: what W GPUs reach with free communication), the bytes every collective moves, the
    communication time a >= 6x speed-up leaves (T1 / 6 - max_r T_r) and the bus bandwidth that implies.
    ``only="hubs_sage"``: the SAGEConv vertex cut alone (the smaller worlds of the 1 / 2 / 4 / 8 curve)."""
    import npi_gnn_amd as npi
    from npi_gnn_amd import dist as ND
    from npi_gnn_amd.synth import protein_mask
    N, E, F = args.nodes, args.edges, args.hidden
    gen = torch.Generator().manual_seed(3)
    Wm = ((torch.rand(F, F, generator=gen) * 2 - 1) / F ** 0.5).to(dev)
    bias = ((torch.rand(F, generator=gen) * 2 - 1) / F ** 0.5).to(dev)
    att = (torch.randn(1, 1, 2 * F, generator=gen) * 0.1).to(dev)
    # T1 of the plain GATConv on this graph (the SAGE T1 is the headline measurement)
    conv = npi.GATConv(F, F, heads=1).to(dev)
    xx = c4["x"].detach().requires_grad_(True)

    def gat_step():
        for p in conv.parameters():
            p.grad = None
        xx.grad = None
        conv(xx, c4["graph"]).backward(c4["go"])
    t1_gat = _timeit(gat_step, 5, 2) if only is None else None
    del conv, xx

    out = {}
    with stub_collectives(W, dev) as stub:
        hub = protein_mask(N).to(dev)
        in_count = torch.bincount(ei_dev[1][ei_dev[0] != ei_dev[1]], minlength=N)
        kinds = ("hubs_sage", "hubs_gat", "rows_sage", "edges_sage") if only is None else (only,)
        res = {k: ([], [], None) for k in kinds}
        for r in range(W):
            for partition in (("hubs", "rows", "edges") if only is None else ("hubs",)):
                if partition == "edges":
                    sg = ND.EdgeShardedGraph(ei_dev, N, r, W, dev, in_count=in_count)
                    x = torch.randn(N, F, device=dev).requires_grad_(True)
                    go = torch.randn(sg.hi - sg.lo, F, device=dev)
                    layers = [("edges_sage", ND.EdgeShardedSAGELayer(sg, Wm, bias))]
                else:
                    sg = ND.ShardedGraph(ei_dev, N, r, W, dev, hub_mask=hub if partition == "hubs" else None)
                    x = torch.randn(sg.n_local, F, device=dev).requires_grad_(True)
                    go = torch.randn(sg.n_local, F, device=dev)
                    layers = [(partition + "_sage", ND.ShardedSAGELayer(sg, Wm, bias))]
                    if partition == "hubs" and only is None:
                        layers.append(("hubs_gat", ND.ShardedGATLayer(sg, Wm, att, bias)))
                for key, layer in layers:
                    def step(layer=layer, x=x, go=go):
                        layer.zero_grad()
                        x.grad = None
                        layer(x).backward(go)
                    ms, coll = time_virtual_rank(step, stub)
                    res[key][0].append(ms)
                    res[key][1].append(int(sg.local_nnz))
                    if r == 0:
                        res[key] = (res[key][0], res[key][1], coll)
                del sg, x, go, layers, layer
                torch.cuda.empty_cache()
        notes = {"hubs_sage": "SAGEConv, protein rows replicated (vertex cut): all-gather + reduce-scatter of hub rows per direction",
                 "rows_sage": "SAGEConv, destination-row shards: all-gather of every row per direction",
                 "edges_sage": "SAGEConv, the north-star's literal split: a slice of the edge list per GPU, x replicated, "
                               "all-reduce of the partial [N,F] sums per direction",
                 "hubs_gat": "GATConv (1 head), vertex cut with the cross-rank softmax"}
        for key, (ms, nnz, coll) in res.items():
            out[key] = virtual_summary(W, t1_gat if key == "hubs_gat" else t1_sage_ms, ms, nnz, coll, notes[key])
    if "hubs_sage" in out and only is None:
        # The same rank step with the exchanges EMULATED: in front of every stand-in copy a kernel that computes nothing holds 16
        # CUs (64 KB of LDS each, as a collective's resident workgroups hold theirs) for 20 us + wire bytes per rank / B -- an
        # estimate of the W-GPU step under two stated assumptions (the rate B a GPU sustains over its xGMI links for these
        # exchanges; the CUs RCCL's kernel sits on), NOT a measurement of xGMI.  Rank 0 only (the ranks are balanced to 1 %).
        # (in a CHILD process: this one has created a dozen HIP streams by now, more than the hardware has queues, and a stand-in
        # that holds its queue for hundreds of us then also holds whatever compute stream shares that queue)
        out["hubs_sage"]["emulated_wire"] = emulated_wire(["--conv", "sage", "--steps", "30"], t1_sage_ms)
    if "hubs_gat" in out and W == 8 and (N, E, F) == (1_000_000, 20_000_000, 256):
        # the GATConv rank step is ~110 launches of a few us: eager it is bounded by the HOST and moves with the box's CPU
        # (1.7-2.2 ms).  Its GPU time: rank 0's step replayed from a HIP graph, stand-in copies on the compute stream (capture with
        # the copy stream's nested forks takes the HIP runtime down at capture_end, hence a CHILD process; None if it fails)
        rep = None
        try:
            cp = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "virtual_rank_probe.py"),
                                 "--conv", "gat", "--capture", "--inline-copies", "--steps", "50"], capture_output=True, text=True,
                                timeout=240)
            m = [l for l in cp.stdout.splitlines() if "events" in l and "capture=True" in l]
            if m:
                rep = float(m[-1].split("events")[1].split("ms/step")[0])
        except Exception as e:                                  # noqa: BLE001 -- a side measurement
            sys.stderr.write(f"graph replay of a GATConv rank step failed: {type(e).__name__}: {e}\n")
        out["hubs_gat"]["rank0_ms_graph_replay"] = rep
        out["hubs_gat"]["compute_ceiling_graph_replay"] = (t1_gat / rep) if rep else None
        out["hubs_gat"]["graph_replay_note"] = ("rank 0's step replayed from a HIP graph (stand-in copies on the compute stream): the "
                                                "GPU's time; per_rank_ms is the eager step, which the host bounds at this size")
    out["note"] = ("one GPU, ranks run one after the other; collectives are local copies of the same shapes on a stream of their own "
                   "(issued when their input is ready, waited for where their result is consumed: the real run's dependency graph), "
                   "so per_rank_ms is local compute + host launch work only; wire bytes: all-gather / reduce-scatter of S bytes move "
                   "S (W-1)/W per rank, an all-reduce 2 S (W-1)/W; N > 1 itself is NOT measured here")
    return out


# ---------------------------------------------------------------------------------------------------------
def dist_is_up() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rank_check:
        print(json.dumps({"rank_check": True, "rank": rank, "world": world, "local_rank": local_rank,
                          "cuda_initialized": torch.cuda.is_initialized()}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NPI_BENCH_RCCL_SOLO=1 (with --force-sharded): a world of ONE through everything the N > 1 run goes through -- a real RCCL
    # process group and the second communicator, every collective of the layer (dist.ALWAYS_COMMUNICATE), the set-up's
    # candidates, the all-reduced timings -- the one-GPU box's rehearsal of the node run (tests/test_dist_rccl.py)
    rccl = world > 1 or (args.force_sharded and os.environ.get("NPI_BENCH_RCCL_SOLO") == "1")
    if rccl:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            from npi_gnn_amd import dist as _ND
            _ND.ALWAYS_COMMUNICATE = True
        else:
            dist.init_process_group("nccl", device_id=dev)

    import npi_gnn_amd as npi
    from npi_gnn_amd import functional as NF
    from npi_gnn_amd.synth import bipartite_edge_index

    N, E, F = args.nodes, args.edges, args.hidden
    if args.control_only:
        print(json.dumps({"control_uniform": control_uniform(dev, N, E, F)}), flush=True)
        return
    sharded = world > 1 or args.force_sharded
    fallback = None                                             # set when the sharded pre-flight step fell back (below)
    ei = bipartite_edge_index(N, E, seed=args.graph_seed)
    g = torch.Generator().manual_seed(1)
    x_full = torch.randn(N, F, generator=g)
    W = ((torch.rand(F, F, generator=g) * 2 - 1) / F ** 0.5)
    bias = ((torch.rand(F, generator=g) * 2 - 1) / F ** 0.5)
    go_full = torch.randn(N, F, generator=g)

    seg_events = []                       # (start, end) HIP events around every npi_segsum launch
    gemm_events = []                      # (name, flops, start, end) around every projection GEMM
    NF._PROFILE = None
    c4 = {}

    if not sharded:
        t0 = time.time()
        # every row's entries in column order (CSRGraph(sort_columns=): a second key for the build's sort, once per graph;
        # nothing for this SAGEConv line, 0.8-1.2 % for the GATConv configs -- EXPERIMENTS A22); --plain-csr: list order
        graph = npi.CSRGraph(ei.to(dev), N, sort_columns=not args.plain_csr)
        _ = graph.by_src
        torch.cuda.synchronize()
        t_build = time.time() - t0
        conv = {"sage": npi.SAGEConv, "gcn": npi.GCNConv, "gat": npi.GATConv}[args.conv](F, F).to(dev)
        with torch.no_grad():
            conv.weight.copy_(W)
            conv.bias.copy_(bias)
        x = x_full.to(dev).requires_grad_(True)
        go = go_full.to(dev)
        norm = NF.GCNNorm(graph) if args.conv == "gcn" else None
        c4.update(graph=graph, x=x, go=go, F=F, E=E)

        def step():
            for p in conv.parameters():
                p.grad = None
            x.grad = None
            if norm is not None:
                out = NF.gcn_conv(x, None, conv.weight, conv.bias, norm=norm)
            else:
                out = conv(x, graph)
            out.backward(go)
        seg_launch_bytes = [algorithmic_bytes(E, N, F)]
    else:
        from npi_gnn_amd import dist as ND
        from npi_gnn_amd.schedule import CONSERVATIVE, DEFAULT
        from npi_gnn_amd.synth import protein_mask
        t0 = time.time()
        # every rank ships only ITS slice of the edge list to its GPU; the partitioner routes the edges (dist.route_edges)
        ei_mine = ei[:, rank * E // world: (rank + 1) * E // world] if world > 1 else ei
        att_full = torch.randn(1, 1, 2 * F, generator=g) * 0.1
        # a second communicator for the small exchanges (per-row scalars, the softmax's MAX, parameter-gradient sums): on the main
        # one they would queue behind the hub-row tables issued before them (ShardedGraph(small_group=))
        small_group = None
        if rccl:
            import torch.distributed as dist
            try:
                small_group = dist.new_group()
            except Exception as e:                              # noqa: BLE001 -- one communicator then (config.communicators says so)
                sys.stderr.write(f"rank {rank}: no second communicator ({type(e).__name__}: {e}); every exchange on the first\n")

        def build_sharded(schedule, one_communicator=False):
            if args.partition == "edges":
                sg_ = ND.EdgeShardedGraph(ei_mine, N, rank, world, dev, sliced=world > 1)
                layer_ = ND.EdgeShardedSAGELayer(sg_, W.to(dev), bias.to(dev))
                x_ = x_full.to(dev).requires_grad_(True)        # x is REPLICATED in this split
                return sg_, layer_, x_, sg_.shard(go_full).to(dev), [algorithmic_bytes(sg_.local_nnz, N, F)]
            sg_ = ND.ShardedGraph(ei_mine, N, rank, world, dev, hub_mask=protein_mask(N) if args.partition == "hubs" else None,
                                  sliced=world > 1, schedule=schedule,
                                  small_group=None if (schedule is CONSERVATIVE or one_communicator) else small_group)
            layer_ = {"sage": ND.ShardedSAGELayer, "gcn": ND.ShardedGCNLayer}[args.conv](sg_, W.to(dev), bias.to(dev)) \
                if args.conv != "gat" else ND.ShardedGATLayer(sg_, W.to(dev), att_full.to(dev), bias.to(dev))
            x_ = sg_.shard(x_full).to(dev).requires_grad_(True)  # this rank's rows: its ncRNAs, then its proteins
            # per direction this rank launches side A (its rows) and, with hubs, side B (partial hub sums)
            nbytes = [algorithmic_bytes(sg_.A.nnz_max - sg_.n_local, sg_.n_local, F)]
            if sg_.B is not None:
                nbytes.append(algorithmic_bytes(sg_.B.nnz_max, sg_.part.hub_rows, F) - sg_.part.hub_rows * F * 4)
            return sg_, layer_, x_, sg_.shard(go_full).to(dev), nbytes
        sg, layer, x, go, seg_launch_bytes = build_sharded(DEFAULT)

        def step():
            layer.zero_grad()
            x.grad = None
            out = layer(x)
            out.backward(go)

        # Pre-flight: ONE step of the layer as configured.  If any rank raises (a code path that no test box could run: the
        # schedule with the third stream, the merge-free hub layout, the split projection, the rank-2 store epilogue have only
        # met RCCL through this bench), every rank falls back to the round-2 schedule -- classic layout, one extra stream, one
        # GEMM per direction -- rebuilds its shard and says so in the line (`config.fallback`), instead of leaving the scaling
        # run without a number.  The parity block below checks whichever schedule ran.
        err = None
        try:
            if os.environ.get("NPI_BENCH_INJECT_FAILURE") == "1":
                raise RuntimeError("injected pre-flight failure (NPI_BENCH_INJECT_FAILURE=1)")
            step()
            torch.cuda.synchronize()
        except Exception as e:                                  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"[:300]
        bad = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=dev)
        if rccl:
            import torch.distributed as dist
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad) != 0:
            fallback = err or "another rank failed its pre-flight step"
            del sg, layer, x, go
            torch.cuda.empty_cache()
            sg, layer, x, go, seg_launch_bytes = build_sharded(CONSERVATIVE)      # an argument of the shard, no process-wide switch
        torch.cuda.synchronize()
        t_build = time.time() - t0
        # Set-up, N > 1 only: three arrangements whose worth depends on what RCCL's kernels do beside ours -- nothing one GPU can
        # tell (EXPERIMENTS A9, A16): the projection GEMMs on 16 CUs fewer (a persistent GEMM whose workgroup finds its CU held
        # by a collective starts late with its full share of tiles) the hub rows of dAgg projected first, and every exchange on ONE communicator.  Each candidate
        # is built, run 5 + 2 x 8 steps between barriers (MAX over ranks), and the fastest becomes THE schedule of the timed region;
        # every rank takes the same decision from the same all-reduced numbers.  `config.autotune` lists what was measured.
        autotune = None
        if fallback is None and args.partition == "hubs" and args.conv in ("sage", "gcn") and not args.no_autotune and (
                rccl or os.environ.get("NPI_BENCH_AUTOTUNE_SOLO") == "1"):
            cands = {"default": DEFAULT, "gemm_reserve_cus=16": DEFAULT.but(gemm_reserve_cus=16),
                     "early_hub_gather": DEFAULT.but(early_hub_gather=True),
                     "gemm_reserve_cus=16,early_hub_gather": DEFAULT.but(gemm_reserve_cus=16, early_hub_gather=True),
                     "one communicator": DEFAULT}                   # (the default has the small exchanges on a second one)
            autotune, best = {}, ("default", None)
            try:
                for name, sch in cands.items():
                    if name != "default":
                        del sg, layer, x, go
                        torch.cuda.empty_cache()
                        sg, layer, x, go, seg_launch_bytes = build_sharded(sch, name == "one communicator")
                    for _ in range(5):
                        step()
                    regions = []
                    for _ in range(2):                          # the better of two regions: the first one after a rebuild is noisy
                        if rccl:
                            dist.barrier()
                        torch.cuda.synchronize()
                        a0 = time.perf_counter()
                        for _ in range(8):
                            step()
                        torch.cuda.synchronize()
                        tt = torch.tensor([(time.perf_counter() - a0) / 8 * 1e3], dtype=torch.float64, device=dev)
                        if rccl:
                            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                        regions.append(float(tt.item()))
                    autotune[name] = min(regions)
                    if best[1] is None or autotune[name] < best[1]:
                        best = (name, autotune[name])
                if best[0] != list(cands)[-1]:                       # the last candidate is the one that is built right now
                    del sg, layer, x, go
                    torch.cuda.empty_cache()
                    sg, layer, x, go, seg_launch_bytes = build_sharded(cands[best[0]], best[0] == "one communicator")
                autotune = {"ms_per_step": autotune, "chosen": best[0]}
            except Exception as e:                              # noqa: BLE001 -- keep the default, say why
                autotune = {"error": f"{type(e).__name__}: {e}"[:300], "chosen": "default"}
                sg, layer, x, go, seg_launch_bytes = build_sharded(DEFAULT)

    def barrier():
        if rccl:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # steady state first: the first launches of a process pay lazy initialisation (kernel code upload, allocator growth, clock
    # ramp) that W = 3 warm-up steps do not always cover -- the first timed region measured 0.07-0.1 ms per step above the
    # following ones.  SETUP_STEPS untimed steps belong to the set-up, then the contract's W warm-up steps and K timed ones.
    for _ in range(SETUP_STEPS):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    captured = False
    if args.capture and not sharded:
        # the same kernels on the same two streams, recorded once: the replayed step does not depend on the host's pace
        eager_step = step
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            eager_step()
        step = gr.replay
        captured = True
        step()
        barrier()
    NF._PROFILE = seg_events if not captured else None
    NF._PROFILE_GEMM = gemm_events if not captured else None
    comm_events = []                      # (tag, start, end) around every wait on a collective (sharded path only)
    if sharded:
        from npi_gnn_amd import dist as ND_
        ND_._COMM_PROFILE = comm_events
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    NF._PROFILE = None
    NF._PROFILE_GEMM = None
    if sharded:
        ND_._COMM_PROFILE = None
    # two more regions of exactly K steps, reported BESIDE the contract's measurement (never instead of it): the spread says
    # whether the timed region met a quiet GPU (one run of this bench on a shared box measured 9.7 ms between two of 7.0)
    repeats = []
    if world == 1:
        for _ in range(2):
            barrier()
            r0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            repeats.append((time.perf_counter() - r0) / args.steps * 1e3)
    if rccl:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    value = E * args.steps / dt
    parity = None
    if sharded:
        try:
            parity = sharded_parity(dev, args, world, sg, layer, x, go, ei, x_full, go_full, W, bias,
                                    att=att_full if args.conv == "gat" and args.partition != "edges" else None)
        except Exception as e:                                  # on every rank alike (same code, same data)
            parity = {"parity_max_err": None, "error": f"{type(e).__name__}: {e}"[:300]}

    # dominant kernel: segsum (fwd + bwd launches have the same algorithmic bytes when F_in == F_out)
    seg_ms = [s.elapsed_time(e) for s, e in seg_events]
    seg_avg_ms = sum(seg_ms) / max(len(seg_ms), 1)
    alg_bytes = sum(seg_launch_bytes) / len(seg_launch_bytes)          # average over the launches of one direction
    achieved = alg_bytes / (seg_avg_ms * 1e-3) / 1e9 if seg_ms else 0.0
    pmc = pmc_traffic()
    traffic = None
    if not sharded and args.conv == "sage" and (N, E, F) == (1_000_000, 20_000_000, 256) and not pmc.get("stale"):
        traffic = pmc.get("segsum_kernel_bytes_per_launch")
    frac_alg = achieved / HBM_PEAK_GBS
    frac_traffic = (traffic / (seg_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and seg_ms) else None

    # aggregation-only rate beside the layer total, and the projection against the MFMA peak
    seg_total_ms = sum(seg_ms)
    gem = {}
    for name, flops, e0, e1 in gemm_events:
        g_ = gem.setdefault(name, [0.0, 0.0, 0])
        g_[0] += flops
        g_[1] += e0.elapsed_time(e1)
        g_[2] += 1
    # the two GEMMs that run alone on the chip give the MFMA figure; dW is listed with the duration it has while it
    # shares every CU with the backward aggregation
    solo = [v for k, v in gem.items() if k != "bwd_weight"]
    solo_flops, solo_ms = sum(v[0] for v in solo), sum(v[1] for v in solo)
    solo_tf = (solo_flops / (solo_ms * 1e-3) / 1e12) if solo_ms else None
    # the communication that was NOT hidden -- how long a stream stood still at each wait on a collective
    exposed = {}
    for tag, e0, e1 in comm_events:
        exposed[tag] = exposed.get(tag, 0.0) + e0.elapsed_time(e1)
    exchange = None
    if sharded:
        exchange = {"exposed_ms_per_step": sum(exposed.values()) / args.steps,
                    "by_collective_ms_per_step": {k: v / args.steps for k, v in sorted(exposed.items())},
                    "note": "rank 0; stall of the waiting stream at each collective (HIP events around work.wait())"}
    split_on = bool(int(npi.load().npi_gemm_mode(-1)))
    extra = {
        "exchange": exchange,
        "aggregation_only": {"edges_per_s": (E * args.steps / (seg_total_ms * 1e-3)) if seg_ms and world == 1 else None,
                             "ms_per_step": seg_total_ms / args.steps if seg_ms else None,
                             "note": "gather + segmented reduction, forward + transposed backward launches of one layer"},
        "projection": {
            "bound": "mfma",
            "kernels": "fwd + bwd_data: gemm_split_ws_kernel (f32 operands split 3-way into bf16, six "
                       "v_mfma_f32_32x32x16_bf16 per f32 product, f32 accumulate)" if split_on else
                       "fwd + bwd_data: exact-f32 v_mfma_f32_32x32x2_f32 kernels",
            "achieved_f32_equivalent": solo_tf, "unit": "TFLOP/s",
            # the pipe the kernel runs on: six bf16 MFMA flops are issued per f32-equivalent flop
            "achieved": (6.0 * solo_tf) if (solo_tf and split_on) else solo_tf,
            "peak": MFMA_BF16_PEAK_TF if split_on else MFMA_F32_PEAK_TF,
            "frac": ((6.0 * solo_tf / MFMA_BF16_PEAK_TF) if split_on else (solo_tf / MFMA_F32_PEAK_TF)) if solo_tf else None,
            "frac_note": "issued bf16-MFMA flops (6 x f32-equivalent) / dense bf16 MFMA peak" if split_on else
                         "f32 flops / f32 MFMA peak",
            "f32_equivalent_vs_f32_mfma_peak": (solo_tf / MFMA_F32_PEAK_TF) if solo_tf else None,
            # what the bf16 pipe sustains on random operands (the chip lowers its clock under MFMA load): measured with a bare
            # v_mfma_f32_32x32x16_bf16 loop, tools/micro/mfma_bf16_rate.hip, 1.59-1.90 PF/s box to box
            "sustained_random_operands": {"peak": MFMA_BF16_RANDOM_TF, "unit": "TFLOP/s", "source": "offline micro-benchmark",
                                          "frac": (6.0 * solo_tf / MFMA_BF16_RANDOM_TF) if (solo_tf and split_on) else None},
            "per_gemm_ms": {k: v[1] / v[2] for k, v in gem.items()},
            "per_gemm_tflops_f32_equivalent": {k: v[0] / (v[1] * 1e-3) / 1e12 for k, v in gem.items() if v[1] > 0},
            "note": "bwd_weight (gemm_dw_split_kernel: both operands split on the fly) is timed while it shares the CUs "
                    "with the backward aggregation on a second stream; alone it takes 0.93 ms"},
    }
    res = None
    if rank == 0:
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                # two fractions of the 8 TB/s HBM peak, both from the same live launch duration:
                "frac_algorithmic": frac_alg,      # SURVEY 8(d) bytes (every gathered row counted, no cache credit) / time
                "frac_traffic": frac_traffic,      # bytes that crossed the L2 <-> fabric boundary (PMC) / time
                "frac": frac_traffic if frac_traffic is not None else frac_alg,
                "frac_basis": ("traffic: PMC bytes / live duration / peak -- the HBM-roofline fraction; frac_algorithmic "
                               "exceeds it (and can exceed 1) because gathers of the 100k protein rows are served by the "
                               "XCD L2s / Infinity Cache and never reach HBM") if frac_traffic is not None else
                              "algorithmic bytes / live duration / peak (no current PMC pass on file for this configuration; "
                              "this figure counts every gathered row as an HBM read and can exceed 1 when caches serve gathers)",
                "traffic": traffic if traffic else ("stale" if pmc.get("stale") else None),
                "traffic_source": (f"OFFLINE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE separately; FETCH_SIZE x "
                                   f"{pmc.get('fetch_scale')} gfx950 calibration), {pmc.get('from')}, measured "
                                   f"on this very source (sha {pmc.get('source_sha16')}) -- not measured in this run")
                if traffic else (f"STALE: {pmc.get('from')} was measured on another {pmc.get('stale')}; frac falls back to "
                                 "frac_algorithmic" if pmc.get("stale") else None),
                "kernel": "segsum_kernel (one launch per aggregation: rows cut by an item boundary are finished inside it), avg of fwd and bwd launches",
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": seg_avg_ms,
                "launches_timed": len(seg_ms)}
        res = {
            "metric": "edges/sec per GNN layer (fwd+bwd)", "value": value, "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "ms_per_step_repeats": repeats,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"C4 synthetic ncRNA-protein bipartite graph, N={N} nodes, E={E} directed edges "
                                   f"(both directions, Zipf-skewed protein side), 1 {args.conv.upper()}Conv layer "
                                   f"{F}->{F} fp32, fwd+bwd incl. dX/dW/db, graph+features resident in HBM",
                       "parallelism": parallelism(args, world), "hip_graph_replay": captured, "setup_steps": SETUP_STEPS,
                       "csr_build_s": round(t_build, 4), "csr_sorted_columns": (not args.plain_csr) if not sharded else False,
                       "fallback": fallback,
                       "autotune": autotune if sharded else None,
                       "communicators": (2 if getattr(sg, "small_group", None) is not getattr(sg, "group", None) else 1) if sharded else None},
            "roofline": roof,
        }
        res.update(extra)
        if parity is not None:
            res["parity_max_err"] = parity["parity_max_err"]
            res["parity"] = parity
    if rccl:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    # ---- extras after the timed region (one GPU only): control, per-config block, CPU baseline -------------
    if world == 1 and not sharded:
        del x_full, go_full
        if not args.no_control and args.conv == "sage":
            try:
                res["roofline"]["control_uniform"] = control_uniform(dev, N, E, F)
            except Exception as e:
                res["roofline"]["control_uniform"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        vw = None
        if args.virtual_world > 1 and args.conv == "sage":
            try:
                ei_dev = ei.to(dev)
                vw = virtual_world(dev, args, ei_dev, c4, ms_per_step, args.virtual_world)
                # the smaller worlds of the metric's 1 / 2 / 4 / 8 curve, vertex cut only: compute-side ceilings per world size
                curve = {}
                for w_ in (2, 4):
                    if w_ < args.virtual_world:
                        v_ = virtual_world(dev, args, ei_dev, c4, ms_per_step, w_, only="hubs_sage")["hubs_sage"]
                        curve[str(w_)] = {k: v_[k] for k in ("per_rank_ms", "balance", "compute_ceiling", "wire_bytes_per_rank_per_step")
                                          if k in v_}
                if "hubs_sage" in vw:
                    curve[str(args.virtual_world)] = {k: vw["hubs_sage"][k] for k in ("per_rank_ms", "balance", "compute_ceiling",
                                                                                      "wire_bytes_per_rank_per_step") if k in vw["hubs_sage"]}
                vw["hubs_sage_by_world"] = curve
                del ei_dev
            except Exception as e:
                vw = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
        if not args.no_configs and args.conv == "sage":
            del graph, x, go, step, conv
            res["configs"] = run_configs(dev, args, c4)
        if vw is not None:
            res.setdefault("configs", {})[f"C4_w{args.virtual_world}_virtual"] = vw
    if not args.no_cpu_baseline:
        # rank 0, after the process group is gone (the other ranks have left): the N > 1 line carries the CPU path timed in the
        # same run as well -- the same sample as the N = 1 line, so the two are comparable
        res["cpu_baseline"] = cpu_baseline(args, ei)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
