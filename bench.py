#!/usr/bin/env python3
"""bench.py -- edges/sec per GNN layer (fwd+bwd) on the synthetic ncRNA-protein bipartite graph.

Contract: `python bench.py --gpus N --steps K --warmup W`; rank 0's LAST stdout line is ONE compact strict-JSON object
(< 8 KB: metric, value, roofline, cpu_baseline, parity, one number per side config); everything longer -- per-tensor tables,
per-rank arrays, prose -- goes to `bench_detail.json` in the working directory.  For N > 1 the driver launches
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`; a plain `python bench.py --gpus N` (no
WORLD_SIZE in the environment) starts exactly that command itself as a child process BEFORE anything in this process touches
the GPU, and exits with the child's code.

A "step" is one pass of the hot path over the whole graph: one SAGEConv layer (gather -> segmented mean -> MFMA projection)
forward AND backward (dX, dW, db), the call pattern of reference src/classes.py:62 + src/train_with_twoDataset.PY:52-54, with x
and the graph already resident in HBM.  Workload = BASELINE.json configs[3] ("C4"): N = 1M nodes, E = 20M directed edges,
hidden = 256, fp32 -- it fits one GPU, so N=1 runs the full graph; N>1 shards the same graph (strong scaling;
npi_gnn_amd/dist.py, --partition).

The default N=1 run: graph build, 40 + W + K steps, the roofline block (live HIP-event durations of the aggregation launches),
the parity of exactly what was timed (every row of out / dX and dW, db against the formulas in fp64 torch ops), the
cache-hostile control, one number per other BASELINE config inside a wall-clock budget, and the CPU path on the host cores.
`--extras` adds the lab harness (tools/bench_extras.py: per-head sweeps, the virtual worlds of every partition, the emulated
wire, the C5 stack in its 8-rank form with its parity) -- detail file only.

N > 1: every rank launched by torch.distributed.run is a SUPERVISOR that never touches the GPU; it starts the rank's worker as
a child process.  A worker that fails (or a pre-flight phase that does not finish in time) fails the whole attempt: every
supervisor ends its own worker and starts a FRESH one on the conservative schedule (`config.fallback` says so) -- no in-process
recovery after a HIP fault, no collective that a failed rank would have to join.
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
SETUP_STEPS = 40               # untimed steps before the warm-up (lazy initialisation, clock ramp: ~0.3 s of load; reported as config.setup_steps)
LINE_CAP = 8192                # hard cap of the final stdout line (the driver's parser lost a 20.9 KB line in round 4)
DETAIL_FILE = "bench_detail.json"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=20_000_000)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--graph-seed", type=int, default=20260310,
                    help="seed of the synthetic bipartite graph (2 = the C5 graph: its PMC passes)")
    ap.add_argument("--conv", choices=["sage", "gcn", "gat"], default="sage")
    ap.add_argument("--storage", choices=["f32", "bf16"], default="f32",
                    help="N=1: bf16 = features, weights and gradients stored as bf16 (f32 accumulation inside the kernels, as "
                         "BASELINE configs[1]) -- NOT the metric's precision: for the rocprofv3 passes of the bf16 kernels")
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
    ap.add_argument("--no-configs", action="store_true", help="skip the one-number-per-config summary")
    ap.add_argument("--no-parity", action="store_true", help="N=1: skip the fp64 check of the timed configuration")
    ap.add_argument("--configs-budget", type=float, default=45.0,
                    help="wall-clock seconds the default run may spend on the per-config summary (entries that no longer fit are skipped)")
    ap.add_argument("--extras", action="store_true",
                    help="the lab harness (tools/bench_extras.py) without a budget: per-head sweeps, every partition's virtual "
                         "world, emulated wire, the C5 stack on the host-generated graph with its 8-rank parity")
    ap.add_argument("--plain-csr", action="store_true",
                    help="build the graphs with every row's entries in edge-list order (default: in column order, CSRGraph(sort_columns=True))")
    ap.add_argument("--no-autotune", action="store_true",
                    help="N > 1: keep the default Schedule instead of timing its alternatives during set-up")
    ap.add_argument("--conservative", action="store_true", help="N > 1: the round-2 schedule (what a second attempt runs)")
    ap.add_argument("--skip-c5", action="store_true", help="configs summary without the 4M / 100M GAT stack")
    ap.add_argument("--capture", action="store_true",
                    help="N=1: capture the step (both streams) into one HIP graph and time replays of it")
    ap.add_argument("--virtual-world", type=int, default=8,
                    help="N=1: time every rank's local work of a W-rank run on this GPU (configs.w<W>_hubs_sage_*); 0 = skip")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="N=1: do not run the two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) that measure the aggregation "
                         "kernel's HBM-side bytes in THIS run; roofline.traffic then comes from profiles/pmc_traffic.json (offline)")
    ap.add_argument("--setup-steps", type=int, default=SETUP_STEPS, help="untimed steps before the warm-up")
    ap.add_argument("--detail-file", default=DETAIL_FILE,
                    help="where the full record goes ('' = nowhere; the live PMC child passes write none)")
    ap.add_argument("--rank-check", action="store_true",
                    help="every rank prints {rank, world} and exits before any GPU call (launcher test)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# launcher and supervisor (neither touches the GPU)
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


def rendezvous_dir(world: int = 0) -> str:
    """One directory per LAUNCH, the same for every rank of it.  NPI_BENCH_RDV when set (a worker gets it from its supervisor;
    a test may name one).  Otherwise, world > 1: the ranks are siblings started by one torch.distributed.run agent, so the
    agent's pid + its start time (a pid can be reused, the pair cannot) + MASTER_PORT name the launch; world == 1: a fresh
    temporary directory."""
    d = os.environ.get("NPI_BENCH_RDV")
    if not d:
        tmp = os.environ.get("TMPDIR", "/tmp")
        if world <= 1:
            import tempfile
            d = tempfile.mkdtemp(prefix="npi_bench_", dir=tmp)
        else:
            ppid = os.getppid()
            try:
                born = open(f"/proc/{ppid}/stat").read().rsplit(")", 1)[1].split()[19]      # field 22: starttime
            except (OSError, IndexError):
                born = "0"
            d = os.path.join(tmp, f"npi_bench_{ppid}_{born}_{os.environ.get('MASTER_PORT', '0')}")
        os.environ["NPI_BENCH_RDV"] = d
    os.makedirs(d, exist_ok=True)
    return d


def _write_atomic(path: str, text: str) -> None:
    """a marker file appears with its content (another rank may read it the moment it exists)"""
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def supervise(argv, rank: int, world: int) -> int:
    """The process torch.distributed.run started for this rank: it makes NO GPU call.  It runs the rank's worker (this file with
    NPI_BENCH_WORKER=1) as a child and watches the launch's rendezvous directory: attempt a has FAILED for everybody as soon as
    any rank's worker exits non-zero (its supervisor drops `fail_<a>_<rank>`) or a worker does not report the end of its set-up
    phase (`ok_<a>_<rank>`: pre-flight step + schedule candidates done) within NPI_BENCH_PREFLIGHT_TIMEOUT seconds.  Then every
    supervisor ends ITS OWN worker (exact pid) and starts attempt 1: a fresh process on the conservative schedule, rendezvous
    through a fresh file store.  A second failure is final (exit code 1)."""
    rdv = rendezvous_dir(world)
    limit = float(os.environ.get("NPI_BENCH_PREFLIGHT_TIMEOUT", "900"))
    total = float(os.environ.get("NPI_BENCH_TOTAL_TIMEOUT", "3000"))     # a worker that sits in a collective for ever ends here
    for attempt in (0, 1):
        env = dict(os.environ, NPI_BENCH_WORKER="1", NPI_BENCH_ATTEMPT=str(attempt), NPI_BENCH_RDV=rdv)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env)
        t0 = time.time()
        failed = None
        while True:
            rc = p.poll()
            if rc == 0:
                if world <= 1:
                    import shutil
                    shutil.rmtree(rdv, ignore_errors=True)
                return 0
            if rc is not None:
                failed = f"rank {rank}: worker exit code {rc}"
            elif any(f.startswith(f"fail_{attempt}_") and ".tmp" not in f for f in os.listdir(rdv)):
                failed = "another rank's worker failed"
            elif not os.path.exists(os.path.join(rdv, f"ok_{attempt}_{rank}")) and time.time() - t0 > limit:
                failed = f"rank {rank}: set-up phase not finished after {limit:.0f} s"
            elif time.time() - t0 > total:
                failed = f"rank {rank}: worker still running after {total:.0f} s"
            if failed:
                break
            time.sleep(0.2)
        why = failed
        errf = os.path.join(rdv, f"err_{attempt}_{rank}")
        if os.path.exists(errf):
            why += ": " + open(errf).read()[:300]
        _write_atomic(os.path.join(rdv, f"fail_{attempt}_{rank}"), why)
        if p.poll() is None:                                    # this rank's worker, by pid: it may sit in a collective for ever
            p.kill()
            p.wait()
        sys.stderr.write(f"[bench supervisor] attempt {attempt} failed ({why})"
                         + ("; starting a fresh worker on the conservative schedule\n" if attempt == 0 else "; giving up\n"))
    return 1


def worker_note(kind: str, text: str = "") -> None:
    """ok_<attempt>_<rank> / err_<attempt>_<rank> in the rendezvous directory (a worker under a supervisor only)"""
    if os.environ.get("NPI_BENCH_WORKER") != "1":
        return
    a, r = os.environ.get("NPI_BENCH_ATTEMPT", "0"), os.environ.get("RANK", "0")
    _write_atomic(os.path.join(rendezvous_dir(), f"{kind}_{a}_{r}"), text)


def first_attempt_failure():
    """why attempt 0 failed, for `config.fallback` of the second attempt's line"""
    rdv = rendezvous_dir()
    why = [open(os.path.join(rdv, f)).read()[:300] for f in sorted(os.listdir(rdv)) if f.startswith("fail_0_") and ".tmp" not in f]
    first_hand = [w for w in why if w and not w.startswith("another rank")]  # the rank that failed, not the ones that followed
    return (first_hand or why or ["attempt 0 failed"])[0]


def fake_worker(rank: int, world: int) -> int:
    """NPI_BENCH_FAKE_WORKER=fail0 (tests/test_bench_launcher.py, no GPU): attempt 0 -- the last rank fails its pre-flight, the
    others report ok and then sit in a 'collective' for ever; attempt 1 -- every rank finishes, rank 0 prints a line."""
    attempt = int(os.environ.get("NPI_BENCH_ATTEMPT", "0"))
    if attempt == 0:
        if rank == world - 1:
            worker_note("err", "RuntimeError: injected pre-flight failure")
            return 3
        worker_note("ok")
        time.sleep(600)
        return 0
    worker_note("ok")
    if rank == 0:
        print(json.dumps({"fake": True, "attempt": attempt, "fallback": first_attempt_failure()}), flush=True)
    return 0


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
      forward  (npi_gat_aggregate_fused, round 5: the statistics inside the launch, the scores recomputed from the gathered
               rows): per entry a gathered h row 4F + col 4; per node the output row 4F + rowptr 4 + a_dst 4 + (m, s) written 8H
      backward (npi_gat_backward_fused_heads): per by-source entry a gathered dOut row 4F + col 4 + rowidx 4 + the target's
               packed scalars 16 + dz written 4; per node its own h row 4F + the d h row written 4F + rowptr 4 + a_src 4"""
    nnz = E + N
    return {"gat_fwd_aggregate": nnz * (4 * F + 4) + N * (4 * F + 4 + 4 + 8 * H),
            "gat_bwd_fused": nnz * (4 * F + 4 + 4 + 16 + 4) + N * (8 * F + 8)}


def roofline_rates(alg_bytes, traffic, avg_ms):
    """(achieved, achieved_traffic, frac_algorithmic, frac_traffic) of one aggregation launch: GB/s of SURVEY 8(d)'s algorithmic
    bytes and of the PMC (L2-miss) bytes over the same live duration, and both as fractions of the HBM peak.  The roofline object
    carries all four, so that `achieved / peak == frac_algorithmic` and `achieved_traffic / peak == frac_traffic == frac`."""
    if not avg_ms:
        return 0.0, None, 0.0, None
    ach = alg_bytes / (avg_ms * 1e-3) / 1e9
    ach_t = (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None
    return ach, ach_t, ach / HBM_PEAK_GBS, (ach_t / HBM_PEAK_GBS if ach_t is not None else None)


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
    box's host cores: a bounded sample of about 30 s.  The thread count is chosen on the 1/10-scale sample (the index ops stop
    scaling long before the host's thread count: 2 x 64-core EPYC 9575F, 32 threads 1.9 M edges/s, 256 threads 0.27 M -- a sweep
    of round 1, EXPERIMENTS.md); the reported figure is the metric's OWN configuration -- the full C4 graph, one warm-up and one
    timed run -- whenever the host has the memory for its [E+N, F] message tensors (MemAvailable >= 128 GB), else the sample."""
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

    ncpu = os.cpu_count() or 1
    Ns, Es = args.cpu_nodes, args.cpu_edges
    small = data(Ns, Es)
    budget = time.time() + 8.0
    best, best_threads, runs = None, 1, 0
    for nt in sorted({min(t, ncpu) for t in (16, 32, 64)}):
        torch.set_num_threads(nt)
        for it in range(2):                                     # one warm-up, one timed
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
    where = f"os.cpu_count()={os.cpu_count()}, cpu='{cpu_model()}'"
    res = {"value": Es / best, "unit": "edges/s", "cores": best_threads, "kind": "port",
           "sample": f"oracle/ref_conv.sage_layer_fwd_bwd, 1 SAGE layer fwd+bwd, N={Ns} E={Es} F={F} fp32 (1/10-scale C4), best of "
                     f"{runs} timed runs over 16/32/64 torch threads; {where}",
           "ran": "1/10-scale sample", "mem_available_gb": round(mem, 1)}
    full = (args.nodes, args.edges) == (1_000_000, 20_000_000) and not args.cpu_small
    if full and mem >= 128.0:
        try:
            torch.set_num_threads(best_threads)
            big = data(args.nodes, args.edges, ei_full)
            times = []
            for it in range(2):                                 # one warm-up, one timed
                t0 = time.time()
                R.sage_layer_fwd_bwd(*big)
                times.append(time.time() - t0)
            del big
            res.update(value=args.edges / times[1], ran="full C4 configuration", small_sample_edges_per_s=Es / best,
                       sample=f"oracle/ref_conv.sage_layer_fwd_bwd, 1 SAGE layer fwd+bwd on the metric's own configuration "
                              f"(N={args.nodes} E={args.edges} F={F} fp32, the graph of the GPU line): one timed run after one "
                              f"warm-up ({times[0]:.1f} / {times[1]:.1f} s) at {best_threads} torch threads = the best of 16/32/64 "
                              f"on a 1/10-scale sample; {where}")
        except Exception as e:                                  # e.g. the host ran out of memory after all
            res["full_c4_error"] = f"{type(e).__name__}: {e}"[:200]
    elif full:
        res["ran"] = f"1/10-scale sample (MemAvailable {mem:.0f} GB < 128 GB needed for the full C4 message tensors)"
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


FETCH_SCALE = 1.992            # gfx950: FETCH_SIZE under-reports the gathers' wide coalesced reads by this factor (the guide's HBM section;
                               # calibrated in round 1, tools/pmc_calibrate.py, and confirmed by the uniform control: PMC = 1.00 x algorithmic)


def under_profiler() -> bool:
    """is this process itself running under rocprofv3 / rocprof (its tool library preloaded, or its environment set)?  The live PMC
    child passes are then skipped: nested counter collection conflicts with the parent's, and the wall time would be wasted"""
    env = os.environ
    if any("rocprof" in env.get(k, "").lower() for k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES")):
        return True
    return any(k.startswith(("ROCPROF_", "ROCPROFILER_", "ROCP_")) for k in env)


def live_pmc(args, timeout_s: float = 300.0, control: bool = False):
    """HBM-side bytes per launch of the headline aggregation kernel measured IN THIS RUN: two child processes, each this very
    file under `rocprofv3 --pmc <counter>` (FETCH_SIZE and WRITE_SIZE in passes of their own, nothing else traced, as the guide
    prescribes), on a short run of the same workload; started BEFORE this process touches the GPU.  Returns
    {"bytes_per_launch", "fetch_KB_raw", "write_KB", "launches", "source"} or {"error": ...} (the offline constants of
    profiles/pmc_traffic.json then stand in, and the line says so).  ``control``: the same two passes over `--control-only` (the
    uniform-source control graph of roofline.control_uniform)."""
    import csv
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="npi_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    child = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--setup-steps", "2", "--no-cpu-baseline",
             "--no-configs", "--no-control", "--virtual-world", "0", "--no-parity", "--no-live-pmc", "--detail-file", "", "--nodes", str(args.nodes),
             "--edges", str(args.edges), "--hidden", str(args.hidden), "--graph-seed", str(args.graph_seed)]
    if args.plain_csr:
        child.append("--plain-csr")
    if control:
        child = [sys.executable, os.path.abspath(__file__), "--control-only", "--nodes", str(args.nodes), "--edges", str(args.edges),
                 "--hidden", str(args.hidden)]
    env = dict(os.environ, TMPDIR="/tmp")
    means, t0 = {}, time.time()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cp = subprocess.run([prof, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--"] + child, cwd="/tmp",
                                env=env, capture_output=True, text=True, timeout=max(30.0, timeout_s - (time.time() - t0)))
            rows = []
            for root_, _, files in os.walk(out):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        rows += list(csv.DictReader(open(os.path.join(root_, f))))
            vals = [float(r["Counter_Value"]) for r in rows
                    if r.get("Counter_Name", counter) == counter
                    # (forward: the unweighted mean, WMODE 0; backward, aggregate-first: per-entry weights, WMODE 1)
                    and any(k in r.get("Kernel_Name", "") for k in ("segsum_kernel<float, 4, 1, 0", "segsum_kernel<float, 4, 1, 1"))]
            if not vals:
                return {"error": f"no {counter} rows for the aggregation kernel (rc {cp.returncode}): {(cp.stderr or cp.stdout)[-200:]}"}
            means[counter] = (sum(vals) / len(vals), len(vals))
    except Exception as e:                                      # noqa: BLE001 -- a side measurement: the offline constants stand in
        return {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch_kb, write_kb = means["FETCH_SIZE"][0], means["WRITE_SIZE"][0]
    return {"bytes_per_launch": (fetch_kb * FETCH_SCALE + write_kb) * 1024.0, "fetch_KB_raw": fetch_kb, "write_KB": write_kb,
            "launches": means["FETCH_SIZE"][1], "wall_s": round(time.time() - t0, 1),
            "source": f"LIVE: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of this run (separate passes, avg over "
                      f"{means['FETCH_SIZE'][1]} {'control' if control else 'fwd+bwd aggregation'} launches; FETCH_SIZE x {FETCH_SCALE} "
                      "gfx950 calibration + WRITE_SIZE)"}


# ---------------------------------------------------------------------------------------------------------
# roofline.control_uniform: the same kernel with no cache-resident hub table
# ---------------------------------------------------------------------------------------------------------
def control_uniform(dev, N, E, F, launches=10, live=None):
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
    is_live = bool(live and live.get("bytes_per_launch"))
    if is_live:
        traffic = live["bytes_per_launch"]
    res = {"workload": f"N={N} E={E} uniform random sources and targets (every row of the {N * F * 4 / 1e9:.2f} GB table "
                       f"equally likely: no cache-resident hub side), F={F} fp32, aggregation launches only",
           "avg_launch_ms": ms, "launches_timed": len(ev), "algorithmic_bytes_per_launch": alg,
           "achieved": ach, "frac_algorithmic": ach / HBM_PEAK_GBS,
           "traffic": traffic,
           "frac_traffic": (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
           "traffic_source": live["source"] if is_live else (t.get("control_from") if traffic else
           (f"STALE: {t.get('control_from')} was measured on another {t.get('stale')}" if t.get("stale") else None)),
           "live_pmc": live}
    del graph, x
    torch.cuda.empty_cache()
    return res


# ---------------------------------------------------------------------------------------------------------
# is what was timed RIGHT?
# ---------------------------------------------------------------------------------------------------------
def timed_config_parity(ei_dev, N, x, go, conv, out, chunk=2_000_000):
    """N = 1: the tensors of the LAST timed step (the conv on the very CSR the timed region walked: column-sorted unless
    --plain-csr) against the SAGEConv formulas of SURVEY.md Appendix B evaluated in fp64 by plain torch ops on the GPU --
    index_add_ over the ORIGINAL edge list in chunks, fp64 matmuls; nothing of this package.  Every row of out and dX, all of dW
    and db.  Errors are max |diff| / max |reference|."""
    W64, b64 = conv.weight.detach().double(), conv.bias.detach().double()
    src, dst = ei_dev[0], ei_dev[1]
    keep = src != dst                                           # add_remaining_self_loops drops existing loops first
    cnt = (torch.bincount(dst[keep], minlength=N) + 1).double().view(-1, 1)
    x64 = x.detach().double()

    def spread(table, frm, to):                                # acc[to[e]] += table[frm[e]] over the kept edges, + the self loop
        acc = table.clone()
        for lo in range(0, src.numel(), chunk):
            k = keep[lo: lo + chunk]
            acc.index_add_(0, to[lo: lo + chunk][k], table[frm[lo: lo + chunk][k]])
        return acc
    agg = spread(x64, src, dst) / cnt
    go64 = go.double()
    want_out = agg @ W64 + b64
    want_dw = agg.t() @ go64
    del agg
    want_db = go64.sum(0)
    dagg = (go64 @ W64.t()) / cnt
    del go64
    want_dx = spread(dagg, dst, src)
    del dagg

    def rel(a, b):
        return float((a.detach().double() - b).abs().max() / b.abs().max().clamp(min=1e-300))
    err = {"out": rel(out, want_out), "dX": rel(x.grad, want_dx), "dW": rel(conv.weight.grad, want_dw),
           "db": rel(conv.bias.grad, want_db)}
    return {"parity_max_err": max(err.values()), "by_tensor": err,
            "against": f"every row of out / dX and dW, db of the last timed step (the conv on the CSR the timed region used) against "
                       "mean-aggregate @ W + b and its transpose evaluated in fp64 torch ops over the original edge list; max |diff| / "
                       "max |reference|"}


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


def dist_is_up() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


# ---------------------------------------------------------------------------------------------------------
# the final line
# ---------------------------------------------------------------------------------------------------------
def _num(v, nd=6):
    """a JSON-safe number with a bounded mantissa (strict JSON: no NaN / Infinity)"""
    if v is None or isinstance(v, (bool, int, str)):
        return v
    v = float(v)
    if v != v or v in (float("inf"), float("-inf")):
        return None
    return float(f"{v:.{nd}g}")


def _short(s, n=160):
    return s if (s is None or len(s) <= n) else s[: n - 3] + "..."


def configs_summary(cfg: dict) -> dict:
    """one number per side config for the compact line (ms per step unless the key says otherwise)"""
    out = {}

    def ms(name, key="ms_per_step"):
        v = cfg.get(name)
        if not isinstance(v, dict):
            return None
        if "error" in v:
            return "error"
        if "skipped" in v:
            return "skipped"
        return _num(v.get(key), 4)
    for name in ("C1", "C3"):
        if name in cfg:
            out[name] = ms(name, "ms_per_step_graph") or ms(name)
            out[name + "_eager"] = ms(name)
    if "C2" in cfg:
        out["C2"] = ms("C2", "ms_per_step_graph") or ms("C2")
        out["C2_eager"] = ms("C2")
        out["C2_f32"] = ms("C2", "ms_per_step_graph_f32")
    if "R_net1_step" in cfg:
        out["R_net1_step"] = ms("R_net1_step")
        out["R_net1_step_eager"] = ms("R_net1_step", "ms_per_step_eager")
    for name in ("gcn_c4", "gat_c4", "C4_bf16_storage", "C5_1gpu"):
        if name in cfg:
            out[name] = ms(name)
    for k, v in cfg.items():
        if k.startswith("C4_w") and isinstance(v, dict):
            hs = v.get("hubs_sage")
            W = k[4:].split("_")[0]
            if isinstance(hs, dict) and "per_rank_ms" in hs:
                out[f"w{W}_hubs_sage_rank_ms_max"] = _num(max(hs["per_rank_ms"]), 4)
                out[f"w{W}_hubs_sage_ceiling"] = _num(hs.get("compute_ceiling"), 4)
            for w_, c in (v.get("hubs_sage_by_world") or {}).items():
                if w_ != W and isinstance(c, dict):
                    out[f"w{w_}_hubs_sage_ceiling"] = _num(c.get("compute_ceiling"), 4)
            hg = v.get("hubs_gat")
            if isinstance(hg, dict) and "per_rank_ms" in hg:
                out[f"w{W}_hubs_gat_rank_ms_max"] = _num(max(hg["per_rank_ms"]), 4)
            if "error" in v:
                out[f"w{W}_virtual"] = "error"
    return out


def compact_line(res: dict) -> str:
    """The LAST stdout line: strict JSON, target <= 4 KB, never above LINE_CAP.  `res` is the full record (written to
    bench_detail.json); this keeps the contract's keys, the roofline and cpu_baseline blocks and one number per side config."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    line = {k: _num(res.get(k), 7) for k in keep}
    line["ms_per_step_repeats"] = [_num(v, 5) for v in res.get("ms_per_step_repeats") or []]
    cfg = dict(res.get("config") or {})
    at = cfg.get("autotune")
    if isinstance(at, dict):
        cfg["autotune"] = {"chosen": at.get("chosen"), "error": _short(at.get("error"), 120),
                           "ms_per_step": {k: _num(v, 4) for k, v in (at.get("ms_per_step") or {}).items()}}
    cfg["fallback"] = _short(cfg.get("fallback"), 200)
    line["config"] = cfg
    if "parity_max_err" in res:
        line["parity_max_err"] = _num(res["parity_max_err"], 3)
        p = res.get("parity") or {}
        line["parity"] = {"by_tensor": {k: _num(v, 3) for k, v in (p.get("by_tensor") or {}).items()},
                          "against": _short(p.get("against"), 260), "error": _short(p.get("error"), 200)}
    r = res.get("roofline") or {}
    roof = {k: _num(r.get(k), 6) for k in ("bound", "kernel", "avg_launch_ms", "launches_timed", "algorithmic_bytes_per_launch",
                                           "achieved", "achieved_traffic", "peak", "unit", "frac", "frac_algorithmic", "frac_traffic",
                                           "traffic", "fwd_launch_ms", "bwd_launch_ms")}
    roof["frac_basis"] = ("achieved_traffic/peak; traffic = L2-miss bytes by PMC (hits in the 256 MiB memory-side cache included): an upper bound on DRAM "
                          "utilisation, control_uniform is the cache-free figure" if r.get("frac_traffic") is not None else
                          "achieved/peak (algorithmic bytes; no PMC traffic on file)")
    roof["kernel"] = _short(roof.get("kernel"), 120)
    roof["traffic_source"] = _short(r.get("traffic_source"), 230)
    cu = r.get("control_uniform")
    if isinstance(cu, dict):
        roof["control_uniform"] = ({"error": _short(cu["error"], 120)} if "error" in cu else
                                   {"avg_launch_ms": _num(cu.get("avg_launch_ms"), 5), "frac_algorithmic": _num(cu.get("frac_algorithmic"), 4),
                                    "frac_traffic": _num(cu.get("frac_traffic"), 4),
                                    "frac": _num(cu.get("frac_traffic") or cu.get("frac_algorithmic"), 4),
                                    "what": "same kernel, sources and targets uniform over all rows (nothing cache-resident)"})
    line["roofline"] = roof
    pj = res.get("projection")
    if isinstance(pj, dict):
        line["projection"] = {"bound": "mfma", "achieved": _num(pj.get("achieved"), 5), "peak": pj.get("peak"), "unit": "TFLOP/s",
                              "frac": _num(pj.get("frac"), 4), "achieved_f32_equivalent": _num(pj.get("achieved_f32_equivalent"), 5),
                              "per_gemm_ms": {k: _num(v, 4) for k, v in (pj.get("per_gemm_ms") or {}).items()},
                              "what": _short(pj.get("frac_note"), 100)}
    ex = res.get("exchange")
    if isinstance(ex, dict):
        line["exchange"] = {"exposed_ms_per_step": _num(ex.get("exposed_ms_per_step"), 4),
                            "by_collective_ms_per_step": {k: _num(v, 4) for k, v in (ex.get("by_collective_ms_per_step") or {}).items()}}
    ao = res.get("aggregation_only")
    if isinstance(ao, dict):
        line["aggregation_only"] = {"edges_per_s": _num(ao.get("edges_per_s"), 5), "ms_per_step": _num(ao.get("ms_per_step"), 5)}
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {"value": _num(cb.get("value"), 5), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": _short(cb.get("sample"), 420), "ran": _short(cb.get("ran"), 120)}
    if isinstance(res.get("configs"), dict):
        line["configs"] = configs_summary(res["configs"])
    line["detail"] = res.get("detail")
    line["wall_s"] = res.get("wall_s")
    s = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(s) > LINE_CAP:                                       # never expected; drop the optional blocks rather than lose the line
        for k in ("configs", "exchange", "projection", "aggregation_only", "parity"):
            line.pop(k, None)
            s = json.dumps(line, allow_nan=False, separators=(",", ":"))
            if len(s) <= LINE_CAP:
                break
    return s


def emit(res: dict, detail_file: str = DETAIL_FILE) -> None:
    """detail file first (best effort), then the compact line as the last thing on stdout"""
    if detail_file:
        try:
            with open(detail_file, "w") as f:
                json.dump(res, f, indent=1, default=str)
            res["detail"] = detail_file
        except OSError as e:
            res["detail"] = f"not written: {e}"[:120]
    else:
        res["detail"] = None
    sys.stdout.flush()
    print(compact_line(res), flush=True)


# ---------------------------------------------------------------------------------------------------------
def main():
    t_start = time.time()
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
    sharded = world > 1 or args.force_sharded
    is_worker = os.environ.get("NPI_BENCH_WORKER") == "1"
    if sharded and not is_worker and not args.control_only:
        raise SystemExit(supervise(sys.argv[1:], rank, world))
    if is_worker and os.environ.get("NPI_BENCH_FAKE_WORKER"):
        raise SystemExit(fake_worker(rank, world))
    attempt = int(os.environ.get("NPI_BENCH_ATTEMPT", "0")) if is_worker else 0
    pmc_live = pmc_live_control = None
    if (not sharded and not args.no_live_pmc and not args.control_only and args.conv == "sage" and args.storage == "f32"
            and not args.capture and not under_profiler()):
        pmc_live = live_pmc(args)                               # child processes; this one has made no GPU call yet
        if not args.no_control and pmc_live.get("bytes_per_launch"):
            pmc_live_control = live_pmc(args, control=True)
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
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a FRESH file store per attempt: nothing of a failed attempt (keys in the launcher's store, half-open sockets) is met again
        store = f"file://{os.path.join(rendezvous_dir(), f'store_{attempt}')}"
        dist.init_process_group("nccl", init_method=store, rank=rank, world_size=world, device_id=dev)
        if world == 1:
            from npi_gnn_amd import dist as _ND
            _ND.ALWAYS_COMMUNICATE = True

    import npi_gnn_amd as npi
    from npi_gnn_amd import functional as NF
    from npi_gnn_amd.synth import bipartite_edge_index

    N, E, F = args.nodes, args.edges, args.hidden
    if args.control_only:
        print(json.dumps({"control_uniform": control_uniform(dev, N, E, F)}), flush=True)
        return
    fallback = None
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
    autotune = None
    last = {}                             # the output of the most recent step (N = 1: the parity check reads it)

    if not sharded:
        t0 = time.time()
        ei_dev = ei.to(dev)
        # every row's entries in column order (CSRGraph(sort_columns=): a second key for the build's sort, once per graph;
        # nothing for this SAGEConv line, 0.8-1.2 % for the GATConv configs -- EXPERIMENTS A22); --plain-csr: list order
        graph = npi.CSRGraph(ei_dev, N, sort_columns=not args.plain_csr)
        _ = graph.by_src
        torch.cuda.synchronize()
        t_build = time.time() - t0
        conv = {"sage": npi.SAGEConv, "gcn": npi.GCNConv, "gat": npi.GATConv}[args.conv](F, F).to(dev)
        with torch.no_grad():
            conv.weight.copy_(W)
            conv.bias.copy_(bias)
        st = torch.bfloat16 if args.storage == "bf16" else torch.float32
        conv = conv.to(st)
        x = x_full.to(dev).to(st).requires_grad_(True)
        go = go_full.to(dev).to(st)
        norm = NF.GCNNorm(graph) if args.conv == "gcn" else None
        c4.update(graph=graph, x=x, go=go, F=F, E=E)
        # GATConv: the row scales of the input features, once (they do not change between steps, like the CSR): the projection x W then
        # runs on two fp16 pieces per operand (functional.gat_conv(x_scales=))
        gat_scales = NF.row_scales(x.detach()) if (args.conv == "gat" and x.dtype == torch.float32) else None

        def step():
            for p in conv.parameters():
                p.grad = None
            x.grad = None
            if norm is not None:
                out = NF.gcn_conv(x, None, conv.weight, conv.bias, norm=norm)
            elif gat_scales is not None:
                out = conv(x, graph, x_scales=gat_scales)
            else:
                out = conv(x, graph)
            out.backward(go)
            last["out"] = out.detach()                          # (detached: the step's autograd graph must not outlive the step)
        seg_launch_bytes = [algorithmic_bytes(E, N, F, s=2 if args.storage == "bf16" else 4)]
    else:
        from npi_gnn_amd import dist as ND
        from npi_gnn_amd.schedule import CONSERVATIVE, DEFAULT
        from npi_gnn_amd.synth import protein_mask
        t0 = time.time()
        conservative = args.conservative or attempt > 0
        if attempt > 0:
            fallback = "second attempt on the conservative schedule after: " + first_attempt_failure()
        # every rank ships only ITS slice of the edge list to its GPU; the partitioner routes the edges (dist.route_edges)
        ei_mine = ei[:, rank * E // world: (rank + 1) * E // world] if world > 1 else ei
        att_full = torch.randn(1, 1, 2 * F, generator=g) * 0.1
        # a second communicator for the small exchanges (per-row scalars, the softmax's MAX, parameter-gradient sums): on the main
        # one they would queue behind the hub-row tables issued before them (ShardedGraph(small_group=))
        small_group = None
        if rccl and not conservative:
            import torch.distributed as dist
            small_group = dist.new_group()

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

        def step():
            layer.zero_grad()
            x.grad = None
            out = layer(x)
            out.backward(go)

        # Set-up phase.  Anything that raises here ends THIS process with a non-zero code (the error in err_<attempt>_<rank>): the
        # supervisors then end every rank's worker and start fresh ones on the conservative schedule.  No rank tries to recover
        # in-process (a HIP fault is sticky) and no collective is needed to agree on the failure.
        try:
            sg, layer, x, go, seg_launch_bytes = build_sharded(CONSERVATIVE if conservative else DEFAULT)
            if os.environ.get("NPI_BENCH_INJECT_FAILURE") == "1" and attempt == 0:
                raise RuntimeError("injected pre-flight failure (NPI_BENCH_INJECT_FAILURE=1)")
            step()                                              # pre-flight: ONE step of the layer as configured
            torch.cuda.synchronize()
            t_build = time.time() - t0
            # N > 1 only: three arrangements whose worth depends on what RCCL's kernels do beside ours -- nothing one GPU can tell
            # (EXPERIMENTS A9, A16): the projection GEMMs on 16 CUs fewer, the hub rows of dAgg projected first, and every exchange
            # on ONE communicator.  Each candidate is built, run 5 + 2 x 8 steps between barriers (MAX over ranks), and the fastest
            # becomes THE schedule of the timed region; every rank takes the same decision from the same all-reduced numbers.
            if (not conservative and args.partition == "hubs" and args.conv in ("sage", "gcn") and not args.no_autotune
                    and (rccl or os.environ.get("NPI_BENCH_AUTOTUNE_SOLO") == "1")):
                cands = {"default": DEFAULT, "gemm_reserve_cus=16": DEFAULT.but(gemm_reserve_cus=16),
                         "early_hub_gather": DEFAULT.but(early_hub_gather=True),
                         "gemm_reserve_cus=16,early_hub_gather": DEFAULT.but(gemm_reserve_cus=16, early_hub_gather=True),
                         "one communicator": DEFAULT}                   # (the default has the small exchanges on a second one)
                tuned, best = {}, ("default", None)
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
                    tuned[name] = min(regions)
                    if best[1] is None or tuned[name] < best[1]:
                        best = (name, tuned[name])
                if best[0] != list(cands)[-1]:                       # the last candidate is the one that is built right now
                    del sg, layer, x, go
                    torch.cuda.empty_cache()
                    sg, layer, x, go, seg_launch_bytes = build_sharded(cands[best[0]], best[0] == "one communicator")
                autotune = {"ms_per_step": tuned, "chosen": best[0]}
            if rccl:
                dist.barrier()                                  # every rank is through its set-up
            torch.cuda.synchronize()
        except Exception as e:                                  # noqa: BLE001 -- fatal for the attempt, by design
            worker_note("err", f"{type(e).__name__}: {e}"[:300])
            if is_worker:
                sys.stderr.write(f"rank {rank}: set-up phase failed: {type(e).__name__}: {e}\n")
                sys.stderr.flush()
                os._exit(3)                                     # no destructors, no collective: the supervisors take over
            raise
        worker_note("ok")

    def barrier():
        if rccl:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # steady state first: the first launches of a process pay lazy initialisation (kernel code upload, allocator growth, clock
    # ramp) that W = 3 warm-up steps do not always cover -- the first timed region measured 0.07-0.1 ms per step above the
    # following ones.  SETUP_STEPS untimed steps belong to the set-up, then the contract's W warm-up steps and K timed ones.
    for _ in range(max(args.setup_steps, 0)):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    captured = False
    eager_step = step
    if args.capture and not sharded:
        # the same kernels on the same two streams, recorded once: the replayed step does not depend on the host's pace
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
    elif args.conv == "sage" and not args.no_parity and args.storage == "f32":
        try:
            if captured:
                eager_step()                                    # the tensors of a replay live in the graph's pool: one eager step
            torch.cuda.synchronize()
            parity = timed_config_parity(ei_dev, N, x, go, conv, last["out"])
        except Exception as e:                                  # noqa: BLE001 -- reported, never hidden
            parity = {"parity_max_err": None, "error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()

    # dominant kernel: segsum (fwd + bwd launches have the same algorithmic bytes when F_in == F_out)
    seg_ms = [s.elapsed_time(e) for s, e in seg_events]
    seg_avg_ms = sum(seg_ms) / max(len(seg_ms), 1)
    alg_bytes = sum(seg_launch_bytes) / len(seg_launch_bytes)          # average over the launches of one direction
    pmc = pmc_traffic()
    traffic = None
    traffic_live = bool(pmc_live and pmc_live.get("bytes_per_launch"))
    if traffic_live:
        traffic = pmc_live["bytes_per_launch"]
    elif not sharded and args.conv == "sage" and (N, E, F) == (1_000_000, 20_000_000, 256) and not pmc.get("stale"):
        traffic = pmc.get("segsum_kernel_bytes_per_launch" if args.storage == "f32" else "bf16_segsum_bytes_per_launch")
    achieved, ach_traffic, frac_alg, frac_traffic = roofline_rates(alg_bytes, traffic, seg_avg_ms if seg_ms else 0.0)

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
    split_on = True                      # the layers' default arithmetic: f32 operands split for the 16-bit matrix cores
    # matrix products ISSUED per f32-equivalent product: six bf16 ones, or three fp16 ones where a projection runs on two fp16
    # pieces per operand (Schedule.f16x2_min_rows: SAGEConv / GCNConv at 256 features -- the forward GEMM with the row scales its
    # aggregation writes, and, round 6, the backward's data GEMM: the layer runs its backward aggregate-first, dX = (A^T dOut) W^T,
    # so the transposed aggregation writes that GEMM's left operand and its scales)
    from npi_gnn_amd.schedule import DEFAULT as _SCH
    f16_fwd = (not sharded and args.conv in ("sage", "gcn") and args.storage == "f32" and F == 256
               and NF._f16x2(_SCH, N, F, F, torch.float32))
    f16_bwd = f16_fwd and _SCH.aggregate_first_backward
    products = {"fwd": 3.0 if f16_fwd else 6.0, "bwd_data": 3.0 if f16_bwd else 6.0}
    issued_tf = (sum(v[0] * products.get(k, 6.0) for k, v in gem.items() if k != "bwd_weight") / (solo_ms * 1e-3) / 1e12) if solo_ms else None
    res = None
    if rank == 0:
        # `achieved` = SURVEY 8(d)'s algorithmic bytes / live duration (every gathered row counted as a read, no cache credit: it can
        # exceed the peak when caches serve gathers); `achieved_traffic` = the bytes the PMC counters saw cross the L2 <-> fabric
        # boundary / the same duration, and `frac` = achieved_traffic / peak.  FETCH_SIZE counts the L2's fabric-side requests, so
        # rows served by the 256 MiB Infinity Cache are still inside it (guide, HBM section): `frac` is an UPPER BOUND on DRAM
        # utilisation; the cache-free figure is control_uniform's.
        per_dir = None
        if not sharded and args.conv in ("sage", "gcn") and len(seg_ms) == 2 * args.steps:
            per_dir = (sum(seg_ms[0::2]) / args.steps, sum(seg_ms[1::2]) / args.steps)      # launched in the order forward, backward
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "achieved_traffic": ach_traffic,
                "frac_algorithmic": frac_alg,      # SURVEY 8(d) bytes (every gathered row counted, no cache credit) / time / peak
                "frac_traffic": frac_traffic,      # bytes that crossed the L2 <-> fabric boundary (PMC) / time / peak
                "frac": frac_traffic if frac_traffic is not None else frac_alg,
                "frac_basis": ("achieved_traffic / peak: L2-miss bytes (PMC FETCH_SIZE x calibration + WRITE_SIZE) over the live launch "
                               "duration.  FETCH_SIZE counts the L2's fabric-side requests, Infinity-Cache hits included, so this is an "
                               "UPPER BOUND on DRAM utilisation; control_uniform is the cache-free HBM figure.  frac_algorithmic "
                               "(achieved / peak) exceeds it, and can exceed 1, because gathers of the 100k protein rows hit the XCD L2s")
                if frac_traffic is not None else
                              "achieved / peak: algorithmic bytes over the live launch duration (no current PMC pass on file for this "
                              "configuration; every gathered row counts as an HBM read, so this can exceed 1 when caches serve gathers)",
                "fwd_launch_ms": per_dir[0] if per_dir else None,       # the forward launch has the chip to itself,
                "bwd_launch_ms": per_dir[1] if per_dir else None,       # the backward one shares it with the dW GEMM
                "traffic": traffic if traffic else ("stale" if pmc.get("stale") else None),
                "traffic_source": (pmc_live["source"] + "; durations are live too") if traffic_live else
                (f"OFFLINE rocprofv3 --pmc passes (FETCH_SIZE x {pmc.get('fetch_scale')} gfx950 calibration + WRITE_SIZE, "
                 f"separate runs), {pmc.get('from')}, on this very kernel source (sha checked); durations are live"
                 + (f"; the live passes failed: {pmc_live.get('error')}" if pmc_live else ""))
                if traffic else (f"STALE: {pmc.get('from')} was measured on another {pmc.get('stale')}; frac falls back to "
                                 "frac_algorithmic" if pmc.get("stale") else None),
                "kernel": "segsum_kernel (one launch per aggregation, cut rows finished inside it), avg of fwd and bwd launches",
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": seg_avg_ms,
                "launches_timed": len(seg_ms), "live_pmc": pmc_live,
                "offline_pmc_bytes_per_launch": None if pmc.get("stale") else pmc.get("segsum_kernel_bytes_per_launch")}
        res = {
            "metric": "edges/sec per GNN layer (fwd+bwd)", "value": value, "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "ms_per_step_repeats": repeats,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.storage == "f32" else "bf16 storage / f32 accumulate (NOT the metric's precision)",
            "data": "synthetic",
            "config": {"workload": f"C4 synthetic ncRNA-protein bipartite graph, N={N} nodes, E={E} directed edges "
                                   f"(both directions, Zipf-skewed protein side), 1 {args.conv.upper()}Conv layer "
                                   f"{F}->{F} fp32, fwd+bwd incl. dX/dW/db, graph+features resident in HBM",
                       "parallelism": parallelism(args, world), "hip_graph_replay": captured, "setup_steps": args.setup_steps,
                       "csr_build_s": round(t_build, 4), "csr_sorted_columns": (not args.plain_csr) if not sharded else False,
                       "fallback": fallback,
                       "autotune": autotune if sharded else None,
                       "communicators": (2 if getattr(sg, "small_group", None) is not getattr(sg, "group", None) else 1) if sharded else None},
            "roofline": roof,
            "exchange": exchange,
            "aggregation_only": {"edges_per_s": (E * args.steps / (seg_total_ms * 1e-3)) if seg_ms and world == 1 else None,
                                 "ms_per_step": seg_total_ms / args.steps if seg_ms else None,
                                 "note": "gather + segmented reduction, forward + transposed backward launches of one layer"},
            "projection": {
                "bound": "mfma",
                "kernels": ("fwd + bwd_data: gemm_split_ws_kernel<.., true> (f32 operands as two scaled fp16 pieces, three "
                            "v_mfma_f32_32x32x16_f16 per f32 product, f32 accumulate); the backward runs aggregate-first, so bwd_data is "
                            "dX = T W^T behind the transposed aggregation that writes T and its row scales" if f16_bwd else
                            "fwd: gemm_split_ws_kernel<.., true> (f32 operands as two scaled fp16 pieces, three v_mfma_f32_32x32x16_f16 per "
                            "f32 product); bwd_data: gemm_split_ws_kernel (three bf16 pieces, six v_mfma_f32_32x32x16_bf16); f32 accumulate"
                            if f16_fwd else
                            "fwd + bwd_data: gemm_split_ws_kernel (f32 operands split 3-way into bf16, six "
                            "v_mfma_f32_32x32x16_bf16 per f32 product, f32 accumulate)") if split_on else
                           "fwd + bwd_data: exact-f32 v_mfma_f32_32x32x2_f32 kernels",
                "achieved_f32_equivalent": solo_tf, "unit": "TFLOP/s",
                # the pipe the kernel runs on: six bf16 MFMA flops are issued per f32-equivalent flop
                "achieved": issued_tf if (solo_tf and split_on) else solo_tf,
                "peak": MFMA_BF16_PEAK_TF if split_on else MFMA_F32_PEAK_TF,
                "frac": ((issued_tf / MFMA_BF16_PEAK_TF) if split_on else (solo_tf / MFMA_F32_PEAK_TF)) if solo_tf else None,
                "issued_products_per_f32_product": products if split_on else None,
                "frac_note": ("issued fp16 MFMA flops (3 x the f32-equivalent, both GEMMs) / dense fp16 = bf16 MFMA peak" if f16_bwd else
                              "issued 16-bit MFMA flops (fwd 3 x, bwd_data 6 x the f32-equivalent) / dense bf16 = fp16 MFMA peak" if f16_fwd else
                              "issued bf16-MFMA flops (6 x f32-equivalent) / dense bf16 MFMA peak") if split_on else
                             "f32 flops / f32 MFMA peak",
                "per_gemm_ms": {k: v[1] / v[2] for k, v in gem.items()},
                "per_gemm_tflops_f32_equivalent": {k: v[0] / (v[1] * 1e-3) / 1e12 for k, v in gem.items() if v[1] > 0},
                "note": "bwd_weight (gemm_dw_split_kernel: both operands split on the fly) is timed while it shares the CUs "
                        "with the backward aggregation on a second stream; alone it takes 0.93 ms"},
        }
        if parity is not None:
            res["parity_max_err"] = parity["parity_max_err"]
            res["parity"] = parity
    if rccl:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    # ---- after the timed region (one GPU only): control, per-config summary, CPU baseline -----------------
    if world == 1 and not sharded:
        del x_full, go_full
        if not args.no_control and args.conv == "sage" and args.storage == "f32":
            try:
                res["roofline"]["control_uniform"] = control_uniform(dev, N, E, F, live=pmc_live_control)
            except Exception as e:
                res["roofline"]["control_uniform"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not args.no_configs and args.conv == "sage" and args.storage == "f32":
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_extras as X
            quick = not args.extras
            deadline = (time.time() + args.configs_budget) if quick else None
            vw = None
            if args.virtual_world > 1:
                # what one GPU can measure of the W-GPU run: every rank's step timed alone, collectives = local copies.
                # default run: the vertex cut of SAGEConv at W ranks; --extras: every partition, GATConv, 2 / 4 ranks, emulated wire
                try:
                    Wv = args.virtual_world
                    if quick:
                        vw = X.virtual_world(dev, args, ei_dev, c4, ms_per_step, Wv, only="hubs_sage")
                        curve = {}
                        for w_ in (2, 4):                       # the smaller worlds of the metric's 1 / 2 / 4 / 8 curve (ceilings)
                            if w_ < Wv and time.time() + 6 < deadline - 20:
                                v_ = X.virtual_world(dev, args, ei_dev, c4, ms_per_step, w_, only="hubs_sage")["hubs_sage"]
                                curve[str(w_)] = {k: v_[k] for k in ("per_rank_ms", "balance", "compute_ceiling") if k in v_}
                        vw["hubs_sage_by_world"] = curve
                    else:
                        vw = X.virtual_world(dev, args, ei_dev, c4, ms_per_step, Wv)
                        curve = {}
                        for w_ in (2, 4):
                            if w_ < Wv:
                                v_ = X.virtual_world(dev, args, ei_dev, c4, ms_per_step, w_, only="hubs_sage")["hubs_sage"]
                                curve[str(w_)] = {k: v_[k] for k in ("per_rank_ms", "balance", "compute_ceiling",
                                                                     "wire_bytes_per_rank_per_step") if k in v_}
                        vw["hubs_sage_by_world"] = curve
                except Exception as e:
                    vw = {"error": f"{type(e).__name__}: {e}"[:300]}
                torch.cuda.empty_cache()
            del ei_dev, graph, x, go, step, eager_step, conv
            last.clear()
            res["configs"] = X.run_configs(dev, args, c4, quick=quick, deadline=deadline)
            if vw is not None:
                res["configs"][f"C4_w{args.virtual_world}_virtual"] = vw
    if not args.no_cpu_baseline:
        # rank 0, after the process group is gone (the other ranks have left): the N > 1 line carries the CPU path timed in the
        # same run as well -- the same sample as the N = 1 line, so the two are comparable
        res["cpu_baseline"] = cpu_baseline(args, ei)
    res["wall_s"] = round(time.time() - t_start, 1)
    emit(res, args.detail_file)


if __name__ == "__main__":
    main()
