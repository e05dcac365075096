#!/usr/bin/env python3
"""The lab harness behind `bench.py --extras` and the one-number-per-config summary of the default run: the other
BASELINE.json configs on this GPU (C1-C3 with their parity error against committed oracle outputs, GCN / GAT / bf16 storage at
the C4 shape, C5 on one GPU), the virtual worlds (every rank's step of a W-rank run timed alone on this GPU, collectives =
local copies) and the emulated-wire estimates.  Side measurements: nothing here is the contract's timed region (bench.py)."""
from __future__ import annotations

import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from bench import agg_roofline, algorithmic_bytes, gat_bytes, pmc_traffic  # noqa: E402


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
    # GATConv: the row scales of the INPUT FEATURES, computed once (the feature matrix of a full-batch run does not change between
    # steps, like the CSR) -- the first layer's projection then runs on two fp16 pieces per operand; every layer hands the scales of
    # its output rows (written by its aggregation launch) to the next one
    x_scales = NF.row_scales(xin.detach()) if (kind == "gat" and dtype == torch.float32 and xin.size(1) % 4 == 0) else None

    def step():
        for W, b in params:
            W.grad = b.grad = None
        xin.grad = None
        h, hs = xin, x_scales
        for k, (W, b) in enumerate(params):
            if kind == "sage":
                h = npi.sage_conv(h, graph, W, b)
            elif kind == "gcn":
                h = NF.gcn_conv(h, None, W, b, norm=norm)
            else:
                h, hs = npi.gat_conv(h, graph, W, atts[k], b, heads=1, relu=True, x_scales=hs, return_scales=True)   # F.relu(conv(h)), fused
                continue
            h = torch.relu(h)
        if go is None:
            h.float().pow(2).mean().backward()
        else:
            h.backward(go)
        return h
    return step


def _graphed(kind, weights, x, graph, dtype=torch.float32):
    """npi.GraphedStack over modules carrying ``weights`` = [(W, b), ...]: the mean-square loss of ``_stack_step`` drives the
    backward; ``.replay`` = the captured step, ``.eager`` = the same step launched kernel by kernel"""
    import npi_gnn_amd as npi
    dev = graph.device
    convs = []
    for W, b in weights:
        conv = (npi.SAGEConv if kind == "sage" else npi.GCNConv)(W.size(0), W.size(1)).to(dev)
        with torch.no_grad():
            conv.weight.copy_(W)
            conv.bias.copy_(b)
        convs.append(conv.to(dtype))
    return npi.GraphedStack(convs, graph, x.to(dev).to(dtype), loss=lambda h: h.float().pow(2).mean())


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
        cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "virtual_rank_probe.py")]
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



def run_configs(dev, args, c4, quick=False, deadline=None):
    """ms per full-batch step (fwd+bwd over the layer stack) and, for C1-C3, the max error against the oracle outputs
    committed under tests/golden/ (made by tests/golden/make_golden.py / make_rpi7317.py from the CPU oracle).
    ``quick`` (the default bench run): one number per config inside a wall-clock budget -- fewer regions, no per-head sweep, the
    C5 graph drawn on the device (same distribution, another draw), no virtual C5 world; an entry whose estimated cost no longer
    fits before ``deadline`` (time.time()) is reported as skipped."""
    import npi_gnn_amd as npi
    from npi_gnn_amd import functional as NF
    out = {}
    G = os.path.join(ROOT, "tests", "golden")

    def guarded(name, fn, cost_s=0.0):
        if deadline is not None and time.time() + cost_s > deadline:
            out[name] = {"skipped": "time budget of the default run (bench.py --extras runs everything)"}
            return
        t0 = time.time()
        try:
            out[name] = fn()
        except Exception as e:                                  # the headline line must survive a failing extra
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if isinstance(out[name], dict):
            out[name]["wall_s"] = round(time.time() - t0, 2)
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
        st = _graphed("gcn", fx["gcn64"], x, graph)
        return {"workload": f"{shape}, 2 x GCNConv 178->64->64 fp32, full batch",
                "ms_per_step": _timeit(st.eager, 30, 5), "ms_per_step_graph": _timeit(st.replay, 200, 10),
                "note": "ms_per_step_graph: npi.GraphedStack (the step replayed from a HIP graph); ms_per_step: the same step eager",
                "parity_max_abs_err": float((h[rows] - fx["gcn64_out"]).abs().max()), "parity": "oracle (unpinned: GCNConv)"}

    def c2():
        h = _stack_forward("sage", fx["sage_weights"], x, graph, dtype=torch.bfloat16)
        ref = fx["sage3_out"]
        gb_, gf_ = _graphed("sage", fx["sage_weights"], x, graph, torch.bfloat16), _graphed("sage", fx["sage_weights"], x, graph)
        sb, sf = gb_.eager, gf_.eager
        # eager: host-bound at this size (see _graph_replay_ms) -- the two storage types are timed alternately, best region each
        eb = ef = None
        for _ in range(1 if quick else 3):
            tb, tf = _timeit(sb, 50, 5, rounds=1), _timeit(sf, 50, 5, rounds=1)
            eb, ef = (tb if eb is None else min(eb, tb)), (tf if ef is None else min(ef, tf))
        return {"workload": f"{shape}, 3 x SAGEConv 178->128->128->128, bf16 storage / f32 accumulate, full batch",
                "ms_per_step": eb, "ms_per_step_f32": ef,
                "ms_per_step_graph": _timeit(gb_.replay, 200, 10), "ms_per_step_graph_f32": _timeit(gf_.replay, 200, 10),
                "note": "ms_per_step*: eager (host-bound: ~40 launches per step); ms_per_step_graph*: npi.GraphedStack, the same "
                        "step replayed from a HIP graph = the GPU's time",
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
        st = _graphed("gcn", f3["gcn256"], x3, g3)
        return {"workload": f"RPI7317 graph N={x3.size(0)} E={ei3.size(1)} (7,317 positives + 7,317 seeded negatives), "
                            "3 x GCNConv 178->256->256->256 fp32, full batch",
                "ms_per_step": _timeit(st.eager, 30, 5), "ms_per_step_graph": _timeit(st.replay, 200, 10),
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

    guarded("C1", c1, 3)
    guarded("C2", c2, 5)
    guarded("C3", c3, 3)
    guarded("R_net1_step", r_step, 6)
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
            conv(xx, g4, x_scales=xs4).backward(go4)
        xs4 = NF.row_scales(xx.detach())        # of the input features, once: they do not change between steps (like the CSR)
        ms = _timeit(step, 10, 3)
        tags = {}
        NF._PROFILE_TAGS = tags
        for _ in range(5):
            step()
        NF._PROFILE_TAGS = None
        torch.cuda.synchronize()
        gb = gat_bytes(E4, N4, F)
        roof = {}
        for tag, kern in (("gat_fwd_aggregate", "segsum_kernel<f32, 4, 1, W_GAT_DST_FUSED>: weighted aggregation with the softmax statistics inside the launch"),
                          ("gat_bwd_fused", "segsum_kernel<f32, 4, 1, W_GAT_SRC_FUSED>: by-source aggregation + SDDMM in one gather pass")):
            tr, src = pmc_of(tag + "_bytes_per_launch", "stale_gat")
            roof[tag] = agg_roofline(tags.get(tag, []), gb[tag], tr, kern, src)
        by_heads = {}
        for Hh in (() if quick else (2, 4, 8)):                # the same layer width as 2 / 4 / 8 heads of 128 / 64 / 32 channels
            cv = npi.GATConv(F, F // Hh, heads=Hh).to(dev)

            def hstep(cv=cv):
                for p in cv.parameters():
                    p.grad = None
                xx.grad = None
                cv(xx, g4).backward(go4)
            by_heads[str(Hh)] = _timeit(hstep, 5, 2)
            del cv
        return {"workload": f"C4 graph, 1 x GATConv {F}->{F} (1 head) fp32 fwd+bwd", "ms_per_step": ms,
                "x_scales": "the row scales of the INPUT FEATURES are computed once, outside the timed steps (a full-batch feature matrix is "
                            "static between steps, like the CSR): the projection x W then runs on two fp16 pieces per operand (gat_conv(x_scales=))",
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
        tr = None if pmc.get("stale") else pmc.get("bf16_segsum_bytes_per_launch")
        return {"workload": f"C4 graph, 1 x SAGEConv {F}->{F} fwd+bwd, bf16 storage / f32 accumulate (NOT the metric's precision)",
                "ms_per_step": ms, "edges_per_s": E4 / ms * 1e3, "max_dev_from_f32_output_rel": dev_rel,
                "roofline": agg_roofline(ev, alg, tr, "segsum_kernel<bf16, 4, 1, W_NONE>, avg of the forward and the backward "
                                         "launch (the latter co-resident with dW); NOT the metric's precision",
                                         pmc.get("bf16_from") if tr else "no current PMC pass for the bf16 kernels: algorithmic "
                                         "bytes (516 B per edge, 1,028 B per node) only")}

    guarded("gcn_c4", gcn_c4, 2)
    guarded("gat_c4", gat_c4, 2)
    guarded("C4_bf16_storage", c4_bf16, 2)

    if not args.skip_c5:
        def c5():
            from npi_gnn_amd.synth import bipartite_edge_index, bipartite_edge_index_device
            c4.clear()                                             # release the C4 graph and features first
            torch.cuda.empty_cache()
            N5, E5, F5 = 4_000_000, 100_000_000, 256
            # quick: the same distribution drawn on the device (2 s; the host generator takes a minute at this size)
            ei5 = bipartite_edge_index_device(N5, E5, dev, seed=2) if quick else bipartite_edge_index(N5, E5, seed=2).to(dev)
            g5 = npi.CSRGraph(ei5, N5, sort_columns=not args.plain_csr)
            _ = g5.by_src
            weights = [((torch.randn(F5, F5, generator=gen) / 16), torch.zeros(F5)) for _ in range(3)]
            att = [torch.randn(1, 1, 2 * F5, generator=gen) * 0.1 for _ in range(3)]
            if quick:
                gd = torch.Generator(device=dev).manual_seed(11)
                x5, go5 = (torch.randn(N5, F5, generator=gd, device=dev) for _ in range(2))
            else:
                x5 = torch.randn(N5, F5, generator=gen).to(dev)
                go5 = torch.randn(N5, F5, generator=gen).to(dev)
            # (with output.pow(2).mean() driving the backward, as rounds 1-3 measured this config)
            ms_loss = None if quick else _timeit(_stack_step("gat", weights, x5, g5, att=att), 3, 1, rounds=2)
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
                tr = None if (pmc.get("stale_gat") or quick) else pmc.get("c5_" + tag + "_bytes_per_launch")
                src = pmc.get("c5_from") if tr else (f"STALE: measured on another {pmc.get('stale_gat')}" if pmc.get("stale_gat")
                                                     else "no PMC pass at this size: algorithmic bytes only")
                roof[tag] = agg_roofline(tags.get(tag, []), gb[tag], tr, "as configs.gat_c4.roofline, at the C5 size (3 launches, "
                                         "one per layer)", src)
            res5 = {"workload": f"C5 synthetic bipartite N={N5} E={E5}, 3 x GATConv 256 (1 head) fp32 fwd+bwd, ONE GPU"
                                + (" (graph of the same distribution drawn on the device)" if quick else ""),
                    "ms_per_step": ms, "edge_layers_per_s": 3 * E5 / ms * 1e3, "ms_per_step_with_mse_loss": ms_loss,
                    "x_scales": "row scales of the input features computed once outside the timed steps (static between steps, like the CSR); "
                                "every layer hands the scales of its output rows -- written by its aggregation launch -- to the next one: "
                                "all three projections x W run on two fp16 pieces per operand (gat_conv(x_scales=, return_scales=True))",
                    "step": "forward + backward of the three layers from a given output gradient, as the headline's step "
                            "(ms_per_step_with_mse_loss: with output.pow(2).mean() driving the backward -- 8 ms of elementwise "
                            "kernels over [4M, 256] -- which is how rounds 1-3 timed this config)",
                    "roofline": roof}
            del st, g5, go5
            x5 = x5.detach()
            vw5 = args.virtual_world > 1 and not quick
            ref = (x5, torch.randn(N5, F5, generator=gen).to(dev)) if vw5 else None     # inputs of the parity check
            torch.cuda.empty_cache()
            if vw5:
                # BASELINE.json configs[4] in its 8-GPU form, rank by rank on this GPU: the same 3-layer GATConv stack on the
                # hub cut, a rank's output rows being the next layer's input rows (collectives = local copies, as C4_w8_virtual),
                # then the same ranks in exact lock step against the single-GPU stack (parity)
                try:
                    res5["w8_virtual"] = virtual_c5(dev, ei5, N5, F5, weights, att, ms, args.virtual_world, ref=ref)
                except Exception as e:
                    res5["w8_virtual"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            del ei5, ref, x5
            torch.cuda.empty_cache()
            if vw5 and args.virtual_world == 8 and "error" not in res5.get("w8_virtual", {"error": 1}):
                # ONE GATConv layer of this size on rank 0 of 8 with the exchanges emulated (the child builds its own graph of the
                # same shape); the single-GPU figure beside it is a third of the 3-layer step
                res5["w8_virtual"]["emulated_wire_one_layer"] = emulated_wire(
                    ["--conv", "gat", "--nodes", str(N5), "--edges", str(E5), "--steps", "10"], ms / 3, timeout=600)
            return res5
        guarded("C5_1gpu", c5, 8 if quick else 0)
    return out



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
    if "hubs_sage" in out and only is None and W == 8 and (N, E, F) == (1_000_000, 20_000_000, 256):
        # rank 0's SAGEConv step replayed from a HIP graph (the default schedule captures since round 5: the partial side runs in
        # line while a capture is on): the GPU's time for the rank's launches, without the host's ~25 dependent launches
        rep = None
        try:
            cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "virtual_rank_probe.py"), "--conv", "sage", "--capture",
                                 "--inline-copies", "--steps", "50"], capture_output=True, text=True, timeout=240)
            m = [l for l in cp.stdout.splitlines() if "events" in l and "capture=True" in l]
            if m:
                rep = float(m[-1].split("events")[1].split("ms/step")[0])
        except Exception as e:                                  # noqa: BLE001 -- a side measurement
            sys.stderr.write(f"graph replay of a SAGEConv rank step failed: {type(e).__name__}: {e}\n")
        out["hubs_sage"]["rank0_ms_graph_replay"] = rep
        out["hubs_sage"]["compute_ceiling_graph_replay"] = (t1_sage_ms / rep) if rep else None
    if "hubs_gat" in out and W == 8 and (N, E, F) == (1_000_000, 20_000_000, 256):
        # the GATConv rank step is ~110 launches of a few us: eager it is bounded by the HOST and moves with the box's CPU
        # (1.7-2.2 ms).  Its GPU time: rank 0's step replayed from a HIP graph, stand-in copies on the compute stream (capture with
        # the copy stream's nested forks takes the HIP runtime down at capture_end, hence a CHILD process; None if it fails)
        rep = None
        try:
            cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "virtual_rank_probe.py"),
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

