"""The W ranks of the sharded layers (``dist.py``) inside ONE process on ONE GPU -- SURVEY.md 8(e): "run G virtual shards
sequentially on one GPU to validate partition + reduction numerics".  Two tools, both swap the collective calls of ``dist``
for in-process stand-ins while they are active:

``LockStep``         exact numerics.  The ranks run one after the other, pass after pass; collective number k of a pass
                     returns its TRUE result (the gather / sum / max over all ranks' inputs) once every rank's input to it is
                     known from an earlier pass, a dummy before.  The inputs of collective k depend only on the results of the
                     collectives before it and the kernels are deterministic, so pass p resolves collective p and the last
                     pass is an exact lock-step execution of the W-rank run.  Used by the ``-m gpu`` parity tests up to the
                     full C5 size and by ``bench.py`` to put a parity error beside every virtual-rank timing.
``StubCollectives``  timing.  Every collective becomes a local copy of the same shape (one GPU stands in for rank r of W); the
                     bytes each call would move are logged.  What a rank's step costs WITHOUT the wire.

Neither is a transport: a real run uses RCCL (``dist.all_gather_rows`` etc.).
"""
from __future__ import annotations

from typing import Callable, List

import torch
import torch.distributed as tdist

from . import dist as ND


class _Done:
    def wait(self):
        return True


class _EventWork:
    """the pending stand-in collective: ``wait()`` makes the CURRENT stream wait for its event (what ``Work.wait()`` of an RCCL
    collective does)"""
    __slots__ = ("ev", "dev")

    def __init__(self, ev, dev):
        self.ev, self.dev = ev, dev

    def wait(self):
        torch.cuda.current_stream(self.dev).wait_event(self.ev)
        return True


# stands for ``ShardedGraph(small_group=)`` in a virtual world: ``StubCollectives`` runs the collectives issued on it on its
# second stream (``copy_stream2``), as a second communicator runs its own in their own order
SMALL_LANE = "small-lane"


class _Patched:
    """swap dist's collective entry points while active; ``_solo`` is forced False so that a rank of a virtual world takes the
    same code path as a rank of a real one"""

    def _install(self, ag, rs, ar):
        self._saved = (ND.all_gather_rows, ND.reduce_scatter_rows, ND._all_reduce, ND._solo)
        ND.all_gather_rows, ND.reduce_scatter_rows, ND._all_reduce = ag, rs, ar
        ND._solo = lambda w: False

    def __exit__(self, *exc):
        ND.all_gather_rows, ND.reduce_scatter_rows, ND._all_reduce, ND._solo = self._saved
        return False


class LockStep(_Patched):
    """``with LockStep(W) as ls: results = ls.run(run_rank)`` -- ``run_rank(r)`` executes rank r's whole step (forward and
    backward through layers built on ITS ``ShardedGraph``) and returns whatever the caller wants to keep of it; ``run`` calls
    it for every rank, pass after pass, and returns the W results of the first pass in which no collective had to be faked.
    ``run_rank`` must be deterministic and must not depend on results of earlier passes."""

    def __init__(self, world: int, max_passes: int = 512):
        self.W, self.max_passes = int(world), int(max_passes)
        self.inputs, self.results = [], []      # per collective index: {rank: tensor} / resolved result or ("pending", fn)
        self.state = {"rank": 0, "k": 0, "valid": True}
        self.passes = 0

    # ---- the collective protocol ---------------------------------------------------------------------------------------------
    def _collective(self, value: torch.Tensor, fn: Callable[[List[torch.Tensor]], torch.Tensor]):
        st = self.state
        k, r = st["k"], st["rank"]
        st["k"] += 1
        while len(self.inputs) <= k:
            self.inputs.append({})
            self.results.append(None)
        res = self.results[k]
        if res is not None and not isinstance(res, tuple):
            return res
        if st["valid"] and res is None:
            self.inputs[k][r] = value.detach().clone()
            if len(self.inputs[k]) == self.W:
                self.results[k] = ("pending", fn)
        st["valid"] = False
        return None

    def _resolve(self):
        for k, res in enumerate(self.results):
            if isinstance(res, tuple) and res[0] == "pending":
                self.results[k] = res[1]([self.inputs[k][r] for r in range(self.W)])
                self.inputs[k] = {}               # the ranks' inputs are not needed again

    def __enter__(self):
        me = self

        def all_gather_rows(block, out, w, group=None, async_op=False):
            res = me._collective(block, lambda v: torch.cat([t.reshape(t.size(0), -1) for t in v]))
            out.copy_(res.view_as(out)) if res is not None else out.zero_()
            return _Done() if async_op else None

        def reduce_scatter_rows(part_sums, out, rank, w, group=None, async_op=False):
            def total(v):
                acc = v[0].clone()
                for t in v[1:]:
                    acc += t
                return acc
            res = me._collective(part_sums, total)
            out.copy_(res.view(w, out.size(0), -1)[rank].view_as(out)) if res is not None else out.zero_()
            return _Done() if async_op else None

        def all_reduce(t, w, group=None, op=None, tag="", async_op=False):
            mx = op == tdist.ReduceOp.MAX
            res = me._collective(t, (lambda v: torch.stack(v).max(0)[0]) if mx else (lambda v: torch.stack(v).sum(0)))
            if res is not None:
                t.copy_(res)
            return _Done() if async_op else None

        self._install(all_gather_rows, reduce_scatter_rows, all_reduce)
        return self

    def run(self, run_rank: Callable[[int], object]) -> list:
        for _ in range(self.max_passes):
            self.passes += 1
            out, complete = [], True
            for r in range(self.W):
                self.state.update(rank=r, k=0, valid=True)
                out.append(run_rank(r))
                # a rank's step uses several HIP streams (partial side, weight gradient); the inputs it left for the collectives
                # were cloned on whichever stream issued them: everything has to have landed before the next rank / the resolution
                # (on the default stream) reads them
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                complete = complete and self.state["valid"]
            if complete:
                return out
            del out
            self._resolve()
        raise RuntimeError(f"LockStep: {self.W} virtual ranks did not converge in {self.max_passes} passes")


class StubCollectives(_Patched):
    """Replace the collectives by local copies of the same shapes and log what every call would move: payload bytes, and bytes
    on the wire per rank (an all-gather / reduce-scatter of S bytes moves S (W-1)/W per rank, an all-reduce 2 S (W-1)/W).
    ``copy_stream``: run the stand-in copies on that HIP stream with event edges where RCCL's own stream would sit (the
    collective is issued when its input is ready on the compute stream, ``wait()`` makes the compute stream wait for it) --
    the dependency graph of the real run; None: plain copies on the compute stream.
    ``wire_gbps``: EMULATE the exchange's duration and footprint as well -- in front of every stand-in copy a kernel that
    computes nothing holds ``held_cus`` CUs (``npi_hold_cus``: 64 KB of LDS each, as a collective's resident workgroups hold
    theirs) for ``latency_us`` + wire bytes per rank / ``wire_gbps``.  What comes out is an estimate under those two stated
    assumptions, not a measurement of xGMI."""

    def __init__(self, W: int, copy_stream: "torch.cuda.Stream | None" = None, wire_gbps: "float | None" = None,
                 held_cus: int = 16, latency_us: float = 20.0, copy_stream2: "torch.cuda.Stream | None" = None):
        self.W, self.log, self.copy_stream = int(W), {}, copy_stream
        self.copy_stream2 = copy_stream2                  # collectives issued on SMALL_LANE (None: the same stream as the rest)
        self.wire_gbps, self.held_cus, self.latency_us = wire_gbps, int(held_cus), float(latency_us)

    def _hold(self, wire_bytes: float, dev) -> None:
        if self.wire_gbps:
            from ._lib import check, load, stream_ptr
            ns = int(self.latency_us * 1e3 + wire_bytes / self.wire_gbps)          # bytes / (GB/s) = ns
            words = getattr(self, "_words", None)
            if words is None or self._next >= words.numel():
                words = self._words = torch.zeros(1 << 16, dtype=torch.int64, device=dev)   # one zeroed start word per exchange
                self._next = 0
            check(load().npi_hold_cus(self.held_cus, ns, words.data_ptr() + 8 * self._next, stream_ptr(dev)), "npi_hold_cus")
            self._next += 1

    def note(self, kind, nbytes, wire):
        e = self.log.setdefault(kind, {"calls": 0, "payload_bytes": 0, "wire_bytes_per_rank": 0})
        e["calls"] += 1
        e["payload_bytes"] += nbytes
        e["wire_bytes_per_rank"] += wire

    def _issue(self, fn, tensors, async_op, wire_bytes: float = 0.0, group=None):
        """run ``fn`` (the stand-in copy) where the collective would run"""
        cs = self.copy_stream2 if (group is SMALL_LANE and self.copy_stream2 is not None) else self.copy_stream
        if cs is not None and torch.cuda.is_current_stream_capturing():
            # inside a HIP-graph capture the stand-in runs in line: a stream of its own would be a fork of a fork (launch ->
            # side -> copy), which takes the HIP runtime down in hipStreamEndCapture (EXPERIMENTS A3; dist.HipBackend does the
            # same with its partial stream)
            cs = None
        if cs is None:
            self._hold(wire_bytes, tensors[0].device)
            fn()
            return _Done() if async_op else None
        cur = torch.cuda.current_stream(cs.device)
        cs.wait_stream(cur)                               # inputs are ready on the issuing stream
        with torch.cuda.stream(cs):
            self._hold(wire_bytes, cs.device)
            fn()
        for t in tensors:
            t.record_stream(cs)
        ev = torch.cuda.Event()
        ev.record(cs)
        w = _EventWork(ev, cs.device)
        if not async_op:
            w.wait()
            return None
        return w

    def __enter__(self):
        frac = (self.W - 1) / self.W
        me = self

        def ag(block, out, w, group=None, async_op=False):
            nb = out.numel() * out.element_size()
            me.note("all_gather", nb, nb * frac)
            return me._issue(lambda: out.view(w, -1).copy_(block.reshape(1, -1).expand(w, -1)), (block, out), async_op, nb * frac, group)

        def rs(part_sums, out, rank, w, group=None, async_op=False):
            nb = part_sums.numel() * part_sums.element_size()
            me.note("reduce_scatter", nb, nb * frac)
            return me._issue(lambda: out.copy_(part_sums.view(w, -1)[rank].view_as(out)), (part_sums, out), async_op, nb * frac, group)

        def ar(t, w, group=None, op=None, tag="all_reduce", async_op=False):
            nb = t.numel() * t.element_size()
            me.note("all_reduce", nb, 2 * nb * frac)
            if me.wire_gbps:                              # emulated wire: it takes its turn on its communicator's stream
                return me._issue(lambda: None, (t,), async_op, 2 * nb * frac, group)
            return _Done() if async_op else None

        self._install(ag, rs, ar)
        return self


def sharded_stack_errors(world: int, edge_index: torch.Tensor, num_nodes: int, hub_mask, make_layers, x_full: torch.Tensor,
                         go_full: torch.Tensor, ref_out: torch.Tensor, ref_dx: torch.Tensor, ref_grads: list, dev,
                         relu_between: bool = True, schedule=None) -> dict:
    """A stack of sharded layers on ``world`` virtual ranks in exact lock step against a reference run of the same stack on the
    whole graph (normally this package's single-GPU layers, themselves checked by the ``-m gpu`` parity tests).

    ``make_layers(sg)`` -> the rank's list of ``dist.Sharded*Layer`` (same parameters on every rank); a rank's output rows
    are the next layer's input rows (``torch.relu`` behind every layer when ``relu_between``); the last output is driven
    backward with ``go_full[sg.own]``.  ``ref_out`` / ``ref_dx`` ``[N, F]``: the reference's output and input gradient;
    ``ref_grads[k]``: dict parameter name -> gradient of layer k (already summed over all nodes, as the all-reduce leaves it).

    Returns, as the MAX over ranks, for ``out``, ``dX`` and every ``layer<k>.<name>``: ``<name>`` = max |value - reference| /
    max |reference| and ``<name>.l2`` = ||value - reference|| / ||reference|| over the rank's rows (``out`` / ``dX``: the two
    norms are summed over the ranks first, i.e. the figure is the whole tensor's).  With ReLUs in the stack the max-abs figure
    of the GRADIENTS is not a rounding measure: an activation whose pre-activation is within rounding of zero may have either
    sign in two correct fp32 evaluations, and one flipped mask bit moves single gradient elements by their full size -- the L2
    figure and the no-ReLU stack are the ones to hold to a rounding-level bar."""
    from .schedule import DEFAULT

    def rel(a, r):
        return (a.detach() - r).abs().max() / r.abs().max().clamp(min=1e-30)

    def sq(a):
        return a.detach().double().pow(2).sum()

    with LockStep(world) as ls:
        sgs = [ND.ShardedGraph(edge_index, num_nodes, r, world, dev, hub_mask=hub_mask, schedule=schedule or DEFAULT)
               for r in range(world)]

        def run_rank(r):
            sg = sgs[r]
            layers = make_layers(sg)
            own = sg.own.to(x_full.device)
            x = x_full.detach()[own].to(dev).clone().requires_grad_(True)          # a fresh leaf, whatever x_full is
            h = x
            for layer in layers:
                h = layer(h)
                if relu_between:
                    h = torch.relu(h)
            h.backward(go_full.detach()[own].to(dev))
            ro, rx = ref_out[own].to(dev), ref_dx[own].to(dev)
            errs = {"out": rel(h, ro), "dX": rel(x.grad, rx)}
            norms = {"out": (sq(h - ro), sq(ro)), "dX": (sq(x.grad - rx), sq(rx))}
            for k, layer in enumerate(layers):
                for name, g in ref_grads[k].items():
                    got = getattr(layer, name).grad.reshape(g.shape)
                    errs[f"layer{k}.{name}"] = rel(got, g.to(dev))
                    errs[f"layer{k}.{name}.l2"] = (sq(got - g.to(dev)) / sq(g.to(dev)).clamp(min=1e-300)).sqrt()
            return errs, norms
        per_rank = ls.run(run_rank)
        passes = ls.passes
    names = sorted(per_rank[0][0])
    out = {n: max(float(e[n]) for e, _ in per_rank) for n in names}
    for n in ("out", "dX"):
        num = sum(float(nm[n][0]) for _, nm in per_rank)
        den = sum(float(nm[n][1]) for _, nm in per_rank)
        out[n + ".l2"] = (num / max(den, 1e-300)) ** 0.5
    out["lockstep_passes"] = passes
    return out


def gat_stack_reference(edge_index: torch.Tensor, num_nodes: int, params, x: torch.Tensor, go: torch.Tensor, relu: bool,
                        permute_seed=None):
    """The single-GPU GATConv stack (one head) on the whole graph: ``params`` = [(weight, att, bias), ...] on the device.
    Returns (out, dX, [{"weight", "att", "bias"} gradients per layer]).  ``permute_seed``: the same stack on the same graph
    with the COLUMNS OF THE EDGE LIST IN ANOTHER ORDER -- the same mathematics, other summation orders inside every row: the
    distance between two such runs is the fp32 noise floor of the stack on this data (a deep stack of random GAT layers is
    ill-conditioned in its backward: the score gradient alpha (<dOut_i, h_j> - D_i) cancels once the features are smooth)."""
    from .functional import gat_conv
    from .graph import CSRGraph
    ei = edge_index
    if permute_seed is not None:
        g = torch.Generator(device=ei.device).manual_seed(int(permute_seed))
        ei = ei[:, torch.randperm(ei.size(1), generator=g, device=ei.device)].contiguous()
    graph = CSRGraph(ei, num_nodes)
    ps = [tuple(t.detach().clone().requires_grad_(True) for t in p) for p in params]
    xin = x.detach().clone().requires_grad_(True)
    h = xin
    for W, a, b in ps:
        h = gat_conv(h, graph, W, a, b, heads=1, relu=relu)
    h.backward(go)
    return h.detach(), xin.grad, [{"weight": W.grad, "att": a.grad, "bias": b.grad} for W, a, b in ps]


def stack_distance(a, b) -> dict:
    """the figures of ``sharded_stack_errors`` between two whole-graph results ``(out, dX, grads)``"""
    def rel(p, r):
        return float((p - r).abs().max() / r.abs().max().clamp(min=1e-30))

    def l2(p, r):
        return float((p - r).double().norm() / r.double().norm().clamp(min=1e-300))
    out = {"out": rel(a[0], b[0]), "out.l2": l2(a[0], b[0]), "dX": rel(a[1], b[1]), "dX.l2": l2(a[1], b[1])}
    for k, (ga, gb) in enumerate(zip(a[2], b[2])):
        for n in gb:
            out[f"layer{k}.{n}"] = rel(ga[n], gb[n])
            out[f"layer{k}.{n}.l2"] = l2(ga[n], gb[n])
    return out
