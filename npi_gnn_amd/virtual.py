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

        def all_reduce(t, w, group=None, op=None, tag=""):
            mx = op == tdist.ReduceOp.MAX
            res = me._collective(t, (lambda v: torch.stack(v).max(0)[0]) if mx else (lambda v: torch.stack(v).sum(0)))
            if res is not None:
                t.copy_(res)

        self._install(all_gather_rows, reduce_scatter_rows, all_reduce)
        return self

    def run(self, run_rank: Callable[[int], object]) -> list:
        for _ in range(self.max_passes):
            self.passes += 1
            out, complete = [], True
            for r in range(self.W):
                self.state.update(rank=r, k=0, valid=True)
                out.append(run_rank(r))
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
    the dependency graph of the real run; None: plain copies on the compute stream."""

    def __init__(self, W: int, copy_stream: "torch.cuda.Stream | None" = None):
        self.W, self.log, self.copy_stream = int(W), {}, copy_stream

    def note(self, kind, nbytes, wire):
        e = self.log.setdefault(kind, {"calls": 0, "payload_bytes": 0, "wire_bytes_per_rank": 0})
        e["calls"] += 1
        e["payload_bytes"] += nbytes
        e["wire_bytes_per_rank"] += wire

    def _issue(self, fn, tensors, async_op):
        """run ``fn`` (the stand-in copy) where the collective would run"""
        cs = self.copy_stream
        if cs is None:
            fn()
            return _Done() if async_op else None
        cur = torch.cuda.current_stream(cs.device)
        cs.wait_stream(cur)                               # inputs are ready on the issuing stream
        with torch.cuda.stream(cs):
            fn()
        for t in tensors:
            t.record_stream(cs)
        ev = torch.cuda.Event()
        ev.record(cs)

        class _Work:
            def wait(self_inner):
                torch.cuda.current_stream(cs.device).wait_event(ev)
                return True
        w = _Work()
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
            return me._issue(lambda: out.view(w, -1).copy_(block.reshape(1, -1).expand(w, -1)), (block, out), async_op)

        def rs(part_sums, out, rank, w, group=None, async_op=False):
            nb = part_sums.numel() * part_sums.element_size()
            me.note("reduce_scatter", nb, nb * frac)
            return me._issue(lambda: out.copy_(part_sums.view(w, -1)[rank].view_as(out)), (part_sums, out), async_op)

        def ar(t, w, group=None, op=None, tag="all_reduce"):
            nb = t.numel() * t.element_size()
            me.note("all_reduce", nb, 2 * nb * frac)

        self._install(ag, rs, ar)
        return self


def sharded_stack_errors(world: int, edge_index: torch.Tensor, num_nodes: int, hub_mask, make_layers, x_full: torch.Tensor,
                         go_full: torch.Tensor, ref_out: torch.Tensor, ref_dx: torch.Tensor, ref_grads: list, dev,
                         relu_between: bool = True, schedule=None) -> dict:
    """A stack of sharded layers on ``world`` virtual ranks in exact lock step against a reference run of the same stack on the
    whole graph (normally this package's single-GPU layers, themselves checked by the ``-m gpu`` parity tests).

    ``make_layers(sg)`` -> the rank's list of ``dist.Sharded*Layer`` (same parameters on every rank); a rank's output rows
    are the next layer's input rows (``torch.relu`` in between when ``relu_between``); the last output is driven backward
    with ``go_full[sg.own]``.  ``ref_out`` / ``ref_dx`` ``[N, F]``: the reference's output and input gradient;
    ``ref_grads[k]``: dict parameter name -> gradient of layer k (already summed over all nodes, as the all-reduce leaves it).
    Returns the MAX over ranks of |value - reference| / max |reference| for ``out``, ``dX`` and every ``layer<k>.<name>``."""
    from .schedule import DEFAULT

    def rel(a, r):
        return (a.detach() - r).abs().max() / r.abs().max().clamp(min=1e-30)

    with LockStep(world) as ls:
        sgs = [ND.ShardedGraph(edge_index, num_nodes, r, world, dev, hub_mask=hub_mask, schedule=schedule or DEFAULT)
               for r in range(world)]

        def run_rank(r):
            sg = sgs[r]
            layers = make_layers(sg)
            own = sg.own.to(x_full.device)
            x = x_full[own].to(dev).requires_grad_(True)
            h = x
            for k, layer in enumerate(layers):
                h = layer(h)
                if relu_between:
                    h = torch.relu(h)
            h.backward(go_full[own].to(dev))
            errs = {"out": rel(h, ref_out[own].to(dev)), "dX": rel(x.grad, ref_dx[own].to(dev))}
            for k, layer in enumerate(layers):
                for name, g in ref_grads[k].items():
                    errs[f"layer{k}.{name}"] = rel(getattr(layer, name).grad.reshape(g.shape), g.to(dev))
            return errs
        per_rank = ls.run(run_rank)
        passes = ls.passes
    names = sorted(per_rank[0])
    out = {n: max(float(e[n]) for e in per_rank) for n in names}
    out["lockstep_passes"] = passes
    return out
