"""The readout sum and the MLP head of the reference's ``Net_1`` (``src/classes.py:74-80``) as three launches:

    x = x1 + x2 + x3; x = relu(lin1(x)); x = dropout(x, p); x = relu(lin2(x)); x = lin3(x); log_softmax(x, -1)

With 200 rows per batch every library GEMM and element-wise kernel of the head is launch-latency bound (30 launches,
0.3 ms of a 1.6 ms training step); ``npi_mlp_head_fwd`` / ``npi_mlp_head_bwd`` (``csrc/head.hip``) do the same
arithmetic in one forward and two backward kernels.  The parameters stay the three ``torch.nn.Linear`` modules of the
model (``lin1 / lin2 / lin3``: the reference's checkpoints load unchanged); the dropout mask is drawn by
``Tensor.bernoulli_`` -- torch's generator, graph-capture safe -- so it is a Bernoulli(1 - p) mask scaled by 1 / (1 - p)
like ``F.dropout``'s, but not the same random bits for a given seed.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from ._lib import check, load, ptr, require_gpu, stream_ptr


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"the MLP head is float32 (got {t.dtype})")
    return t if t.is_contiguous() else t.contiguous()


def head_dims_ok(D0: int, D1: int, D2: int, D3: int) -> bool:
    return 0 < D0 <= 1024 and D0 % 4 == 0 and 0 < D1 <= 256 and D1 % 4 == 0 and 0 < D2 <= 256 and D2 % 4 == 0 and 0 < D3 <= 32


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, scale, n_r, act, *args):
        rs, (W1, b1, W2, b2, W3, b3) = args[:n_r], args[n_r:]
        dev = require_gpu(*rs, W1, b1, W2, b2, W3, b3, mask)
        for r in rs:
            if r.dtype != torch.float32:
                raise TypeError("the MLP head is float32")
        rs = [r.detach() if r.stride(1) == 1 else r.detach().contiguous() for r in rs]      # row-strided views are fine
        W1, b1, W2, b2, W3, b3 = (_f32(t.detach()) for t in (W1, b1, W2, b2, W3, b3))
        B, D0 = rs[0].shape
        D1, D2, D3 = W1.size(0), W2.size(0), W3.size(0)
        if W1.size(1) != D0 or W2.size(1) != D1 or W3.size(1) != D2 or any(r.shape != (B, D0) for r in rs):
            raise ValueError("MLP head: shapes of the readouts and the three layers do not chain")
        keep = any(ctx.needs_input_grad)            # (grad mode is off inside forward: ask the context)
        f32 = dict(dtype=torch.float32, device=dev)
        s = torch.empty((B, D0), **f32) if keep else None
        h1 = torch.empty((B, D1), **f32) if keep else None
        h2 = torch.empty((B, D2), **f32) if keep else None
        logp = torch.empty((B, D3), **f32)
        if mask is not None:
            mask = _f32(mask)
            if mask.shape != (B, D1):
                raise ValueError("MLP head: the dropout mask must be [B, D1]")
        pr = [(ptr(r), r.stride(0)) for r in rs] + [(None, 0)] * (3 - len(rs))
        check(load().npi_mlp_head_fwd(pr[0][0], pr[0][1], pr[1][0], pr[1][1], pr[2][0], pr[2][1], B, D0, ptr(W1), ptr(b1), D1,
                                      ptr(W2), ptr(b2), D2, ptr(W3), ptr(b3), D3, ptr(mask), float(scale), int(act), ptr(s), ptr(h1),
                                      ptr(h2), ptr(logp), stream_ptr(dev)), "npi_mlp_head_fwd")
        ctx.n_r, ctx.scale, ctx.act = n_r, float(scale), int(act)
        ctx.has_mask = mask is not None
        if keep:
            ctx.save_for_backward(W1, W2, W3, s, h1, h2, logp, *([mask] if mask is not None else []))
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        W1, W2, W3, s, h1, h2, logp, *rest = ctx.saved_tensors
        mask = rest[0] if ctx.has_mask else None
        dev = s.device
        dlogp = _f32(dlogp)
        B, D0 = s.shape
        D1, D2, D3 = W1.size(0), W2.size(0), W3.size(0)
        f32 = dict(dtype=torch.float32, device=dev)
        want_x = any(ctx.needs_input_grad[4:4 + ctx.n_r])
        ds = torch.empty((B, D0), **f32) if want_x else None
        dW1, db1 = torch.empty_like(W1), torch.empty(D1, **f32)
        dW2, db2 = torch.empty_like(W2), torch.empty(D2, **f32)
        dW3, db3 = torch.empty_like(W3), torch.empty(D3, **f32)
        lib = load()
        n_ws = int(lib.npi_mlp_head_workspace_elems(B, D1, D2, D3))
        ws = torch.empty(n_ws, **f32)
        check(lib.npi_mlp_head_bwd(B, D0, D1, D2, D3, ptr(W1), ptr(W2), ptr(W3), ptr(mask), ctx.scale, ctx.act, ptr(s), ptr(h1), ptr(h2),
                                   ptr(logp), ptr(dlogp), ptr(ds), ptr(dW1), ptr(db1), ptr(dW2), ptr(db2), ptr(dW3), ptr(db3),
                                   ptr(ws), n_ws, stream_ptr(dev)), "npi_mlp_head_bwd")
        grads_r = [ds if ctx.needs_input_grad[4 + i] else None for i in range(ctx.n_r)]
        return (None, None, None, None, *grads_r, dW1, db1, dW2, db2, dW3, db3)


ACTIVATIONS = {"log_softmax": 0, "sigmoid": 1}           # NPI_HEAD_LOG_SOFTMAX / NPI_HEAD_SIGMOID


def mlp_head(readouts: Sequence[torch.Tensor], lin1: torch.nn.Linear, lin2: torch.nn.Linear, lin3: torch.nn.Linear,
             p: float = 0.5, training: bool = False, mask: Optional[torch.Tensor] = None,
             activation: str = "log_softmax") -> torch.Tensor:
    """``log_softmax(lin3(relu(lin2(dropout(relu(lin1(sum(readouts))), p)))))``; up to three readouts ``[B, D0]``.
    ``mask`` (``[B, D1]`` of 0 / 1): the dropout mask to use instead of a fresh one (tests).
    ``activation="sigmoid"``: ``torch.sigmoid(lin3(...))`` instead of the log-softmax -- the head of the reference's one-output
    variant (``src/train_with_twoDataset_modelOnlyOneOutput.py:45-82``; ``lin3`` is 64 -> 1, the loss binary cross entropy)."""
    if activation not in ACTIVATIONS:
        raise ValueError(f"mlp_head: activation must be one of {sorted(ACTIVATIONS)}")
    readouts = list(readouts)
    if not 1 <= len(readouts) <= 3:
        raise ValueError("mlp_head takes one to three readouts")
    if any(l.bias is None for l in (lin1, lin2, lin3)):
        raise NotImplementedError("mlp_head: the three layers have biases in the reference")
    scale = 1.0
    if training and p > 0.0:
        if p >= 1.0:
            raise ValueError("mlp_head: dropout p must be < 1")
        if mask is None:
            mask = torch.empty((readouts[0].size(0), lin1.out_features), dtype=torch.float32,
                               device=readouts[0].device).bernoulli_(1.0 - p)
        scale = 1.0 / (1.0 - p)
    else:
        mask = None
    return _HeadFn.apply(mask, scale, len(readouts), ACTIVATIONS[activation], *readouts, lin1.weight, lin1.bias, lin2.weight,
                         lin2.bias, lin3.weight, lin3.bias)
