"""``GraphedStack``: forward + backward of a conv stack on a STATIC full-batch graph, captured into one HIP graph.

The reference's real workload is small (``src/train_with_twoDataset.PY:46-57``: batches of 200 enclosing subgraphs; the
bundled full graphs of BASELINE.json configs 1-3 have 5,085 / 1,992 nodes): a three-layer step is ~40 launches of 5-25 us of
GPU work each, and launched one by one it is bounded by the HOST (C2: 0.54 ms eager against 0.33 ms of GPU time).  When the
graph and the shapes do not change from step to step -- full-batch training or inference on one graph -- the whole step is a
fixed sequence of kernels on fixed addresses, which a HIP graph replays at the GPU's own pace.  ``net1.GraphedEpoch`` does
this for ``Net_1``'s mini-batches; this is the same thing for a plain stack of ``SAGEConv`` / ``GCNConv`` / ``GATConv``
modules, packaged so that the fast path is the one a user of the modules gets::

    convs = [npi.SAGEConv(178, 128), npi.SAGEConv(128, 128), npi.SAGEConv(128, 128)]
    stack = npi.GraphedStack(convs, npi.CSRGraph(edge_index, N), x)       # warms up, captures
    out = stack(x_new, grad_out)        # copies the inputs into the static buffers, replays: out, stack.x.grad, p.grad
    opt.step()                          # any optimizer: the parameters' .grad tensors are refreshed by every replay

Numerically a replay IS the eager step: the same kernels on the same streams in the same order (``stack.eager()`` runs it
uncaptured; ``tests/test_gpu_schedule.py`` holds the two bit-equal).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch

from . import functional as F_
from .graph import CSRGraph
from .nn import GATConv, GCNConv, SAGEConv


class GraphedStack:
    """``convs``: modules of this package, applied in order as ``h = act(conv(h, graph))``.
    ``graph``: the ``CSRGraph`` all of them aggregate over (built once; both orientations are taken here).
    ``x``: example input ``[N, F_in]`` -- a copy of it becomes the static input buffer ``self.x`` (a leaf that requires grad).
    ``grad_out``: example gradient of the stack's output (``[N, F_out]``; default: ones) -- the backward starts from the static
    buffer ``self.grad_out``; or ``loss``: a callable ``out -> scalar`` whose ``backward()`` drives it instead.
    ``relu``: ``F.relu`` behind every layer (fused into the layer where the layer can: SAGEConv's projection epilogue,
    GATConv's aggregation epilogue), as ``Net_1`` applies it (``src/classes.py:62,66,70``).
    ``optimizer``: stepped INSIDE the captured step when given (it must be capturable: ``Adam(..., capturable=True)`` with a
    tensor learning rate); otherwise step any optimizer after each call."""

    def __init__(self, convs: Sequence[torch.nn.Module], graph: CSRGraph, x: torch.Tensor, grad_out: Optional[torch.Tensor] = None,
                 loss: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, relu: bool = True, optimizer=None,
                 warmup: int = 3, capture: bool = True):
        self.convs, self.graph, self.relu, self.loss, self.optimizer = list(convs), graph, bool(relu), loss, optimizer
        for c in self.convs:
            if not isinstance(c, (SAGEConv, GCNConv, GATConv)):
                raise TypeError(f"GraphedStack: {type(c).__name__} is not a conv of this package")
            # every layer of a stack aggregates over ONE adjacency: the self-loop-augmented CSR of `graph`.  Layer settings that mean
            # another adjacency (PyG 1.4.2: GCNConv(normalize=False) and SAGEConv(concat=True) take the edge list as it is, no self
            # loop) or another epilogue order are refused here -- computing the default layer in their place would differ from
            # the module's own forward without a word
            if isinstance(c, GCNConv) and not c.normalize:
                raise ValueError("GraphedStack: GCNConv(normalize=False) aggregates raw edge weights over the edge list without self "
                                 "loops; a stack shares one self-loop-augmented graph -- call the module itself")
            if isinstance(c, SAGEConv) and c.concat:
                raise ValueError("GraphedStack: SAGEConv(concat=True) aggregates over the edge list without self loops; a stack shares "
                                 "one self-loop-augmented graph -- call the module itself")
            if isinstance(c, SAGEConv) and c.normalize and relu:
                raise ValueError("GraphedStack(relu=True): SAGEConv(normalize=True) normalises AFTER the projection; the fused ReLU "
                                 "would come before it -- build the stack with relu=False")
            if isinstance(c, GATConv) and c.dropout > 0.0 and c.training:
                raise ValueError("GraphedStack: GATConv(dropout > 0) in training mode draws a new mask every step; a captured step "
                                 "would replay one mask -- call eval() on the layer or the module itself")
        if not isinstance(graph, CSRGraph):
            raise TypeError("GraphedStack: graph must be a CSRGraph (a static graph is sorted once)")
        _ = graph.by_src                                            # the backward's orientation, before anything is captured
        # GCNConv's normalisation depends on the graph only: computed once, outside the step
        self._norms = [F_.GCNNorm(graph, None, c.improved) if isinstance(c, GCNConv) else None for c in self.convs]
        self.x = x.detach().clone().requires_grad_(True)
        self.grad_out = None
        self.out = None
        self._graph = None
        self._params = [p for c in self.convs for p in c.parameters()]
        self._warmup = max(int(warmup), 1)
        # one forward tells the output's shape (and warms every lazy initialisation up)
        if loss is None:
            with torch.no_grad():
                probe = self._forward(self.x.detach())
            self.grad_out = (torch.ones_like(probe) if grad_out is None else grad_out.detach().clone())
            del probe
        if capture:
            self.capture()
        else:
            for _ in range(self._warmup):
                self.eager()

    # ---- the step ------------------------------------------------------------------------------------------------------------
    def _forward(self, h: torch.Tensor) -> torch.Tensor:
        for conv, norm in zip(self.convs, self._norms):
            if isinstance(conv, GCNConv):
                h = F_.gcn_conv(h, None, conv.weight, conv.bias, norm=norm, schedule=conv.schedule)
                if self.relu:
                    h = torch.relu(h)
            else:
                h = conv(h, self.graph, relu=self.relu)
        return h

    def eager(self) -> torch.Tensor:
        """one step launched kernel by kernel (what a replay repeats): forward, backward, optional optimizer step"""
        for p in self._params:
            p.grad = None
        self.x.grad = None
        out = self._forward(self.x)
        if self.loss is not None:
            self.loss(out).backward()
        else:
            out.backward(self.grad_out)
        if self.optimizer is not None:
            self.optimizer.step()
        # detached: holding the output WITH its autograd graph would keep this step's AccumulateGrad nodes alive into the next
        # one -- nodes bound to the stream they were created on, which breaks a capture on another stream (capture_end faults)
        self.out = out.detach()
        return self.out

    def capture(self) -> None:
        """record the step once (on a side stream, as ``torch.cuda.graph`` requires; the layers' own second stream is forked
        and joined inside the capture)"""
        torch.cuda.synchronize(self.x.device)
        s = torch.cuda.Stream(device=self.x.device)
        s.wait_stream(torch.cuda.current_stream(self.x.device))
        with torch.cuda.stream(s):
            # the warm-up runs on the CAPTURE stream: its scratch buffers (keyed by stream) and autograd's per-leaf accumulation
            # nodes (bound to the stream of their first use) then belong to the stream that is captured
            for _ in range(self._warmup):
                self.eager()
            torch.cuda.synchronize(self.x.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                self.eager()
        torch.cuda.current_stream(self.x.device).wait_stream(s)
        self._graph = g
        self._grads = [p.grad for p in self._params]                # the tensors every replay rewrites
        self._xgrad = self.x.grad
        self._out = self.out

    # ---- use -----------------------------------------------------------------------------------------------------------------
    def replay(self) -> torch.Tensor:
        if self._graph is None:
            return self.eager()
        self._graph.replay()
        for p, g in zip(self._params, self._grads):                 # an optimizer's zero_grad(set_to_none=True) in between
            if p.grad is not g:
                p.grad = g
        if self.x.grad is not self._xgrad:
            self.x.grad = self._xgrad
        self.out = self._out                                         # (an eager() in between left its own output here)
        return self.out

    def __call__(self, x: Optional[torch.Tensor] = None, grad_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """copy new inputs into the static buffers (when given), run the step, return the static output tensor; the gradients
        are in ``self.x.grad`` and in every parameter's ``.grad`` (rewritten by the next call)"""
        if x is not None:
            with torch.no_grad():
                self.x.copy_(x)
        if grad_out is not None:
            if self.grad_out is None:
                raise ValueError("GraphedStack was built with a loss: there is no grad_out buffer")
            self.grad_out.copy_(grad_out)
        return self.replay()
