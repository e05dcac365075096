"""The caller of the conv hot path, mirrored: the reference's ``Net_1``
(``src/classes.py:45-82``) built from this package's modules, the batch source that replaces its
PyG ``DataLoader`` over pre-built subgraph files (``src/train_with_twoDataset.PY:142-143``) with
device-side extraction from target pairs, and its training loop (``:46-57`` and ``:154-184``)
followed statement by statement.  Parameter names and shapes equal the reference's, so
``Net_1.load_state_dict(torch.load('result/<proj>/model_<k>_fold/<epoch>'))`` works unchanged.
"""
from __future__ import annotations

import time
from typing import Callable, Optional

import torch
import torch.nn.functional as F

from . import head as NH
from . import metrics as NM
from . import pool as NP
from .graph import GraphBatch
from .nn import SAGEConv
from .subgraph import InteractionGraph


class Net_1(torch.nn.Module):
    """Reference ``src/classes.py:45-82``: three SAGEConv(., 128) + TopKPooling(128, 0.5) stages, the
    ``[global_max_pool || global_mean_pool]`` readout after each summed, then the 256-128-64-2 MLP with
    dropout 0.5 after ``lin1`` and ``log_softmax``."""

    #: what follows lin3 (head.mlp_head's ``activation``); the one-output variant below overrides it
    head_activation = "log_softmax"

    def __init__(self, num_node_features, num_of_classes=2, dropout: float = 0.5):
        super().__init__()
        self.dropout = dropout                                # 0.5 in the reference (src/classes.py:76)
        self.conv1 = SAGEConv(num_node_features, 128)
        self.pool1 = NP.TopKPooling(128, ratio=0.5, padded_edges=True)
        self.conv2 = SAGEConv(128, 128)
        self.pool2 = NP.TopKPooling(128, ratio=0.5, padded_edges=True)
        self.conv3 = SAGEConv(128, 128)
        self.pool3 = NP.TopKPooling(128, ratio=0.5, padded_edges=True)
        self.lin1 = torch.nn.Linear(256, 128)
        self.lin2 = torch.nn.Linear(128, 64)
        self.lin3 = torch.nn.Linear(64, num_of_classes)

    def forward(self, data):
        # one explicit object travels through the layers (graph.GraphBatch); a PyG-style batch (x, edge_index, batch,
        # num_graphs attributes) is wrapped -- then without the optional knowledge (host sizes, symmetry, padded features)
        gb = data if isinstance(data, GraphBatch) else \
            GraphBatch(data.x, data.edge_index, data.batch, getattr(data, "num_graphs", None))
        readouts = []
        for conv, pool in ((self.conv1, self.pool1), (self.conv2, self.pool2), (self.conv3, self.pool3)):
            gb = conv(gb, relu=True)                         # F.relu(conv(x, edge_index)), the ReLU in the GEMM's epilogue
            gb, _, _ = pool(gb)                              # x, edge_index, _, batch, _, _ = pool(x, edge_index, None, batch)
            # torch.cat([gmp(x, batch), gap(x, batch)], dim=1) as ONE kernel (src/classes.py:64,68,72)
            readouts.append(NP.global_max_mean_pool(gb))
        if NH.head_dims_ok(self.lin1.in_features, self.lin1.out_features, self.lin2.out_features, self.lin3.out_features):
            # x1 + x2 + x3 and the whole MLP head (src/classes.py:74-80) in one forward / two backward launches
            return NH.mlp_head(readouts, self.lin1, self.lin2, self.lin3, self.dropout, self.training,
                               activation=self.head_activation)
        x = readouts[0] + readouts[1] + readouts[2]
        x = F.relu(self.lin1(x))
        x = F.dropout(x, p=self.dropout, training=self.training)
        x = F.relu(self.lin2(x))
        x = self.lin3(x)
        return torch.sigmoid(x) if self.head_activation == "sigmoid" else F.log_softmax(x, dim=-1)


class Net_1_onlyOneOutput(Net_1):
    """The reference's one-output variant (``src/train_with_twoDataset_modelOnlyOneOutput.py:45-82``): the same three conv /
    pooling stages and readouts, ``lin3`` 64 -> 1 and ``torch.sigmoid`` instead of the two-class log-softmax; trained with
    ``F.binary_cross_entropy(output, y.float().view(-1, 1))`` (``:89-98``).  The head -- readout sum, MLP and the sigmoid --
    is the same fused forward / backward pair (``head.mlp_head(..., activation="sigmoid")``)."""
    head_activation = "sigmoid"

    def __init__(self, num_node_features, num_of_classes=2, dropout: float = 0.5):
        super().__init__(num_node_features, 1, dropout)       # (the reference ignores num_of_classes too: lin3 = Linear(64, 1))


class Batch(GraphBatch):
    """What a PyG ``DataLoader`` yields, as far as ``Net_1``, ``train()`` and the metrics look at it: a ``GraphBatch`` plus
    the labels ``y``."""
    __slots__ = ("y",)

    def __init__(self, gb: GraphBatch, y: torch.Tensor, sizes: Optional[torch.Tensor] = None):
        super().__init__(gb.x, gb.edge_index, gb.batch, int(y.numel()), sizes=sizes if sizes is not None else gb.sizes,
                         graph_ptr=gb.graph_ptr, symmetric=gb.symmetric, pad_base=gb.pad_base)
        self.y = y


class KeyLoader:
    """``DataLoader(dataset, batch_size=B)`` over a list of target pairs: every batch of enclosing subgraphs is
    built on the device when it is asked for (``InteractionGraph.batch``) instead of being read from the files
    the reference writes at dataset-build time.  ``shuffle()`` is ``dataset.shuffle()`` of
    ``src/train_with_twoDataset.PY:78-79``: ONE random permutation of the samples; the loader itself never
    shuffles (``:142`` passes no ``shuffle=``), so every epoch sees the same batches in the same order."""

    def __init__(self, ig: InteractionGraph, keys: torch.Tensor, y: torch.Tensor, batch_size: int = 200, _sizes=None):
        self.ig, self.keys, self.y, self.batch_size = ig, keys, y, int(batch_size)
        self.dataset = range(int(keys.size(0)))              # len(loader.dataset), as src/methods.py:85 uses it
        # a sample's size depends on its key only: ONE device read for the whole list, then every batch's totals are known
        # on the host and building a batch needs no read-back (the reference knows them too: its samples are files)
        self._nodes, self._pairs = _sizes if _sizes is not None else ig.sizes(keys)

    def shuffle(self, generator: Optional[torch.Generator] = None) -> "KeyLoader":
        perm = torch.randperm(self.keys.size(0), generator=generator)
        dperm = perm.to(self.keys.device)
        return KeyLoader(self.ig, self.keys[dperm], self.y[dperm], self.batch_size, (self._nodes[perm], self._pairs[perm]))

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for i in range(0, len(self.dataset), self.batch_size):
            k = self.keys[i:i + self.batch_size]
            nodes = self._nodes[i:i + self.batch_size]
            gb = self.ig.batch(k, n_nodes=int(nodes.sum()), n_pairs=int(self._pairs[i:i + self.batch_size].sum()))
            # sizes=: the graph sizes known on the host -- TopKPooling then needs no device read
            yield Batch(gb, self.y[i:i + self.batch_size], sizes=nodes)


def train(model, train_loader, optimizer, device) -> float:
    """``train()`` of ``src/train_with_twoDataset.PY:46-57``."""
    model.train()
    # `loss_all += data.num_graphs * loss.item()` with the running sum kept ON the device in float64 -- the same
    # arithmetic (f32 loss widened exactly, product and sum in f64, same order) without a host sync per batch
    loss_all = torch.zeros((), dtype=torch.float64, device=device)
    for data in train_loader:
        data = data.to(device)
        optimizer.zero_grad()
        output = model(data)
        loss = F.nll_loss(output, data.y)
        loss.backward()
        loss_all += data.num_graphs * loss.detach().double()
        optimizer.step()
    return loss_all.item() / len(train_loader.dataset)


class GraphedEpoch:
    """``train()`` with every batch's whole step -- forward, loss, backward, Adam -- captured into its own HIP graph.

    Possible because (i) the reference's loader never reshuffles (``src/train_with_twoDataset.PY:142``), so epoch after
    epoch the same 84 batches arrive in the same order: their input tensors are built ONCE and kept; (ii) with the graph
    sizes known on the host no layer of ``Net_1`` reads anything back from the device (``pool.NO_SYNC``: padded edge
    lists), so a step is a fixed sequence of ~190 kernels.  The first epoch runs eagerly (it also warms every lazy
    initialisation up), the second captures each step and replays it at once, later epochs only replay: the host cost of
    ~190 launches per step (the step is host-bound: 2.5 ms for 1.65 ms of kernels) is paid once per batch instead of
    once per batch and epoch.  Numerically this is the same sequence of kernels as the eager loop.  The optimizer must be
    ``Adam(..., capturable=True)`` with a tensor learning rate (``fit(capture=True)`` makes it so): a captured step
    reads the current rate from that tensor, so the scheduler keeps working."""

    def __init__(self, model, train_loader, optimizer, device, capture_after: int = 1):
        self.model, self.optimizer, self.device = model, optimizer, device
        self.capture_after = capture_after                            # eager epochs before the steps are captured
        self.batches = [d.to(device) for d in train_loader]           # static inputs: fixed addresses for the captures
        for d in self.batches:
            # the input graph of a batch never changes either: its CSR is built here, once, instead of inside every step
            # (the reference's counterpart is the processed dataset file; the graphs AFTER each pooling layer depend on the
            # weights and are rebuilt inside the step)
            d.graph()
            d.segment_ptr()                                                   # likewise the segment starts of its batch vector
        self.n = len(train_loader.dataset)
        self.graphs = [None] * len(self.batches)
        self.losses = [None] * len(self.batches)
        self.pool = None
        self.epochs_done = 0

    def _eager(self, data):
        self.optimizer.zero_grad()
        loss = F.nll_loss(self.model(data), data.y)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def _capture(self, i, data):
        g = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(g, pool=self.pool):
            loss = F.nll_loss(self.model(data), data.y)
            loss.backward()
            self.optimizer.step()
        if self.pool is None:
            self.pool = g.pool()          # the graphs are replayed one after the other, always in this order: one pool
        self.graphs[i], self.losses[i] = g, loss.detach()

    def __call__(self) -> float:
        self.model.train()
        loss_all = torch.zeros((), dtype=torch.float64, device=self.device)
        for i, data in enumerate(self.batches):
            if self.epochs_done < self.capture_after:
                loss = self._eager(data)
            else:
                if self.graphs[i] is None:
                    self._capture(i, data)
                self.graphs[i].replay()
                loss = self.losses[i]
            loss_all += data.num_graphs * loss.double()
        self.epochs_done += 1
        return loss_all.item() / self.n


def fit(model, train_loader, test_loader, device, num_of_epoch: int = 50, LR: float = 0.001,
        L2_weight_decay: float = 0.001, log: Callable[[str], None] = print, eval_train: bool = True,
        capture: bool = False):
    """The epoch loop of ``src/train_with_twoDataset.PY:130-184``: Adam(lr, weight_decay), ``ExponentialLR(0.95)``
    stepped ONLY when the epoch loss rose (``:158-160``), train + test metrics every 5th epoch except the last
    (``:163-172``) and once more at the end (``:186-193``), best test MCC tracked.  Returns a dict with the final
    test metrics, the best-MCC epoch and the wall time.  ``capture``: the same loop with every batch's step replayed from
    a HIP graph after the first epoch (``GraphedEpoch``)."""
    if capture:
        # the same Adam; fused = one multi-tensor kernel per step instead of ~40 element-wise ones (the captured step is
        # GPU-bound on its ~200 small kernels), capturable + a tensor learning rate so that a replayed step sees the
        # scheduler's current rate
        optimizer = torch.optim.Adam(model.parameters(), lr=torch.tensor(LR, device=device), weight_decay=L2_weight_decay,
                                     capturable=True, fused=True)
        epoch_fn = GraphedEpoch(model, train_loader, optimizer, device)
    else:
        optimizer = torch.optim.Adam(model.parameters(), lr=LR, weight_decay=L2_weight_decay)
        epoch_fn = None
    scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer=optimizer, gamma=0.95)
    fmt = '{}, {} dataset, Accuracy: {:.5f}, Precision: {:.5f}, Sensitivity: {:.5f}, Specificity: {:.5f}, MCC: {:.5f}'
    start_time = time.time()
    MCC_max, epoch_MCC_max, lr_steps = -1, 0, 0
    loss_last = float('inf')
    history = []
    for epoch in range(num_of_epoch):
        loss = epoch_fn() if epoch_fn is not None else train(model, train_loader, optimizer, device)
        if loss > loss_last:
            scheduler.step()
            lr_steps += 1
        loss_last = loss
        history.append(loss)
        if (epoch + 1) % 5 == 0 and epoch != num_of_epoch - 1:
            if eval_train:
                log(fmt.format('Epoch: {:03d}'.format(epoch + 1), 'training',
                               *NM.Accuracy_Precision_Sensitivity_Specificity_MCC(model, train_loader, device)))
            m = NM.Accuracy_Precision_Sensitivity_Specificity_MCC(model, test_loader, device)
            log(fmt.format('Epoch: {:03d}'.format(epoch + 1), 'testing', *m))
            if m[4] > MCC_max:
                MCC_max, epoch_MCC_max = m[4], epoch + 1
    train_m = NM.Accuracy_Precision_Sensitivity_Specificity_MCC(model, train_loader, device) if eval_train else None
    if train_m:
        log(fmt.format('result', 'training', *train_m))
    test_m = NM.Accuracy_Precision_Sensitivity_Specificity_MCC(model, test_loader, device)
    log(fmt.format('result', 'testing', *test_m))
    if test_m[4] > MCC_max:
        MCC_max, epoch_MCC_max = test_m[4], num_of_epoch
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    wall = time.time() - start_time
    log('Time consuming: ' + str(wall))
    return {"test": test_m, "train": train_m, "MCC_max": MCC_max, "epoch_MCC_max": epoch_MCC_max, "seconds": wall,
            "loss": history, "lr_steps": lr_steps}
