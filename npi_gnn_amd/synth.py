"""Synthetic ncRNA-protein bipartite interaction graphs shaped like the reference's
(SURVEY.md 8(d), config C4/C5): node ids [0, n_rna) are ncRNAs, [n_rna, N) proteins (~10:1 as in
NPInter2: 4,636 : 449); every undirected pair is emitted in both directions like reference
``src/classes.py:701-704``; no duplicate pairs, no self loops; the protein side is Zipf-skewed so
that the heaviest protein holds ~5.4 % of all pairs (NPInter2: 1,121 of 20,824), the RNA side is
uniform.  Pure numpy on the host -- input construction, not part of the timed path.
"""
from __future__ import annotations

import numpy as np
import torch


def _zipf_probs(n: int, top_share: float) -> np.ndarray:
    """p_i ~ 1/(i + q), q solved so that p_0 = top_share."""
    i = np.arange(n, dtype=np.float64)
    lo, hi = 1e-3, 1e4
    for _ in range(80):
        q = (lo * hi) ** 0.5
        p = 1.0 / (i + q)
        share = p[0] / p.sum()
        if share > top_share:
            lo = q
        else:
            hi = q
    p = 1.0 / (i + q)
    return p / p.sum()


def protein_mask(num_nodes: int) -> torch.Tensor:
    """bool [N]: the protein side of ``bipartite_edge_index(num_nodes, ...)`` (the last tenth of the ids)"""
    n_prot = max(1, num_nodes // 10)
    return torch.arange(num_nodes) >= num_nodes - n_prot


def bipartite_edge_index(num_nodes: int, num_directed_edges: int, seed: int = 20260310,
                         top_share: float = 0.054) -> torch.Tensor:
    """LongTensor [2, E] (E = num_directed_edges, even), both directions, unsorted (shuffled)."""
    assert num_directed_edges % 2 == 0
    pairs = num_directed_edges // 2
    n_prot = max(1, num_nodes // 10)
    n_rna = num_nodes - n_prot
    rng = np.random.default_rng(seed)
    p = _zipf_probs(n_prot, top_share)
    # a protein cannot have more distinct partners than there are RNAs
    cdf = np.cumsum(p)
    got = np.empty(0, dtype=np.int64)
    need = pairs
    while need > 0:
        m = int(need * 1.15) + 1024
        rna = rng.integers(0, n_rna, size=m, dtype=np.int64)
        prot = np.searchsorted(cdf, rng.random(m), side="right").clip(0, n_prot - 1).astype(np.int64)
        key = np.concatenate([got, rna * n_prot + prot])
        got = np.unique(key)
        if got.size > pairs:
            got = rng.permutation(got)[:pairs]
        need = pairs - got.size
    rna = got // n_prot
    prot = got % n_prot + n_rna
    perm = rng.permutation(pairs)
    rna, prot = rna[perm], prot[perm]
    src = np.concatenate([rna, prot])
    dst = np.concatenate([prot, rna])
    perm2 = rng.permutation(2 * pairs)
    return torch.from_numpy(np.stack([src[perm2], dst[perm2]]))


def bipartite_edge_index_device(num_nodes: int, num_directed_edges: int, device, seed: int = 2,
                                top_share: float = 0.054) -> torch.Tensor:
    """The same distribution drawn ON THE DEVICE with torch's generator (another draw than ``bipartite_edge_index`` of the same
    seed: numpy's stream is not reproduced): 1-2 s for 100M directed edges where the host generator takes a minute.  For side
    measurements that need a graph of the C5 SHAPE quickly; fixtures, parity tests and PMC passes use the host generator."""
    assert num_directed_edges % 2 == 0
    pairs = num_directed_edges // 2
    n_prot = max(1, num_nodes // 10)
    n_rna = num_nodes - n_prot
    cdf = torch.from_numpy(np.cumsum(_zipf_probs(n_prot, top_share))).to(device)
    g = torch.Generator(device=device).manual_seed(int(seed))
    got = torch.empty(0, dtype=torch.int64, device=device)
    need = pairs
    while need > 0:
        m = int(need * 1.15) + 1024
        rna = torch.randint(0, n_rna, (m,), generator=g, device=device)
        prot = torch.searchsorted(cdf, torch.rand(m, generator=g, device=device, dtype=torch.float64), right=True).clamp_(max=n_prot - 1)
        got = torch.unique(torch.cat([got, rna * n_prot + prot]))
        del rna, prot
        if got.numel() > pairs:
            got = got[torch.randperm(got.numel(), generator=g, device=device)[:pairs]]
        need = pairs - got.numel()
    got = got[torch.randperm(pairs, generator=g, device=device)]
    rna = got // n_prot
    prot = got % n_prot + n_rna
    del got
    perm = torch.randperm(2 * pairs, generator=g, device=device)
    return torch.stack([torch.cat([rna, prot])[perm], torch.cat([prot, rna])[perm]])
