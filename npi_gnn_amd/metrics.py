"""Evaluation metrics of the reference (``src/methods.py:78-126``) with the per-element Python
comparison loop replaced by one confusion-matrix kernel per batch.

``Accuracy_Precision_Sensitivity_Specificity_MCC(model, loader, device)`` keeps the reference's
name, arguments, printed line and return value; the counts stay on the GPU until the loader is
exhausted (one device read per evaluation instead of one per sample).
"""
from __future__ import annotations

import torch

from ._lib import check, load, ptr, require_gpu, stream_ptr


def confusion_update(scores: torch.Tensor, y: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    """``counts [4] int64 (TP, FN, TN, FP) +=`` the batch; ``scores [B, C]`` float32 (log-probabilities or logits)."""
    dev = require_gpu(scores, y, counts)
    if scores.dtype != torch.float32 or y.dtype != torch.int64 or counts.dtype != torch.int64:
        raise TypeError("confusion_update: scores float32, y int64, counts int64")
    scores = scores.detach()
    if scores.stride(1) != 1:
        scores = scores.contiguous()
    y = y.contiguous()
    if y.numel() != scores.size(0) or counts.numel() != 4:
        raise ValueError("confusion_update: shape mismatch")
    check(load().npi_confusion_update(ptr(scores), scores.stride(0), scores.size(1), ptr(y), scores.size(0), ptr(counts),
                                      stream_ptr(dev)), "npi_confusion_update")
    return counts


def metrics_from_counts(TP: int, FN: int, TN: int, FP: int):
    """The formulas of ``src/methods.py:107-126`` -> (Accuracy, Precision, Sensitivity, Specificity, MCC)."""
    Accuracy = (TP + TN) / (TP + TN + FP + FN) if (TP + TN + FP + FN) != 0 else 0
    Precision = TP / (TP + FP) if (TP + FP) != 0 else 0
    Sensitivity = TP / (TP + FN) if (TP + FN) != 0 else 0
    den = ((TP + FP) * (TP + FN) * (TN + FP) * (TN + FN)) ** 0.5
    MCC = (TP * TN - FP * FN) / den if den != 0 else 0
    Specificity = TN / (FP + TN) if (FP + TN) != 0 else 0
    return Accuracy, Precision, Sensitivity, Specificity, MCC


def Accuracy_Precision_Sensitivity_Specificity_MCC(model, loader, device):
    """Drop-in for ``src/methods.py:87``: ``model(data)`` per batch of ``loader``, ``data.y`` the labels."""
    model.eval()
    counts = torch.zeros(4, dtype=torch.int64, device=device)
    with torch.no_grad():
        for data in loader:
            data = data.to(device)
            confusion_update(model(data).float(), data.y.to(torch.int64), counts)
    TP, FN, TN, FP = counts.tolist()
    from .graph import check_pending
    check_pending()                     # the evaluation's one device read: also the place dropped node ids surface
    print('TP: %d, FN: %d, TN: %d, FP: %d' % (TP, FN, TN, FP))
    return metrics_from_counts(TP, FN, TN, FP)


def accuracy(model, loader, device):
    """Drop-in for ``src/methods.py:78-85`` (``correct / len(loader.dataset)``)."""
    model.eval()
    counts = torch.zeros(4, dtype=torch.int64, device=device)
    with torch.no_grad():
        for data in loader:
            data = data.to(device)
            confusion_update(model(data).float(), data.y.to(torch.int64), counts)
    TP, FN, TN, FP = counts.tolist()
    return (TP + TN) / len(loader.dataset)
