"""shared by the parity tests"""
import torch

GRAD_REL = 1e-5     # parameter gradients (dW, db, d att): max |diff| / max |reference|, the reference in fp64


def rel_max(got: torch.Tensor, ref: torch.Tensor) -> float:
    """max |got - ref| / max |ref|, in fp64 on the host"""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp(min=1e-300))
