"""ctypes binding of ``libnpi_gnn.so`` (the C ABI of ``include/npi_gnn.h``).

There is NO CPU fallback: every wrapper raises if the library is missing or if a tensor is not
on an AMD GPU.  PyTorch only lends device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

import torch  # noqa: F401  (must be imported first: the .so binds to torch's libamdhip64.so.7)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NPI_GNN_LIB") or os.path.join(HERE, "libnpi_gnn.so")   # env override: kernel A/B experiments

NPI_F32 = 0
NPI_BF16 = 1
NPI_GEMM_EXACT_F32 = 1      # flags of the npi_linear_*_ex entry points
NPI_GEMM_SPLIT_BF16 = 2
NPI_GEMM_A_ZERO_PADDED = 4   # A stored with zero pad columns up to a multiple of 128 (include/npi_gnn.h)
NPI_GEMM_WORKSPACE_PREPARED = 8   # the workspace already holds npi_linear_prepare's copy of this weight matrix
NPI_GEMM_SPLIT_F16X2 = 16         # two fp16 pieces per operand, three matrix products per tile pair (needs the row scales of A)
NPI_PREPARE_F16X2 = 4             # npi_linear_prepare(which | this): the fp16 x 2 planes of the weight matrix


def NPI_GEMM_RESERVE_CUS(n: int) -> int:
    """flag bits of npi_linear_fwd_ex / npi_linear_bwd_data_ex: leave ``n`` CUs (a multiple of 8, at most 128) to a kernel running
    beside the GEMM"""
    n = int(n)
    if n < 0 or n > 128 or n % 8:
        raise ValueError(f"NPI_GEMM_RESERVE_CUS: a multiple of 8 in [0, 128], got {n}")
    return (n // 8) << 8

_P = c_void_p
_I = c_int64

# name -> (restype, argtypes); mirrors include/npi_gnn.h one to one
PROTOTYPES = {
    "npi_last_error": (c_char_p, []),
    "npi_abi_version": (c_int, []),
    "npi_csr_workspace_bytes": (_I, [_I, _I]),
    "npi_item_edges": (_I, [_I]),
    "npi_num_items": (_I, [_I, _I]),
    "npi_csr_build_ex": (c_int, [_P, _P, _I, _I, _I, c_int, _I, c_int, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P]),
    "npi_edge_positions": (c_int, [_P, _P, _I, _I, _I, _P, _P]),
    "npi_segsum_carry_elems": (_I, [_I, _I, _I]),
    "npi_segsum_ex": (c_int, [_P, _P, _P, _I, _P, _I, _I, _P, _I, _P, _I, _P, _I, _I, c_int, c_int, _P, _P, _P, _P]),
    "npi_segsum_scales_supported": (c_int, [_I, c_int]),
    "npi_row_weight_sum": (c_int, [_P, _P, _I, _P, _P]),
    "npi_gcn_norm": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "npi_row_inv_count": (c_int, [_P, _I, _P, _P]),
    "npi_entry_col_scale": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "npi_entry_weights": (c_int, [_P, _P, _P, _P, _P, c_float, _I, _I, _P, _P]),
    "npi_relu_backward": (c_int, [_P, _I, _P, _I, _I, _I, _P, _I, _P]),
    "npi_l2_normalize_rows": (c_int, [_P, _I, _I, _I, c_float, _P, _I, _P, _P]),
    "npi_l2_normalize_rows_bwd": (c_int, [_P, _I, _P, _I, _P, _I, _I, c_float, _P, _I, _P]),
    "npi_colsum_workspace_elems": (_I, [_I, _I]),
    "npi_colsum": (c_int, [_P, _I, _I, _I, _P, _P, _I, _P]),
    "npi_linear_bwd_weight_workspace_elems": (_I, [_I, _I, _I]),
    "npi_linear_workspace_bytes": (_I, [_I, _I]),
    "npi_row_scales": (c_int, [_P, _I, _I, _I, _P, _P]),
    "npi_linear_fwd_ex": (c_int, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, c_int, c_int, c_int, _P, _I, _P, _P]),
    "npi_linear_bwd_data_ex": (c_int, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, c_int, c_int, _P, _I, _P, _P]),
    "npi_linear_prepare": (c_int, [_P, _I, _I, _I, c_int, c_int, _P, _I, _P]),
    "npi_hold_cus": (c_int, [c_int, _I, _P, _P]),
    "npi_linear_fwd_scores_supported": (c_int, [_I, _I, _I]),
    "npi_linear_fwd_scores": (c_int, [_P, _I, _P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _P, _I, _P, _P]),
    "npi_linear_bwd_data_rank2_supported": (c_int, [_I, _I, _I]),
    "npi_linear_bwd_data_rank2": (c_int, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P, _P]),
    "npi_gat_rank2_cols": (c_int, [_P, _I, _P, _I, _I, _P, _P]),
    "npi_gat_rank2_tail": (c_int, [_P, _P, _I, _P, _I, _I, _P, _I, _P, _P]),
    "npi_linear_bwd_weight_ex": (c_int, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _I, c_int, c_int, c_int, _P, _P, _P]),
    "npi_col_scales_workspace_elems": (_I, [_I, _I]),
    "npi_col_scales": (c_int, [_P, _I, _I, _I, _P, _P, _P, _I, _P]),
    "npi_conv_fwd": (c_int, [_P, _P, _P, _I, _P, _I, _I, _P, _I, _I, c_int, _P, _I, _P, _P, _I, _P, _P, _I, _I, _I, c_int, c_int, c_int,
                             c_int, _P, _I, _P]),
    "npi_conv_bwd": (c_int, [_P, _I, _P, _I, _P, _I, _I, _I, _I, c_int, c_int, _P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P, _I, _P, _I,
                             c_int, _P, _P, _P, _I, _P, _I, _P, _I, _P, _P]),
    "npi_gat_aggregate_fused": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, c_float, _P, c_int, _P, _P, _P, _P,
                                            _P]),
    "npi_gat_scores": (c_int, [_P, _I, _P, _I, _I, _I, _P, _P, _P]),
    "npi_gat_aggregate_ex": (c_int, [_P, _P, _P, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _P, _P, _P, c_float, c_int,
                                     _P, _P, _P, _P, _P, _P, _P, _P]),
    "npi_gat_pack_targets": (c_int, [_P, _P, _P, _P, _I, _P, _P]),
    "npi_gat_backward_fused_heads": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _P, c_float, _P,
                                                 _P, _P, _P, _P, _I, _P]),
    "npi_gat_rank1_add": (c_int, [_P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "npi_gat_rowdot": (c_int, [_P, _I, _P, _I, _P, _I, _I, _I, _P, _P]),
    "npi_gat_edge_grad_ex": (c_int, [_P, _P, _P, _I, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, c_float,
                                     c_int, _P, _P, _P]),
    "npi_seg_scan_workspace_elems": (_I, [_I, _I]),
    "npi_seg_rowsum_ex": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P]),
    "npi_gat_softmax_stats_ex": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, c_float, _P, _P, _P, _P, _I, _P]),
    "npi_gat_aggregate_scores": (c_int, [_P, _P, _P, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _P, c_int, _P, _P]),
    "npi_gat_rowdot_colsum_workspace_elems": (_I, [_I, _I, _I]),
    "npi_gat_rowdot_colsum_relu": (c_int, [_P, _I, _P, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P]),
    "npi_entry_transpose_map": (c_int, [_P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "npi_gat_att_grad_workspace_elems": (_I, [_I, _I, _I]),
    "npi_gat_att_grad": (c_int, [_P, _I, _P, _P, _I, _I, _I, _P, _P, _I, _P]),
    "npi_topk_score": (c_int, [_P, _I, _P, _I, _I, _P, _P]),
    "npi_graph_bounds": (c_int, [_P, _I, _I, _P, _P]),
    "npi_topk_select": (c_int, [_P, _P, _I, _I, c_float, _P, _P, _P, _P, _I, _P]),
    "npi_topk_sorted_workspace_bytes": (_I, [_I]),
    "npi_topk_select_sorted": (c_int, [_P, _P, _P, _I, _I, c_float, _P, _P, _P, _P, _I, _P]),
    "npi_topk_gather": (c_int, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P]),
    "npi_filter_adj_workspace_elems": (_I, [_I]),
    "npi_filter_adj": (c_int, [_P, _P, _I, _P, _P, _P, _P, _P, c_int, _P]),
    "npi_filter_adj_newpos_offset": (_I, [_I]),
    "npi_csr_filter_max_rows": (_I, []),
    "npi_csr_filter_workspace_elems": (_I, [_I]),
    "npi_csr_filter": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "npi_readout_max_mean": (c_int, [_P, _I, _P, _I, _I, _P, _P]),
    "npi_topk_gather_bwd": (c_int, [_P, _I, _P, _P, _P, _I, _I, _P, _I, _P, _P, _I, _P, _P, _P]),
    "npi_topk_gather_bwd_ex": (c_int, [_P, _I, _P, _P, _P, _I, _I, _P, _I, _P, _P, _I, _P, _P, _P]),
    "npi_topk_weight_grad_workspace_elems": (_I, [_I, _I]),
    "npi_topk_weight_grad": (c_int, [_P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _I, _P]),
    "npi_readout_max_mean_bwd": (c_int, [_P, _I, _P, _I, _I, _P, _P, _P, _I, _P]),
    "npi_readout_max_mean_bwd_ex": (c_int, [_P, _I, _P, _I, _I, _P, _P, _P, _I, _I, _P]),
    "npi_confusion_update": (c_int, [_P, _I, _I, _P, _I, _P, _P]),
    "npi_mlp_head_fwd": (c_int, [_P, _I, _P, _I, _P, _I, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P, c_float, c_int,
                                 _P, _P, _P, _P, _P]),
    "npi_mlp_head_workspace_elems": (_I, [_I, _I, _I, _I]),
    "npi_mlp_head_bwd": (c_int, [_I, _I, _I, _I, _I, _P, _P, _P, _P, c_float, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                 _P, _I, _P]),
    "npi_subgraph_sizes": (c_int, [_P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "npi_subgraph_fill": (c_int, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "npi_subgraph_features": (c_int, [_P, _I, _I, _P, _P, _P, _I, _I, _P, _I, _P]),
}

_lib = None


class NpiError(RuntimeError):
    pass


def load(path: str = LIB_PATH) -> ctypes.CDLL:
    """dlopen the C-ABI library and attach prototypes.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise NpiError(
            f"{path} is missing: the HIP extension has not been built "
            "(run `python -m npi_gnn_amd.build`).  npi_gnn_amd has no CPU fallback.")
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().npi_last_error()
        raise NpiError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def require_gpu(*tensors) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise NpiError("npi_gnn_amd kernels run on MI355X (HIP) tensors only; got a "
                           f"{t.device} tensor.  There is no CPU fallback.")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise NpiError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device) -> int:
    """hipStream_t of torch's current stream on ``device`` (the raw-handle call is ~20x cheaper than building a
    ``torch.cuda.Stream`` object; the small-batch step makes some 20 of these)"""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream
