"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/npi_gnn.h
declares, the host-side mirror of the PyG interface has the reference's parameter layout, and the
product path refuses to run without the GPU (there is no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

import npi_gnn_amd
from npi_gnn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "npi_gnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(npi_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m npi_gnn_amd.build` (or __graft_entry__.build())"


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared_symbols()
    assert len(names) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/npi_gnn.h but not exported"
    assert set(names) == set(_lib.PROTOTYPES), "ctypes prototypes out of sync with the header"


def test_ctypes_prototypes_have_the_headers_argument_lists():
    """every ctypes prototype has as many arguments as the declaration in include/npi_gnn.h, pointers where it has pointers
    and 64-bit / 32-bit / float scalars where it has those (a drifted prototype shifts every later argument silently)"""
    text = open(os.path.join(ROOT, "include", "npi_gnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = dict((m.group(1), m.group(2)) for m in re.finditer(r"\b(npi_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S))
    assert set(decls) == set(_lib.PROTOTYPES)
    for name, args in decls.items():
        args = " ".join(args.split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        want = []
        for a in params:
            if "*" in a:
                want.append(ctypes.c_void_p)
            elif a.startswith("int64_t"):
                want.append(ctypes.c_int64)
            elif a.startswith("float"):
                want.append(ctypes.c_float)
            elif a.startswith("int "):
                want.append(ctypes.c_int)
            else:
                raise AssertionError(f"{name}: cannot classify parameter {a!r}")
        assert _lib.PROTOTYPES[name][1] == want, f"{name}: ctypes prototype differs from the header's argument list"


def test_size_queries_without_gpu():
    lib = _lib.load()
    assert lib.npi_abi_version() == 4
    # the item size is an argument (a property of each CSR); npi_item_edges is only the HINT for a new CSR, a pure function:
    # 64-entry items below 2^22 entries of capacity, 256-entry items from there on
    T = 1 << 22
    assert lib.npi_item_edges(1000) == 64 and lib.npi_item_edges(T - 1) == 64
    assert lib.npi_item_edges(T) == 256 and lib.npi_item_edges(21_000_000) == 256
    for item in (64, 256):
        assert lib.npi_num_items(0, item) == 0 and lib.npi_num_items(1, item) == 1
        assert lib.npi_num_items(item, item) == 1 and lib.npi_num_items(item + 1, item) == 2
        assert lib.npi_num_items(T + 1, item) == T // item + 1
    assert lib.npi_num_items(1000, 128) == -1 and lib.npi_segsum_carry_elems(1000, 100, 256) == -1   # no such item size
    assert lib.npi_csr_workspace_bytes(1000, 10) > 16 * 1000
    # two partial rows per WORKGROUP (4 items) + two span rows per 64 workgroups + the arrival counters
    assert lib.npi_segsum_carry_elems(1000, 64, 256) >= (2 * 4 + 2) * 256 + 4 and lib.npi_segsum_carry_elems(1000, 256, 256) >= (2 + 2) * 256
    assert lib.npi_segsum_carry_elems(21_000_000, 256, 256) < 2 * 82032 * 256       # a quarter of the per-item scratch it replaces
    assert lib.npi_linear_bwd_weight_workspace_elems(1000, 256, 256) >= 256 * 256


def test_argument_errors_do_not_need_a_gpu():
    lib = _lib.load()
    rc = lib.npi_segsum_ex(None, None, None, 64, None, -1, 0, None, 0, None, 0, None, 0, 4, 0, 0, None, None, None, None)
    assert rc == -1
    assert b"npi_segsum" in lib.npi_last_error()


def test_every_family_rejects_bad_sizes_before_touching_the_gpu():
    """status -1 and a message naming the entry point; nothing is launched (this box has no GPU)"""
    lib = _lib.load()
    N = None
    calls = {
        "npi_csr_build_ex": lambda: lib.npi_csr_build_ex(N, N, -1, 4, 4, 1, 0, 1, N, N, N, N, N, 64, N, N, 0, N),
        # build flags: NPI_CSR_DROP_EQUAL | NPI_CSR_SORT_COLUMNS, nothing else
        "npi_csr_build_ex (flags)": lambda: lib.npi_csr_build_ex(8, 8, 4, 4, 4, 1, 0, 4, 8, 8, 8, 8, 8, 64, 8, 8, 1 << 20, N),
        # an item size that does not exist is refused by the build and by every consumer of item_row
        "npi_csr_build_ex (item)": lambda: lib.npi_csr_build_ex(8, 8, 4, 4, 4, 1, 0, 1, 8, 8, 8, 8, 8, 100, 8, 8, 1 << 20, N),
        "npi_csr_filter": lambda: lib.npi_csr_filter(8, 8, 8, 8, 8, 8, 4, 8, 8, 8, 8, 8, 8, 0, 8, 8, N),
        "npi_segsum_ex (item)": lambda: lib.npi_segsum_ex(8, 8, 8, 128, N, 4, 16, 8, 4, N, 0, 8, 4, 4, 0, 0, N, 8, N, N),
        "npi_gat_aggregate_scores": lambda: lib.npi_gat_aggregate_scores(8, 8, 8, 65, 4, 16, 8, 4, N, 0, 8, 4, 4, 8, 8, 8, N, 0, 8, N),
        "npi_gat_backward_fused_heads": lambda: lib.npi_gat_backward_fused_heads(8, 8, 8, 8, 0, 4, 16, 16, 4, N, 0, 16, 4, 16, 4, 1, 4, 16,
                                                                                 16, 0.2, 16, 16, N, N, N, 0, N),
        "npi_colsum": lambda: lib.npi_colsum(N, 0, -1, 8, N, N, 0, N),
        "npi_gat_scores": lambda: lib.npi_gat_scores(N, 0, N, 8, 0, 4, N, N, N),
        "npi_topk_score": lambda: lib.npi_topk_score(N, 0, N, 8, 0, N, N),
        "npi_readout_max_mean": lambda: lib.npi_readout_max_mean(N, 0, N, 2, 0, N, N),
        "npi_topk_gather_bwd": lambda: lib.npi_topk_gather_bwd(N, 0, N, N, N, -1, 8, N, 0, N, N, 0, N, N, N),
        "npi_subgraph_sizes": lambda: lib.npi_subgraph_sizes(N, N, N, N, -1, N, N, N, N),
        "npi_subgraph_features": lambda: lib.npi_subgraph_features(N, 0, 0, N, N, N, 1, 4, N, 0, N),
        "npi_subgraph_fill": lambda: lib.npi_subgraph_fill(N, N, N, N, 4, N, N, N, N, N, N, -1, 0, N, N),
        "npi_confusion_update": lambda: lib.npi_confusion_update(N, 0, 0, N, 4, N, N),
        # alpha read-back is a by-source, one-head, mapped mode: a forward call carrying it is refused (pointers are only
        # compared with NULL before that check, never dereferenced on the host)
        "npi_gat_aggregate_ex": lambda: lib.npi_gat_aggregate_ex(8, 8, 8, 64, 4, 16, 8, 4, N, 0, 8, 4, 1, 4, 8, 8, 8, 8, 0.2, 0, N, N, N,
                                                                 N, 8, N, 8, N),
        "npi_seg_rowsum_ex": lambda: lib.npi_seg_rowsum_ex(N, N, N, N, -1, 0, 1, N, N, 0, N),
        "npi_segsum_ex": lambda: lib.npi_segsum_ex(N, N, N, 64, N, 4, 16, 8, 4, 8, -5, 8, 4, 4, 0, 0, N, 8, N, N),       # bad split
        "npi_linear_fwd_ex": lambda: lib.npi_linear_fwd_ex(N, 0, N, 0, N, N, N, 0, 8, 0, 8, 0, 0, 0, N, 0, N, N),
        "npi_linear_bwd_data_ex": lambda: lib.npi_linear_bwd_data_ex(N, 0, N, 0, N, N, 0, 8, 8, -3, 0, 0, N, 0, N, N),
        "npi_linear_bwd_weight_ex": lambda: lib.npi_linear_bwd_weight_ex(N, 0, N, 0, N, 0, N, 8, 0, 8, N, 0, 0, 0, 1, N, N, N),
        # round 4: the row-dot epilogue serves one column tile (N = 128 / 256) only; the preparation launch wants K, N % 16 == 0
        "npi_linear_fwd_scores": lambda: lib.npi_linear_fwd_scores(16, 256, 16, 192, 16, 16, 192, 16, 16, 1000, 256, 192, N, 0, N, N),
        "npi_linear_prepare": lambda: lib.npi_linear_prepare(16, 256, 178, 128, 3, 0, 16, 1 << 20, N),
        "npi_hold_cus": lambda: lib.npi_hold_cus(1000, 10, N, N),
        "npi_gat_edge_grad_ex": lambda: lib.npi_gat_edge_grad_ex(N, N, N, 4, 16, N, 4, N, 0, N, 4, 0, 4, N, N, N, N, N, 0.2, 1, N, N, N),
    }
    for name, call in calls.items():
        assert call() == -1, name
        name = name.split()[0]
        stem = name[:-3] if name.endswith("_ex") else name           # npi_csr_build_ex reports as npi_csr_build
        assert stem.encode() in lib.npi_last_error(), (name, lib.npi_last_error())
    # a caller workspace that is too small is refused before anything is launched (status -3), and sized by a query
    assert lib.npi_linear_workspace_bytes(256, 256) >= 6 * 256 * 256 and lib.npi_linear_workspace_bytes(0, 8) == -1
    assert lib.npi_linear_fwd_ex(16, 256, 16, 256, N, N, 16, 256, 128, 256, 256, 0, 0, 0, 16, 100, N, N) == -3
    assert b"workspace" in lib.npi_last_error()
    assert lib.npi_linear_bwd_data_ex(16, 256, 16, 256, N, 16, 256, 128, 256, 256, 0, 0, 24, 10 ** 9, N, N) == -3   # misaligned
    assert lib.npi_linear_prepare(16, 256, 256, 256, 3, 0, 16, lib.npi_linear_workspace_bytes(256, 256), N) == -3     # both copies: 2 x
    assert lib.npi_linear_fwd_scores_supported(1000, 256, 256) == 1 and lib.npi_linear_fwd_scores_supported(1000, 256, 192) == 0
    # NPI_GEMM_WORKSPACE_PREPARED (8) without a workspace: refused like every call without one (ABI 3)
    assert lib.npi_linear_fwd_ex(16, 256, 16, 256, N, N, 16, 256, 128, 256, 256, 0, 0, 8, N, 0, N, N) == -3
    assert lib.npi_linear_bwd_weight_workspace_elems(-1, 8, 8) == -1
    # ABI 3: the workspace is REQUIRED (nothing is allocated inside a call); a null one is refused with the workspace status
    assert lib.npi_linear_fwd_ex(16, 256, 16, 256, N, N, 16, 256, 128, 256, 256, 0, 0, 0, N, 0, N, N) == -3
    # the row-dot epilogue needs >= 4 k-steps per tile: K = 32 is refused, K = 64 is served
    assert lib.npi_linear_fwd_scores_supported(1000, 32, 128) == 0 and lib.npi_linear_fwd_scores_supported(1000, 64, 128) == 1


def test_graphed_stack_refuses_layers_it_cannot_honour():
    """GraphedStack runs every layer over ONE self-loop-augmented graph; a GCNConv(normalize=False) or a
    SAGEConv(concat=True) inside it used to compute the default layer silently.  Refused before anything touches the GPU."""
    import npi_gnn_amd as npi
    for conv, what in ((npi.GCNConv(8, 8, normalize=False), "normalize=False"), (npi.SAGEConv(8, 8, concat=True), "concat=True"),
                       (npi.SAGEConv(8, 8, normalize=True), "normalize=True")):
        with pytest.raises(ValueError, match=what.replace("=", "=")):
            npi.GraphedStack([npi.SAGEConv(8, 8), conv], None, None)
    gat = npi.GATConv(8, 8, dropout=0.5)
    with pytest.raises(ValueError, match="dropout"):
        npi.GraphedStack([gat], None, None)
    with pytest.raises(TypeError, match="CSRGraph"):                      # (the layers are fine: the graph is looked at next)
        npi.GraphedStack([gat.eval(), npi.SAGEConv(8, 8), npi.GCNConv(8, 8)], None, None)


def test_no_module_level_switch_on_the_layer_path():
    """the default results of a model must not depend on module globals another import can move.  The three
    that survived round 5 -- functional.GEMM_FLAGS, functional.F16X2_MIN_ROWS, graph.ITEM_SWITCH_ENTRIES -- are gone: the GEMM
    arithmetic is a per-call ``flags`` argument / ``Schedule.f16x2_min_rows``, the item size ``CSRGraph(item=)``; and no module of
    the package assigns an upper-case module attribute from inside a function (``global X`` / ``module.X = ...``) except the
    documented debug and profiling hooks."""
    import ast
    import npi_gnn_amd.functional as NF
    import npi_gnn_amd.graph as NG
    from npi_gnn_amd.schedule import DEFAULT, Schedule
    for mod, names in ((NF, ("GEMM_FLAGS", "F16X2_MIN_ROWS")), (NG, ("ITEM_SWITCH_ENTRIES",))):
        for n in names:
            assert not hasattr(mod, n), f"{mod.__name__}.{n} is back"
    assert DEFAULT.f16x2_min_rows == 100_000 and DEFAULT.aggregate_first_backward is True
    with pytest.raises(Exception):
        DEFAULT.f16x2_min_rows = 0                                  # a Schedule is frozen: a layer's arrangement cannot be moved under it
    assert Schedule.__dataclass_params__.frozen
    hooks = {"_DEBUG", "_PROFILE", "_PROFILE_TAGS", "_PROFILE_GEMM", "_COMM_PROFILE", "ALWAYS_COMMUNICATE", "_LIB"}
    pkg = os.path.join(ROOT, "npi_gnn_amd")
    for fn in sorted(os.listdir(pkg)):
        if not fn.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(pkg, fn)).read())
        upper_globals = {t.id for node in tree.body if isinstance(node, (ast.Assign, ast.AnnAssign))
                         for t in (node.targets if isinstance(node, ast.Assign) else [node.target])
                         if isinstance(t, ast.Name) and t.id.isupper()}
        # mutable-by-design module state must be declared with `global` somewhere: none beyond the hooks
        declared = {n for node in ast.walk(tree) if isinstance(node, ast.Global) for n in node.names}
        assert declared <= hooks | {"_lib", "_err"}, (fn, declared - hooks)
        # and no upper-case module constant is read through `functional.X` / `graph.X` by another module of the package
        src = open(os.path.join(pkg, fn)).read()
        for other, names in (("NF", ("GEMM_FLAGS", "F16X2_MIN_ROWS")), ("NG", ("ITEM_SWITCH_ENTRIES",))):
            for n in names:
                assert n not in src, (fn, n)
        del upper_globals


def test_the_library_allocates_nothing_and_keeps_no_state():
    """ABI 3: no hipMalloc* / hipFree* among the library's undefined symbols, no getenv, and neither the
    header nor the export list carries a process-wide setter or a legacy GEMM entry point."""
    import subprocess
    und = subprocess.run(["nm", "-D", "-u", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for bad in ("hipMalloc", "hipFree", "hipHostMalloc", "getenv"):
        assert bad not in und, [l for l in und.splitlines() if bad in l]
    exported = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = {l.split()[-1] for l in exported.splitlines() if " T " in l and l.split()[-1].startswith("npi_")}
    header = open(os.path.join(ROOT, "include", "npi_gnn.h")).read()
    gone = ("npi_gemm_mode", "npi_dw_shared", "npi_small_graph_entries", "npi_linear_fwd", "npi_linear_bwd_data", "npi_linear_bwd_weight",
            "npi_linear_fwd_t", "npi_linear_bwd_data_t", "npi_linear_bwd_weight_t", "npi_csr_build", "npi_segsum", "npi_gat_aggregate",
            "npi_gat_edge_grad", "npi_gat_rowdot_colsum", "npi_permute_f32")
    declared = set(_declared_symbols())
    for name in gone:
        assert name not in names and name not in declared, name
    assert declared == names == set(_lib.PROTOTYPES), (declared ^ names, names ^ set(_lib.PROTOTYPES))


def test_modules_mirror_pyg_parameter_layout():
    conv = npi_gnn_amd.SAGEConv(178, 128)
    sd = conv.state_dict()
    assert list(sd) == ["weight", "bias"]
    assert tuple(sd["weight"].shape) == (178, 128) and tuple(sd["bias"].shape) == (128,)
    bound = 1.0 / (178 ** 0.5)
    assert float(sd["weight"].abs().max()) <= bound and float(sd["bias"].abs().max()) <= bound
    g = npi_gnn_amd.GCNConv(178, 64)
    assert tuple(g.weight.shape) == (178, 64) and float(g.bias.abs().max()) == 0.0
    # a reference-style checkpoint slice loads unchanged
    conv.load_state_dict({"weight": torch.zeros(178, 128), "bias": torch.ones(128)})
    assert float(conv.bias.sum()) == 128.0


def test_no_cpu_fallback():
    conv = npi_gnn_amd.SAGEConv(8, 4)
    x = torch.randn(5, 8)
    ei = torch.tensor([[0, 1, 2], [1, 2, 3]])
    with pytest.raises(npi_gnn_amd.NpiError):
        conv(x, ei)
    with pytest.raises(npi_gnn_amd.NpiError):
        npi_gnn_amd.CSRGraph(ei, 5)
    from npi_gnn_amd import metrics, pool, subgraph
    with pytest.raises(npi_gnn_amd.NpiError):
        pool.topk_pool(x, ei, torch.zeros(5, dtype=torch.long), torch.randn(1, 8))
    with pytest.raises(npi_gnn_amd.NpiError):
        pool.global_max_mean_pool(x, torch.zeros(5, dtype=torch.long))
    with pytest.raises(npi_gnn_amd.NpiError):
        subgraph.InteractionGraph(torch.tensor([[0, 3]]), torch.tensor([True]), torch.randn(5, 8))
    with pytest.raises(npi_gnn_amd.NpiError):
        metrics.confusion_update(torch.randn(4, 2), torch.zeros(4, dtype=torch.long), torch.zeros(4, dtype=torch.long))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "npi_gnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"
