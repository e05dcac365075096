import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# The capacity below which a CSR is cut into 64-entry items is 2^22 by default (npi_small_graph_entries).  The kernel test
# modules below carry cases "just above 2^20 entries" that exist to exercise the 256-ENTRY items (row ends on item boundaries,
# hub rows cut over many items): they run with the switch at 2^20, so that both item sizes stay covered at test-sized inputs.
_ITEMS_AT_2P20 = ("test_gpu_fuzz", "test_gpu_parity", "test_gpu_gat")


@pytest.fixture(autouse=True, scope="module")
def _item_size_switch(request):
    name = request.module.__name__.split(".")[-1]
    if name not in _ITEMS_AT_2P20:
        yield
        return
    from npi_gnn_amd._lib import load
    lib = load()
    prev = int(lib.npi_small_graph_entries(1 << 20))
    yield
    lib.npi_small_graph_entries(prev)
