"""The long randomised campaigns (marker `slow`: NOT selected by the driver's `-m gpu` run, whose 1,200 s budget the GPU suite must
stay well inside -- VERDICT r5 item 7).  Run them with `pytest -m slow` on a GPU box; `tests/test_gpu_fuzz.py` runs the same
generators with half the cases under `-m gpu`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.slow
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, cases, seed, timeout):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(cases), str(seed)], capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0 and f"{cases} cases ok" in p.stdout, (p.stdout + p.stderr)[-2000:]


def test_gat_random_campaign_full():
    _run("fuzz_gat.py", 60, 11, 900)


def test_sharded_layers_random_campaign_full():
    _run("fuzz_dist.py", 24, 7, 1200)


def test_eight_virtual_ranks_at_a_quarter_of_c4_gat(dev):
    """the GATConv case of tests/test_dist_gpu.py's 8-rank lock-step run (its SAGEConv case runs under `-m gpu`)"""
    import test_dist_gpu as T
    T.test_eight_virtual_ranks_at_a_quarter_of_c4_match_the_single_gpu_layer(dev, "gat1")
