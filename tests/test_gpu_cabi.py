"""The boundary really is plain C: examples/c_abi_demo.cpp (no Python, no torch) is compiled against
include/npi_gnn.h, linked to the in-tree libnpi_gnn.so, and its SAGEConv forward is checked against its own CPU loop."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_from_a_cxx_host(dev, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    exe = str(tmp_path / "c_abi_demo")
    libdir = os.path.join(ROOT, "npi_gnn_amd")
    cmd = [hipcc, "-O2", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.cpp"),
           "-L", libdir, "-lnpi_gnn", f"-Wl,-rpath,{libdir}", "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, (run.stdout + run.stderr)[-2000:]
    assert "max |err|" in run.stdout
