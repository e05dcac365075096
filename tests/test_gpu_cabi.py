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


@pytest.mark.gpu
def test_the_library_this_box_runs_is_the_guarded_build():
    """the ISA guard (tests/test_isa_guard.py) runs where the library is BUILT.  On the GPU box nothing is
    rebuilt -- the prebuilt .so travels with its objects -- so the same guard is run here on those very objects (the box has the
    same llvm tools), and the .so must be the one linked from them: not stale against any source, newer than every object."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import isa_guard as G
    from npi_gnn_amd.build import HERE, HOST_ONLY, LIB, SOURCES, _stale
    assert os.path.exists(LIB) and not _stale(), "the shipped library is older than its sources: it would be rebuilt on this box"
    t_lib = os.path.getmtime(LIB)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        for src in SOURCES:
            obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
            assert os.path.exists(obj) and os.path.getmtime(obj) <= t_lib + 1.0, obj
            if src in HOST_ONLY:                              # sequences of the other entry points: no device code to guard
                continue
            meta, code = G.analyse(obj, tmp)
            for name, m in meta.items():
                assert m["scratch"] == 0 and m["vgpr_spill"] == 0, (src, name, m)
            for sym, instrs in code.items():
                assert G.find_flat(instrs) == [], (src, sym)
                assert G.find_sgpr_hazards(instrs) == [], (src, sym)
                assert G.find_inflight_touch_linear(instrs) == [], (src, sym)


@pytest.mark.gpu
def test_the_ctypes_stub_of_integration_md_runs_as_printed():
    """INTEGRATION.md section 2 shows the binding a maintainer would add to the reference: that very code block, cut out of the
    document and executed (library path made absolute), against the oracle's SAGEConv -- so the document cannot drift from the ABI
    (ABI 4 added an optional pointer to two of the three calls it makes)."""
    import re
    import torch
    from oracle import ref_conv as R
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    stub = [b for b in blocks if "def sage_forward(" in b and "ctypes.CDLL" in b]
    assert len(stub) == 1
    code = stub[0].replace('"npi_gnn_amd/libnpi_gnn.so"', repr(os.path.join(root, "npi_gnn_amd", "libnpi_gnn.so")))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    N, E, Fi, Fo = 2500, 30000, 178, 128
    ei = torch.randint(0, N, (2, E), generator=g)
    x = torch.randn(N, Fi, generator=g)
    W = (torch.rand(Fi, Fo, generator=g) * 2 - 1) / Fi ** 0.5
    b = (torch.rand(Fo, generator=g) * 2 - 1) / Fi ** 0.5
    out = ns["sage_forward"](x.to(dev), ei.to(dev), W.to(dev), b.to(dev))
    torch.cuda.synchronize()
    ref = R.sage_conv(x, ei, W, b)
    assert float((out.cpu() - ref).abs().max()) <= 1e-4
