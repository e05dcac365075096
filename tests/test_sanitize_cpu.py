"""Host-side sanitizer pass: tools/sanitize_cpu.sh builds the library with the HOST code under
AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers do not exist on this pool) and runs the CPU boundary
tests -- argument validation, size queries, launch planning, error text -- against it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="sanitizer builds run on the CPU build host only, never in a process "
                                                         "that can see a GPU (the pool refuses sanitizer runs on GPU boxes)")
@pytest.mark.skipif(os.environ.get("NPI_GNN_LIB") is not None, reason="already running against a variant library")
@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_boundary_tests_are_clean_under_asan_and_ubsan():
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize_cpu.sh")], capture_output=True, text=True, timeout=1500)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "passed" in p.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
