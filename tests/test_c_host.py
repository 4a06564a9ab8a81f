"""The C ABI used from a plain C++ host (examples/c_host_demo.cpp): no PyTorch in the process, only hipMalloc + the
entry points of include/cspn_hip.h.  Built with g++ against libcspn_hip.so and libamdhip64, run on the GPU box, checked
against the oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from cspn_monodepth_amd import _lib
from conftest import ROOT, rel_err
from oracle import cspn_oracle as orc

ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(tmp_path):
    exe = str(tmp_path / "c_host_demo")
    cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROCM, "include"), os.path.join(ROOT, "examples", "c_host_demo.cpp"),
           "-L", os.path.dirname(_lib.SO_PATH), "-lcspn_hip", "-L", os.path.join(ROCM, "lib"), "-lamdhip64",
           "-Wl,-rpath," + os.path.dirname(_lib.SO_PATH), "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=300)
    return exe


def test_c_host_demo_compiles_and_links(tmp_path):
    """CPU: the header is plain C-compatible and every symbol the demo uses resolves against the library."""
    _lib.build()
    exe = _build(tmp_path)
    assert os.path.exists(exe)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 1 and "usage" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("shape,sparse", [((2, 12, 40, 64), True), ((1, 8, 33, 52), False), ((1, 8, 9, 13), True),
                                          ((3, 12, 228, 304), True)])
def test_c_host_demo_matches_oracle(tmp_path, shape, sparse):
    B, C, H, W = shape
    T = 24
    g, d, s = orc.synthetic_inputs(seed=77, B=B, H=H, W=W, C=C, sparse_samples=60 if sparse else None)
    exe = _build(tmp_path)
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        for a in (g, d) + ((s,) if sparse else ()):
            f.write(np.ascontiguousarray(a, np.float32).tobytes())
    run = subprocess.run([exe, inp, outp, str(B), str(C), str(H), str(W), str(T), "1" if sparse else "0"],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr + run.stdout
    got = np.fromfile(outp, np.float32).reshape(B, 1, H, W)
    want = orc.cspn3_forward(g, d, s if sparse else None, T)
    assert rel_err(got, want) <= 1e-5
    if W % 4 == 0:
        assert "one-call == two-call" in run.stdout
        assert "resident == multi-launch bit for bit" in run.stdout, run.stdout       # the default schedule, from plain C++
