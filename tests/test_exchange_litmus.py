"""GPU test: the cross-XCD message-passing litmus of the resident kernels' border exchange (tools/probes/exchange_litmus.hip), short form.

The exchange (cspn_resident.hip / cspnk_resident.hip / cspnk_d2.hip: sc1 payload stores -> s_waitcnt vmcnt(0) -> s_barrier -> relaxed flag
store || relaxed flag poll -> s_barrier -> sc1 payload loads) carries no fence; DESIGN.md §4.1b argues why that is enough on gfx950.  The
probe runs exactly those instruction sequences between workgroup pairs pinned to different XCDs under memory pressure: the product's
sequence must be clean, and the negative control (the vmcnt(0) dropped) must produce stale reads — otherwise the litmus would not be
sensitive to the ordering it is there to test.  The full run (10^9 handshakes) is profiles/r06_exchange_litmus.txt."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_exchange_sequences_pass_the_cross_xcd_litmus(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "exchange_litmus")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(ROOT, "tools", "probes", "exchange_litmus.hip")],
                          stderr=subprocess.DEVNULL)
    out = subprocess.run([exe, "20000", "1", "1", "1"], capture_output=True, text=True, timeout=300)
    text = out.stdout
    assert "== product sequence:" in text and "0 stale words, 0 time-outs -> clean" in text, text[-1500:]
    assert "64 of 64 pairs on two XCDs" in text, text[-1500:]
    assert "fails as it must" in text, text[-1500:]            # the negative control: the test is sensitive
    assert out.returncode == 0, text[-1500:]
