"""The hot-path GPU tests once more, in a child process, with every kernel launch of the library preceded by a NaN fill of the
whole LDS of every CU (include/cspn_hip.h: cspn_debug_set_lds_poison; the child gets CSPN_DEBUG_LDS_POISON=nan).

LDS is not cleared between kernels, so a kernel that reads an LDS word it never wrote is correct or not depending on what ran on
its CU before — invisible to a test that runs it in isolation (round 4's one-off mismatch of the default path at KITTI B = 8 with
sparse depth, root-caused in round 5: DESIGN.md §4.1b).  Under the fill such a read turns into NaN in the result, so the ordinary
assertions of the parity / bit-for-bit tests become assertions that no kernel depends on stale LDS.

Reference: the recurrence of network/libs/post_process/CSPN_new.py:80-92 and CSPN_ours.py:35-53 is deterministic."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["tests/test_hip_parity.py", "tests/test_hip_resident.py", "tests/test_hip_kres.py", "tests/test_hip_production.py",
         "tests/test_hip_backward.py"]


def test_hot_path_tests_pass_with_a_poisoned_lds():
    if os.environ.get("CSPN_DEBUG_LDS_POISON"):
        pytest.skip("this run is poisoned already")
    env = dict(os.environ, CSPN_DEBUG_LDS_POISON="nan", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + FILES,
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = out.stdout[-3000:]
    assert out.returncode == 0, tail
    assert " passed" in tail and " failed" not in tail, tail
