"""GPU smoke test of the training-step plumbing (BASELINE config 5 shape per GPU): stock PyTorch-ROCm network +
HIP CSPN forward/backward + SGD; the loss must go down and the native library must be the thing that ran."""
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_train_steps_reduce_loss():
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import train_ddp_smoke
    losses = train_ddp_smoke.main(["--steps", "25", "--batch", "2", "--H", "76", "--W", "100"])
    assert all(l == l for l in losses)                      # no NaN
    assert sum(losses[-5:]) / 5 < 0.6 * losses[0]
    assert "libcspn_hip.so" in open("/proc/self/maps").read()
