"""GPU test of the bench.py contract the driver depends on: one JSON line with the agreed keys, exactly K timed
steps, roofline + cpu_baseline objects, and the same under a torch.distributed launch (world size 1)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _run(cmd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("CSPN_DEBUG_LDS_POISON", None)      # a poisoned-LDS run of the suite (every launch preceded by an LDS fill) must not reach the timed bench
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_single_process_json_line():
    d = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "12", "--warmup", "2", "--prewarm-s", "0.05"])
    assert REQUIRED <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "depth-maps/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 24 * 12 / (d["ms_per_step"] * 12 / 1e3)) / d["value"] < 1e-6
    r = d["roofline"]                       # the HBM-bound kernel: one launch per propagation step (north_star)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["steps_per_launch"] == 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["launches_timed"] % 24 == 0
    assert 0.5 < r["frac"] < 1.0                                                 # the >= 50 % of HBM roofline target
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-3 * r["achieved"]
    assert r["algorithmic_bytes_per_launch"] == 40 * 24 * 228 * 304
    assert "traffic" in r and "traffic_source" in r
    f = d["default_schedule"]               # the schedule `value` is measured on (temporal blocking)
    assert f["launches_timed"] == 12 * f["launches_per_forward"] and f["steps_per_launch"] >= 1
    assert "frac" not in f and f["compulsory_bytes_per_forward"] == 11 * 4 * 24 * 228 * 304
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert len(c["value_range"]) == 2 and c["value_range"][0] <= c["value"] <= c["value_range"][1]      # min-max over the best leg's repetitions
    assert c["range_over_legs"][0] <= c["range_over_legs"][1]
    sv = d["sparse_variant"]                # the variant the reference's call site runs (sparse_depth always passed), same process
    assert sv["value"] > 0 and sv["bytes_per_px_step"] == 48 and abs(sv["value"] - 24 / (sv["ms_per_step"] / 1e3)) < 1e-6 * sv["value"]
    assert sv["roofline"]["algorithmic_bytes_per_launch"] == 48 * 24 * 228 * 304 and 0.4 < sv["roofline"]["frac"] < 1.0
    assert sv["metrics_check"]["count"] > 0
    so = d["stock_ops_same_gpu"]            # the reference's own GPU path (stock ATen ops; the op-mix port) on this GPU, untimed region
    assert so["kind"] == "port" and so["batch"] == 24 and 0 < so["maps_per_s"] < d["value"]
    assert so["aten_ops_per_forward"] > 24 * 10 and so["max_rel_diff_hip_vs_stock_ops"] < 1e-4
    assert d["cache_cold"]["footprint_MB"] > 256
    assert d["metrics_check"]["count"] > 0


def test_torchrun_world_size_one():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
              "127.0.0.1", "--master-port", "29577", "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "1",
              "--prewarm-s", "0.05", "--no-cpu-baseline", "--no-train-leg", "--cold-sets", "0", "--workload", "kitti"])
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 6 and d["scaling"] == "strong"
