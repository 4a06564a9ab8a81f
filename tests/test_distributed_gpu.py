"""The N > 1 code path on the hardware at hand: two (and three) processes share the ONE visible
GPU with backend gloo — bench.py's oversubscription branch — so shard -> HIP forward_scored -> all-gather runs with
world_size > 1 and real kernels.  (The rccl/xGMI path proper needs an 8-GPU node, which only the driver has.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _torchrun(nproc, port, script_args, timeout=600):
    # several processes share the one GPU here: the weight-resident launches need the device to themselves (include/cspn_hip.h)
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="4", CSPN_RESIDENT="off")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port)] + script_args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    if out.returncode != 0:            # the whole story, not pytest's abbreviated repr of the assertion
        print("---- stdout ----\n%s\n---- stderr ----\n%s" % (out.stdout[-4000:], out.stderr[-12000:]))
    assert out.returncode == 0, "torchrun %s failed (rc %d): see the captured output" % (script_args[0], out.returncode)
    return out.stdout


@pytest.mark.parametrize("world,B", [(2, 8), (3, 8), (2, 1)])
def test_sharded_forward_equals_unsharded_bit_for_bit(world, B):
    """KITTI-shaped frames (config 4), full size for B=8 over 2 ranks; 3 ranks = uneven shards; B=1 = an empty shard."""
    H, W = (352, 1216) if world == 2 else (88, 304)
    out = _torchrun(world, 29611 + world + B, [os.path.join("tests", "dist_shard_worker.py"), "gloo", str(B), str(H), str(W), "24"])
    assert "SHARD_CHECK_OK world=%d" % world in out, out[-2000:]


def test_bench_two_ranks_on_one_gpu_kitti():
    """bench.py --gpus 2 --backend gloo --workload kitti: the driver's N>1 launch line, oversubscribing the one GPU.  The
    two ranks refine contiguous halves of ONE seeded batch, so the gathered metric sums equal the single-process run's."""
    args = ["--steps", "4", "--warmup", "1", "--prewarm-s", "0.05", "--no-cpu-baseline", "--no-train-leg", "--cold-sets", "0",
            "--no-per-step-leg", "--workload", "kitti"]
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + args, cwd=ROOT, capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, PYTHONPATH=ROOT))
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    out = _torchrun(2, 29655, ["bench.py", "--gpus", "2", "--backend", "gloo"] + args)
    d2 = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "strong" and d2["config"]["batch_per_gpu"] == 4
    assert d1["n_gpus"] == 1 and d1["config"]["batch_per_gpu"] == 8
    assert d2["value"] > 0 and abs(d2["value"] - 8 * 4 / (d2["ms_per_step"] * 4 / 1e3)) / d2["value"] < 1e-6
    m1, m2 = d1["metrics_check"], d2["metrics_check"]
    assert m1["count"] == m2["count"] and m1["count"] > 0
    # the refined depth is bit-identical between the schedules (tests/test_hip_resident.py); the fused metric sums are
    # fp32 per thread before they meet in fp64, and the single-process run (weight-resident plan) groups the pixels of a
    # thread differently from the shards' multi-launch plan: equal up to summation order
    for k in ("rmse", "absrel", "delta1"):
        assert abs(m1[k] - m2[k]) <= 1e-6 * abs(m1[k]), (k, m1[k], m2[k])


def test_ddp_training_step_two_ranks():
    """BASELINE config 5's data-parallel branch at world size 2 (VERDICT r2 missing #3): DistributedDataParallel +
    nn.SyncBatchNorm around the re-hosted unet_cspn_nyu with the HIP CSPN in forward and backward, two processes on the one
    GPU (gloo).  The worker asserts: DDP gradients = mean of the ranks' own gradients, SyncBN statistics and parameters
    identical across ranks after optimiser steps, the CSPN pair inside the model equals the oracle.  Replaces
    libs/trainers/multi_gpu_trainer.py:32-37.  (RCCL proper needs >= 2 GPUs: never run here — DESIGN.md §5.)"""
    out = _torchrun(2, 29671, [os.path.join("tests", "dist_ddp_worker.py"), "gloo", "2"], timeout=1500)
    assert "DDP_CHECK_OK world=2" in out, out[-2000:]


def test_bench_train_two_ranks_on_one_gpu():
    """bench.py --gpus 2 --backend gloo --workload train: the driver's launch line for config 5 through bench.py's own
    DDP + SyncBatchNorm branch (bench.py run_train), oversubscribing the one GPU."""
    out = _torchrun(2, 29683, ["bench.py", "--gpus", "2", "--backend", "gloo", "--workload", "train", "--steps", "2",
                               "--warmup", "1"], timeout=1500)
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 6
    assert "DDP x2" in d["config"]["parallelism"] and d["value"] > 0
    assert all(v == v and abs(v) < 1e6 for v in d["loss_first_last"])
    assert d["cspn_module"]["forward_us_p50"] > 0 and d["cspn_module"]["backward_us_p50"] > 0


def test_rccl_process_group_world_size_one():
    """The N > 1 code path of bench.py over RCCL itself (backend nccl), as far as one GPU allows: torchrun with ONE rank and
    CSPN_BENCH_FORCE_DIST=1 initialises the nccl process group with a device id, and the barrier, the all_reduce of the elapsed
    time and the metrics all-gather all run as collectives (world size 1) — RCCL loads, initialises and moves the 10 sums.
    (N > 1 over xGMI needs a multi-GPU node: only the driver has one — DESIGN.md §5.)"""
    env = dict(os.environ, PYTHONPATH=ROOT, CSPN_BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29691", "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "1", "--prewarm-s", "0.05",
           "--no-cpu-baseline", "--no-train-leg", "--no-per-step-leg", "--cold-sets", "0", "--workload", "kitti"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        print(out.stdout[-3000:], out.stderr[-8000:])
    assert out.returncode == 0
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["metrics_check"]["count"] > 0 and d["value"] > 0


def test_bench_train_whole_step_graph_matches_eager():
    """bench.py --workload train --graph on (one process): the whole optimiser step — stock convolutions, the un-pooling, both
    weight-resident CSPN launches with their control-word memsets, the fused SGD — captured as ONE HIP graph; after the same
    number of optimiser steps its loss equals the eager run's to the run-to-run noise of MIOpen's atomics."""
    res = {}
    for graph, warm in (("off", "9"), ("on", "3")):          # the graphed run adds 3 side-stream + 3 replayed warm-up steps
        cmd = [sys.executable, "bench.py", "--workload", "train", "--steps", "6", "--warmup", warm, "--no-cpu-baseline",
               "--graph", graph]
        out = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        res[graph] = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["on"]["config"]["hip_graph"] is True and res["off"]["config"]["hip_graph"] is False
    assert res["on"]["optimiser_steps_run"] == res["off"]["optimiser_steps_run"] == 15
    l_on, l_off = res["on"]["loss_first_last"], res["off"]["loss_first_last"]
    assert l_on[0] == l_off[0]
    assert l_off[1] < l_off[0] and abs(l_on[1] - l_off[1]) <= 5e-3 * l_off[1], (l_on, l_off)


def test_scale_sweep_script_dry_run_on_one_gpu(tmp_path):
    """tools/scale_sweep.sh is the one command that produces the north-star table on a multi-GPU node.  It
    cannot rot: here it runs N = 1, 2 over gloo, oversubscribing the one GPU, for the two inference workloads, and the table
    tool must print a line per (workload, N) with a whole-job rate and an efficiency."""
    env = dict(os.environ, PYTHONPATH=ROOT, GPUS="1 2", BACKEND="gloo", STEPS="4", WORKLOADS="nyu kitti", OUT=str(tmp_path), PORT="29720")
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_sweep.sh")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    assert "FAILED" not in out.stdout, out.stdout[-2000:]
    lines = [l.split() for l in out.stdout.splitlines() if l.split() and l.split()[0] in ("nyu", "kitti")]
    assert sorted((l[0], int(l[1])) for l in lines) == [("kitti", 1), ("kitti", 2), ("nyu", 1), ("nyu", 2)]
    assert all(float(l[2]) > 0 for l in lines)
    d2 = json.load(open(os.path.join(str(tmp_path), "scale_kitti_n2.json")))
    assert d2["n_gpus"] == 2 and d2["scaling"] == "strong"
