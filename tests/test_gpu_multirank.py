"""The N>1 code path of bench.py end to end on whatever GPUs the box has: two ranks over RCCL (`nccl`) when the box
shows at least two devices, else gloo with both ranks sharing device 0 — the criterion's num_masks all-reduce, the
gradient all-reduce (one flat buffer, and buckets started during backward), graph capture with a process group alive,
uneven ranks (a large graphed scene next to a small eager one), the instrumented roofline step on every rank."""
import json
import os
import subprocess
import sys

import pytest

from conftest import free_port as _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _backend():
    """RCCL whenever the box can give every rank its own device (reference: pl.Trainer(gpus=N) over NCCL,
    main_instance_segmentation.py:86-92); gloo on a one-GPU box, where the two ranks share the device."""
    import torch
    return "nccl" if torch.cuda.device_count() >= 2 else "gloo"


def _run(overlap, port, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", USC3D_OVERLAP_ALLREDUCE=overlap,
               **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--voxels", "40000", "--dist-backend", _backend(), "--no-cpu-baseline", "--rotate", "0"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_overlapped_gradient_allreduce_equals_the_single_one():
    """Buckets started during backward vs one all-reduce after it: the same sums, so the same weights and the same
    loss after four steps TO THE BIT — and the buckets of the backbone really do start early.  (Round 2 bounded the
    difference at 5e-6 after a handful of two-rank runs on a shared device had ended 5-6 ulp apart; the stress
    harness of tests/test_gpu_determinism.py — ~1 800 step sequences under load — never reproduced a differing bit,
    see DESIGN.md §6.)"""
    a, b = _run("1", _free_port()), _run("0", _free_port())
    la, lb = a["config"]["loss"], b["config"]["loss"]
    assert la == lb, (la, lb)
    note = a["config"]["grad_allreduce"]
    assert "buckets" in note and int(note.split(", ")[1].split()[0]) >= 1, note
    assert b["config"]["grad_allreduce"] == "one flat buffer after backward"


def test_second_streams_under_the_gradient_exchange():
    """The weight-gradient lane and the decoder's key-preparation stream write parameter gradients on streams of their
    own; a bucket's collective must start behind them (ddp.BucketedGradReducer._launch: lane join, side-stream join, and a
    report that arrives ON the side stream only counts).  With both streams on, overlapped buckets and one all-reduce
    after backward give the same loss to the bit.  (On a one-GPU box bench.py keeps the second streams off for ranks that
    share a device — two processes with four streams each on one device take seconds per step; forced on here.)"""
    on = {"USC3D_WGRAD_LANE_MAX_ROWS": str(1 << 40), "USC3D_KV_SIDE_STREAM": "1"}
    a, b = _run("1", _free_port(), **on), _run("0", _free_port(), **on)
    assert a["config"]["loss"] == b["config"]["loss"], (a["config"]["loss"], b["config"]["loss"])
    assert {r["role"] for r in a["config"]["streams"]} >= {"wgrad-lane", "keys"}, a["config"]["streams"]


def test_bench_two_ranks():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--voxels", "40000", "--dist-backend", _backend(), "--no-cpu-baseline", "--rotate", "0"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                 # rank 0 prints exactly one JSON line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["roofline"] is not None and rec["cpu_baseline"] is None
    cfg = rec["config"]
    assert cfg["dist_backend"] == _backend() and len(cfg["rank_devices"]) == 2
    assert cfg["rccl_ranks_seen"] == (2 if _backend() == "nccl" else 0)
    assert cfg["distinct_devices"] == (2 if _backend() == "nccl" else 1)
    assert cfg["weights_equal_across_ranks"] is True


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_uneven_ranks_neither_hang_nor_diverge(overlap):
    """Rank 0: a 150 k-voxel scene with captured decoder passes (device-bound step); rank 1: a 20 k-voxel scene run
    eagerly (host-bound step, other launch counts per parameter, levels smaller than the sampled key counts).  The
    collectives — num_masks, the learned gradient-write counts, the buckets started during backward on one rank and
    after it on the other — must line up (no hang: the subprocess timeout), and both ranks must end with the same
    weights to the bit and a finite loss."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", USC3D_OVERLAP_ALLREDUCE=overlap)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "2", "--voxels-by-rank", "150000,20000", "--eager-ranks", "1", "--dist-backend", _backend(),
           "--no-cpu-baseline", "--rotate", "0"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    cfg = rec["config"]
    assert rec["n_gpus"] == 2 and cfg["weights_equal_across_ranks"] is True
    assert cfg["loss"] == cfg["loss"] and abs(cfg["loss"]) < 1e4          # finite
    assert abs(cfg["voxels_per_scene"] - 150000) < 3000                     # rank 0's scene


def test_bench_gpus_2_without_a_launcher_spawns_two_ranks():
    """`python bench.py --gpus 2` as ONE process (no torchrun, no WORLD_SIZE): the script launches the two ranks itself
    and the line reports n_gpus == 2 (gloo here because the test box has one device; the default backend is RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--voxels", "40000", "--dist-backend", _backend(), "--no-cpu-baseline", "--rotate", "0"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 2 and rec["config"]["parallelism"] == "dp2"
    import torch
    if torch.cuda.device_count() >= 2:
        # a box with a device per rank: the collectives really spanned two RCCL ranks on two different devices
        c = rec["config"]
        assert c["dist_backend"] == "nccl" and c["rccl_ranks_seen"] == 2 and c["distinct_devices"] == 2, c
        assert c["weights_equal_across_ranks"] is True


def test_dry_collectives_mode_times_the_step_exchange_alone():
    """`bench.py --gpus 2 --dry-collectives`: the 13 num_masks all-reduces (reference models/criterion.py:258-260) and
    the bucket schedule of the gradient exchange without a training step around them — the figure to read beside
    ms_per_step on an N-GPU node."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--dist-backend", _backend(), "--dry-collectives"]
    env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["mode"] == "dry-collectives" and rec["n_gpus"] == 2
    assert 150 < rec["gradient_mb"] < 170 and len(rec["buckets"]) >= 4
    assert abs(sum(b["mb"] for b in rec["buckets"]) - rec["gradient_mb"]) < 1e-3
    assert all(b["blocking_ms"] > 0 for b in rec["buckets"]) and rec["one_flat_allreduce_ms"] > 0


def test_bench_refuses_rccl_ranks_without_devices():
    """nccl with more ranks than devices must fail loudly instead of reporting a shared-device number as N GPUs."""
    import torch
    n = torch.cuda.device_count() + 1
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--rotate", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "devices" in (out.stderr + out.stdout)


def _run_single(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--voxels", "40000",
           "--no-cpu-baseline", "--rotate", "0"] + extra
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_prefetched_batches_give_the_same_training_trajectory():
    """Voxelisation + coordinate maps of the next batch on a side stream (datasets/prefetch.py) vs built at the start
    of the step on the compute stream: same scene, same weights, so the loss after six steps must agree to the bit —
    a cross-stream lifetime bug would show up as a different or non-finite loss."""
    a, b = _run_single([]), _run_single(["--no-prefetch"])
    assert a["config"]["loss"] == b["config"]["loss"], (a["config"]["loss"], b["config"]["loss"])
    assert a["config"]["voxels_per_scene"] == b["config"]["voxels_per_scene"]


def test_one_rank_rccl_reducer_runs_the_real_exchange_and_changes_no_bit():
    """bench.py --force-dist: a ONE-rank RCCL process group on the one device this box has — the bucketed reducer (buckets
    started during backward as synchronous collectives on a second stream behind events of the compute stream and the
    lane) and the criterion's num_masks all-reduce, exactly as on N ranks minus the wire (reference: DDP's bucket
    all-reduce, main_instance_segmentation.py:86-92; models/criterion.py:258-260).  Averaging over one rank is the
    identity: the loss after the steps has the bits of the plain run.  The stream probe under an ASYNCHRONOUS all-reduce
    (streams.recheck_under_collective) is on request only — the step issues no asynchronous collective — and the line
    says what it measured when asked."""
    def run(*extra, **more_env):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), **more_env)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--voxels", "40000",
               "--no-cpu-baseline", "--rotate", "0", "--no-zorder", *extra]
        out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    plain, forced = run(), run("--force-dist")
    assert forced["n_gpus"] == 1 and forced["config"]["loss"] == plain["config"]["loss"], (forced["config"]["loss"], plain["config"]["loss"])
    note = forced["config"]["grad_allreduce"]
    assert note and "buckets" in note and plain["config"]["grad_allreduce"] is None, note
    assert "wgrad-lane under an all-reduce" not in {r["role"] for r in forced["config"]["streams"]}
    assert forced["config"]["rccl_ranks_seen"] == 1
    probed = run("--force-dist", USC3D_STREAM_RECHECK="1")
    assert "wgrad-lane under an all-reduce" in {r["role"] for r in probed["config"]["streams"]}
    assert probed["config"]["loss"] == plain["config"]["loss"]
