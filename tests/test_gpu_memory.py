"""A training step must not leave device memory to the cyclic garbage collector: round 4's soak run (tools/soak.py) showed
the allocator growing by 40 MB per step for 300 steps — coordinate manager -> precomputed geometry -> SparseTensor ->
coordinate manager kept every batch's maps and encodings alive until a generation-2 collection."""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_training_steps_free_their_batch_without_the_cycle_collector(device, monkeypatch):
    import bench

    # the next batch issued by the main thread: with the worker thread the two measurement points may or may not
    # include the batch being prefetched (one 20 k-voxel batch is ~20 MB: the test would be flaky, not the step)
    monkeypatch.setenv("USC3D_PREFETCH_THREAD", "0")
    args = bench.parse(["--no-cpu-baseline", "--voxels", "20000", "--rotate", "0"])
    step = bench.make_mask3d_step(args, device, 0, 1)
    try:
        for _ in range(3):
            step(1)
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        try:
            step(1)
            torch.cuda.synchronize()
            m0 = torch.cuda.memory_allocated()
            for _ in range(6):
                step(1)
            torch.cuda.synchronize()
            m1 = torch.cuda.memory_allocated()
        finally:
            gc.enable()
        assert m1 - m0 < 4 * 2**20, f"allocated memory grew by {(m1 - m0) / 2**20:.1f} MB over 6 steps without gc"
    finally:
        step.close()


def test_steady_state_preparation_keeps_the_step_bit_identical_and_out_of_the_driver_allocator(device, monkeypatch):
    """trainer.prepare_steady_state: the pools are pre-sized and the interpreter heap frozen between two steps — the
    trajectory must not change by a bit (same seeds, same scenes: compared with a run that never calls it), and the
    steps after it must not grow the reserved memory (no hipMalloc in the steady state)."""
    import bench
    from unscene3d_amd.trainer.trainer import prepare_steady_state

    monkeypatch.setenv("USC3D_PREFETCH_THREAD", "0")

    def run(prepare):
        torch.manual_seed(7)
        args = bench.parse(["--no-cpu-baseline", "--voxels", "20000", "--rotate", "0"])
        step = bench.make_mask3d_step(args, device, 0, 1)
        losses, info, reserved = [], None, None
        try:
            for k in range(6):
                if k == 2 and prepare:
                    info = prepare_steady_state(device)
                losses.append(step(1)[0])
                if k == 3:
                    torch.cuda.synchronize()
                    reserved = torch.cuda.memory_reserved()
            torch.cuda.synchronize()
            return torch.stack(losses).cpu(), info, reserved, torch.cuda.memory_reserved()
        finally:
            step.close()
            gc.unfreeze()

    with_prep, info, r3, r5 = run(True)
    without, _, _, _ = run(False)
    assert torch.equal(with_prep, without), (with_prep, without)
    assert info["streams"] >= 2 and info["frozen_objects"] > 1000
    assert info["reserved_after"] >= info["main_bytes"] + (info["streams"] - 1) * info["side_bytes"]
    assert r5 == r3, f"reserved memory grew from {r3} to {r5} bytes after the pools had been pre-sized"


def test_optimizer_ranges_inside_the_backward_pass_keep_the_trajectory_bits(device, monkeypatch):
    """optim.FlatAdamW.enable_early: the trunk's parameters are stepped stage by stage on the key-preparation stream while
    the backward pass is still being issued; the same kernel on the same values — losses and final weights of four steps
    equal those of the one-launch optimizer to the bit, and most of the parameters really did go early."""
    import bench

    monkeypatch.setenv("USC3D_PREFETCH_THREAD", "0")

    def run(early):
        monkeypatch.setenv("USC3D_EARLY_OPTIMIZER", "1" if early else "0")
        args = bench.parse(["--no-cpu-baseline", "--voxels", "20000", "--rotate", "0"])
        step = bench.make_mask3d_step(args, device, 0, 1)
        try:
            early_elems, orig = [], step.opt._launch

            def counted(lo, hi, step_no):
                if torch.cuda.current_stream().cuda_stream != torch.cuda.default_stream().cuda_stream:
                    early_elems.append(hi - lo)
                return orig(lo, hi, step_no)
            step.opt._launch = counted
            losses = [step(1)[0] for _ in range(4)]
            torch.cuda.synchronize()
            return torch.stack(losses).cpu(), step.opt.flat_param.clone().cpu(), sum(early_elems) / 4, step.opt.flat_param.numel()
        finally:
            step.close()

    la, wa, n_early, n_all = run(True)
    lb, wb, n_none, _ = run(False)
    assert n_none == 0 and n_early > 0.8 * n_all, (n_early, n_all)
    assert torch.equal(la, lb) and torch.equal(wa, wb)
