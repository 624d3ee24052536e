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
