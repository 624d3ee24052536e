"""CPU, world_size 2, gloo: the data-parallel pieces of the step that do not need a GPU —
the flat gradient buffer all-reduce (unscene3d_amd/ddp.py) and the criterion's num_masks all-reduce
(reference models/criterion.py:258-260)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from unscene3d_amd.ddp import all_reduce_mean_, flatten_grads
        from unscene3d_amd.models.criterion import SetCriterion
        from unscene3d_amd.models.matcher import HungarianMatcher

        # 1. flat gradient buffer: p.grad are views; one all-reduce averages every parameter
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
        params = list(model.parameters())
        flat = flatten_grads(params)
        x = torch.full((5, 8), float(rank + 1))
        model(x).sum().backward()
        assert all(p.grad.data_ptr() >= flat.data_ptr() for p in params)      # still views after backward
        local = flat.clone()
        all_reduce_mean_(flat, world)
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        assert torch.allclose(flat, sum(gathered) / world)
        assert torch.allclose(params[0].grad.reshape(-1), flat[:params[0].numel()])

        # 1b. bucketed reducer: learns the per-parameter report counts on the first step, then starts buckets during
        #     backward (here through autograd's post-accumulate hooks); result == mean of the local gradients
        from unscene3d_amd.ddp import BucketedGradReducer
        torch.manual_seed(1)
        deep = torch.nn.Sequential(*[torch.nn.Linear(12, 12) for _ in range(6)])
        import copy
        twin = copy.deepcopy(deep)                   # same weights, no reducer: the local gradients to compare with
        tflat = flatten_grads(list(twin.parameters()))
        dparams = list(deep.parameters())
        dflat = flatten_grads(dparams)
        red = BucketedGradReducer(dparams, dflat, world, bucket_bytes=1200)
        assert len(red.bounds) >= 3
        for it in range(3):
            deep.zero_grad(set_to_none=False)
            red.begin_step()
            xin = torch.full((4, 12), float(rank + 1 + it))
            twin.zero_grad(set_to_none=False)
            twin(xin).square().sum().backward()
            local = tflat.clone()
            deep(xin).square().sum().backward()
            early = sum(red.launched)
            red.finish()
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            assert torch.equal(dflat, (gathered[0] + gathered[1]) / world), it
            assert (early == 0) if it == 0 else (early >= len(red.bounds) - 1), (it, early)
            assert all(p.grad.data_ptr() >= dflat.data_ptr() for p in dparams)

        # 2. criterion: num_masks is summed over ranks and divided by the world size
        g = torch.Generator().manual_seed(3)
        T = 2 + 3 * rank                      # 2 targets on rank 0, 5 on rank 1 -> global mean 3.5
        S, Q = 40, 10
        outputs = {"pred_logits": torch.randn(1, Q, 3, generator=g),
                   "pred_masks": [torch.randn(S, Q, generator=g)], "aux_outputs": []}
        seg = torch.rand(T, S, generator=g) < 0.3
        seg[:, 0] = True
        targets = [{"labels": torch.ones(T, dtype=torch.int64), "segment_mask": seg}]
        wd = {"loss_ce": 2.0, "loss_mask": 5.0, "loss_dice": 2.0, "loss_noise_robust": 0.0}
        crit = SetCriterion(3, HungarianMatcher(2.0, 5.0, 2.0, 0.0, -1), wd, 0.1, ["labels", "masks"], -1, 3.0,
                            0.75, -1)
        losses = crit(outputs, targets, "segment_mask")
        out[rank] = {k: float(v) for k, v in losses.items()}
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == {0, 1}
    for r in res.values():
        assert all(torch.isfinite(torch.tensor(v)) for v in r.values())
