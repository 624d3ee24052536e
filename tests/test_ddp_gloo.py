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

        # 1c. ranks whose report counts differ (data-dependent eager/graphed passes): the divergent bucket is not
        #     eligible on ANY rank, so both issue the same collectives; result still the mean
        torch.manual_seed(2)
        net = torch.nn.Sequential(*[torch.nn.Linear(12, 12) for _ in range(4)])
        nparams = list(net.parameters())
        nflat = flatten_grads(nparams)
        ntwin = copy.deepcopy(net)                   # local gradients without a reducer (buckets reduce in place, early)
        ntflat = flatten_grads(list(ntwin.parameters()))
        red2 = BucketedGradReducer(nparams, nflat, world, bucket_bytes=600)
        nb = len(red2.bounds)
        for it in range(3):
            net.zero_grad(set_to_none=False)
            red2.begin_step()
            ntwin.zero_grad(set_to_none=False)
            ntwin(torch.full((4, 12), float(rank + 1))).sum().backward()
            local = ntflat.clone()
            net(torch.full((4, 12), float(rank + 1))).sum().backward()
            if rank == 1:
                red2.on_grad(nparams[-1])          # rank 1 reports the last parameter twice per step
            red2.finish()
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            assert torch.equal(nflat, (gathered[0] + gathered[1]) / world), it
        last_bucket = red2.bucket_of[id(nparams[-1])]
        assert last_bucket not in red2.order and len(red2.order) == nb - 1
        got = [None, None]
        dist.all_gather_object(got, red2.order)
        assert got[0] == got[1]

        # 1d. a write after its bucket's all-reduce started: raised on EVERY rank at the next step, not just the culprit
        for h in red2._hooks:
            h.remove()
        red3 = BucketedGradReducer(nparams, nflat, world, bucket_bytes=600)
        raised = False
        try:
            for it in range(3):
                net.zero_grad(set_to_none=False)
                red3.begin_step()
                net(torch.ones(4, 12)).sum().backward()
                if it == 1 and rank == 0:
                    red3.on_grad(nparams[-1])      # late report on rank 0 only, bucket already in flight
                red3.finish()
        except RuntimeError as err:
            raised = "after its bucket's all-reduce had started" in str(err) and it == 2
        assert raised, rank
        for h in red3._hooks:
            h.remove()

        # 1e. flush(): the flag of the LAST step is looked at too (finish() alone reads it one step later), and the
        #     message names the step
        red4 = BucketedGradReducer(nparams, nflat, world, bucket_bytes=600)
        for it in range(2):
            net.zero_grad(set_to_none=False)
            red4.begin_step()
            net(torch.ones(4, 12)).sum().backward()
            if it == 1 and rank == 1:
                red4.on_grad(nparams[-1])
            red4.finish()
        try:
            red4.flush()
            flushed = False
        except RuntimeError as err:
            flushed = "in step 2 of this reducer" in str(err)
        assert flushed, rank
        red4.flush()                                   # nothing pending any more
        for h in red4._hooks:
            h.remove()

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


def _worker8(rank, world, port, out):
    """World size 8 (the node the metric is quoted on): ranks report DIFFERENT numbers of gradient writes for some
    parameters (a rank whose scene is small runs a decoder pass eagerly, another replays a graph), the reducer must
    still issue the same collectives in the same order everywhere and return the mean of the eight local gradients."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import copy

        from unscene3d_amd.ddp import BucketedGradReducer, flatten_grads
        torch.manual_seed(5)
        net = torch.nn.Sequential(*[torch.nn.Linear(10, 10) for _ in range(8)])
        params = list(net.parameters())
        flat = flatten_grads(params)
        twin = copy.deepcopy(net)
        tflat = flatten_grads(list(twin.parameters()))
        red = BucketedGradReducer(params, flat, world, bucket_bytes=300)
        nb = len(red.bounds)
        assert nb >= 6
        extra = {1: [params[-1]], 5: [params[-1], params[0]], 6: [params[6]] * 2}.get(rank, [])
        for it in range(4):
            net.zero_grad(set_to_none=False)
            twin.zero_grad(set_to_none=False)
            red.begin_step()
            xin = torch.full((3, 10), 0.1 * (rank + 1) + 0.01 * it)
            twin(xin).square().sum().backward()
            local = tflat.clone()
            net(xin).square().sum().backward()
            for p in extra:
                red.on_grad(p)                     # additional reports: counts differ between the ranks
            red.finish()
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            want = gathered[0].clone()
            for g in gathered[1:]:
                want += g
            assert torch.allclose(flat, want / world, rtol=1e-6, atol=1e-9), (rank, it)
        red.flush()
        divergent = {red.bucket_of[id(p)] for p in (params[-1], params[0], params[6])}
        assert not (divergent & set(red.order))            # not eligible on ANY rank
        orders = [None] * world
        dist.all_gather_object(orders, (red.order, red.expected))
        assert all(o == orders[0] for o in orders)         # same plan everywhere
        out[rank] = (len(red.order), nb)
    finally:
        dist.destroy_process_group()


def test_world_size_8_gloo_bucketed_reducer_with_differing_report_counts():
    world = 8
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker8, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == set(range(8))
    n_order, nb = res[0]
    assert 0 < n_order < nb                                 # some buckets start early, the divergent ones do not
