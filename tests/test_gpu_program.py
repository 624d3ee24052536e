"""The Res16UNet trunk as one step program each way (unscene3d_amd/program.py, csrc/units.hip: usc_program_run) against
the per-block native path (units.py) and the per-operator path: same kernels per unit, so the features agree to the
last bit; gradients to rounding (fan-in order and grouped weight gradients differ).
Reference walk: models/res16unet.py:224-297."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import sparse_ref as R

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def _scene(voxels, seed):
    from unscene3d_amd.synthetic import make_scene
    sc = make_scene(seed, target_voxels=voxels, tol=0.05)
    ec = R.voxel_floor(sc["xyz"], 0.02)
    eu, _ = R.sparse_quantize(ec)
    return R.sparse_collate([ec[eu]], [sc["colors"][eu]])


def _run(model, coords4, feats, device, ext_weights):
    from unscene3d_amd import MinkowskiEngine as ME
    x = ME.SparseTensor(features=torch.from_numpy(feats).to(device), coordinates=torch.from_numpy(coords4).to(device),
                        device=device)
    out, fmaps = model(x)
    # every level output gets its own external gradient (the decoder's use of the aux levels), the last one twice
    loss = sum(w * f.F.square().mean() for w, f in zip(ext_weights, fmaps)) + out.F.abs().mean()
    return out, fmaps, loss


@pytest.mark.parametrize("arch,voxels", [("Res16UNet34C", 40_000), ("Res16UNet14", 6_000), ("Res16UNet34C", 3_000)])
def test_step_program_equals_the_per_block_path(device, monkeypatch, arch, voxels):
    from unscene3d_amd import program, units
    from unscene3d_amd.models import res16unet

    coords4, feats = _scene(voxels, 2200 + voxels % 97)
    torch.manual_seed(5)
    cfg = SimpleNamespace(bn_momentum=0.02, conv1_kernel_size=3, dilations=[1, 1, 1, 1])
    model = getattr(res16unet, arch)(3, 20, cfg, out_fpn=True).to(device).train()
    assert program.plan_of(model) is not None
    state = {k: v.clone() for k, v in model.state_dict().items()}
    ran = []
    real = program._Trunk.apply
    monkeypatch.setattr(program._Trunk, "apply", staticmethod(lambda *a: (ran.append(1), real(*a))[1]))
    res = {}
    for mode in ("blocks", "program", "program-again"):
        monkeypatch.setattr(program, "ENABLED", mode != "blocks")
        monkeypatch.setattr(units, "GROUP_WGRAD", mode != "blocks")
        model.load_state_dict(state)
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        n0 = len(ran)
        out, fmaps, loss = _run(model, coords4, feats, device, (0.5, 1.0, 1.5, 2.0, 0.25))
        loss.backward()
        assert (len(ran) > n0) == (mode != "blocks")
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if not n.startswith("final.")}
        stats = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
        res[mode] = (out.F.detach().clone(), [f.F.detach().clone() for f in fmaps], grads, stats, float(loss))
    a, b, c = res["program"], res["blocks"], res["program-again"]
    assert torch.equal(a[0], b[0])
    for fa, fb in zip(a[1], b[1]):
        assert torch.equal(fa, fb)                                          # all five level outputs, bit for bit
    for k in b[3]:
        assert torch.equal(a[3][k], b[3][k]), k                             # running statistics, batch counters
    worst = max((rel_err(a[2][n], b[2][n]), n) for n in b[2])
    assert worst[0] < 2e-5, worst
    for n in a[2]:
        assert torch.equal(a[2][n], c[2][n]), n                             # deterministic run to run


def test_step_program_in_eval_mode_and_fallbacks(device, monkeypatch):
    """eval() + no_grad runs the program on the running statistics (no backward state); parameters without an
    allocated gradient buffer, or a switched-off program, take the per-block path — same numbers."""
    from unscene3d_amd import program
    from unscene3d_amd.models.res16unet import Res16UNet34C

    coords4, feats = _scene(12_000, 2301)
    torch.manual_seed(6)
    cfg = SimpleNamespace(bn_momentum=0.02, conv1_kernel_size=3, dilations=[1, 1, 1, 1])
    model = Res16UNet34C(3, 20, cfg, out_fpn=True).to(device).train()
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    _run(model, coords4, feats, device, (1, 1, 1, 1, 1))[2].backward()      # moves the running statistics
    model.eval()
    with torch.no_grad():
        monkeypatch.setattr(program, "ENABLED", True)
        a = _run(model, coords4, feats, device, (1, 1, 1, 1, 1))
        monkeypatch.setattr(program, "ENABLED", False)
        b = _run(model, coords4, feats, device, (1, 1, 1, 1, 1))
    for fa, fb in zip(a[1], b[1]):
        assert torch.equal(fa.F, fb.F)
    # no gradient buffers: autograd must receive the gradients -> per-block path, results as before
    model.train()
    monkeypatch.setattr(program, "ENABLED", True)
    model.zero_grad(set_to_none=True)
    from unscene3d_amd import MinkowskiEngine as ME
    x = ME.SparseTensor(features=torch.from_numpy(feats).to(device), coordinates=torch.from_numpy(coords4).to(device),
                        device=device)
    assert program.usable(model, x) is None
    out, fmaps, loss = _run(model, coords4, feats, device, (1, 1, 1, 1, 1))
    loss.backward()
    assert model.block4[0].conv1.kernel.grad is not None and bool(torch.isfinite(model.block4[0].conv1.kernel.grad).all())


def _fresh_model(device, seed=7):
    from unscene3d_amd.models.res16unet import Res16UNet34C
    torch.manual_seed(seed)
    cfg = SimpleNamespace(bn_momentum=0.02, conv1_kernel_size=3, dilations=[1, 1, 1, 1])
    model = Res16UNet34C(3, 20, cfg, out_fpn=True).to(device).train()
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    return model


def test_lane_schedule_holds_the_fine_weight_gradients_and_changes_no_bit(device, monkeypatch):
    """usc_wgrad_lane_hold (round 6): the fine decoder stages' weight gradients are noted and only put on the lane when
    the backward walk reaches a coarse map.  Same kernels on the same operands: every gradient is bit-equal to the
    un-held schedule, nothing stays noted after the pass, wherever the release point lies."""
    from unscene3d_amd import program, units
    from unscene3d_amd._lib import lib

    coords4, feats = _scene(30_000, 2317)
    model = _fresh_model(device)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    monkeypatch.setattr(program, "ENABLED", True)
    assert units._lane(torch.device(device)) is not None, "the weight-gradient lane is on by default"
    res = {}
    coarsest = int(np.unique(coords4[:, 1:] // 16, axis=0).shape[0])       # rows of the stride-16 map
    assert 0 < coarsest < 5_000
    for name, hold_min, release in (("plain", 1 << 40, 0), ("held", 20_000, 5_000), ("coarsest-only", 20_000, coarsest)):
        monkeypatch.setattr(units, "LANE_HOLD_MIN_ROWS", hold_min)
        monkeypatch.setattr(units, "LANE_RELEASE_ROWS", release)
        model.load_state_dict(state)
        for p in model.parameters():
            p.grad.zero_()
        held_seen = []
        real_run = lib.usc_program_run

        def run(*a, _real=real_run, _seen=held_seen):
            rc = _real(*a)
            _seen.append(int(lib.usc_wgrad_lane_holding()))
            return rc
        monkeypatch.setattr(lib, "usc_program_run", run)
        _run(model, coords4, feats, device, (0.5, 1.0, 1.5, 2.0, 0.25))[2].backward()
        monkeypatch.setattr(lib, "usc_program_run", real_run)
        torch.cuda.synchronize()
        assert lib.usc_wgrad_lane_holding() == 0
        if name == "held":      # the forward call, then: held over the first (fine) stage calls, released by a coarse one
            assert held_seen[1] == 1 and held_seen[-1] == 0, held_seen
        res[name] = {n: p.grad.clone() for n, p in model.named_parameters() if not n.startswith("final.")}
    for name in ("held", "coarsest-only"):
        for n, g in res["plain"].items():
            assert torch.equal(g, res[name][n]), (name, n)
    assert float(res["plain"]["block8.0.conv1.kernel"].abs().sum()) > 0


def test_joins_are_queued_again_after_a_failed_backward(device, monkeypatch):
    """A backward node that raises makes the engine drop its end-of-backward callbacks.  The queued-once marks of the
    lane join / grouped weight gradients are keyed on the graph task, so the next pass queues its own join and its
    gradients are complete (a sticky flag left every later pass un-joined)."""
    from unscene3d_amd import program, units

    class Poison(torch.autograd.Function):
        @staticmethod
        def forward(ctx, w):
            return w * 1.0

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("poisoned backward node")

    coords4, feats = _scene(12_000, 2323)
    model = _fresh_model(device, seed=8)
    monkeypatch.setattr(program, "ENABLED", True)
    joins = []
    real_join = units.join_lane
    monkeypatch.setattr(units, "join_lane", lambda *a, **k: (joins.append(1), real_join(*a, **k))[1])
    _run(model, coords4, feats, device, (1, 1, 1, 1, 1))[2].backward()
    torch.cuda.synchronize()
    assert joins, "the lane join runs at the end of a backward pass"
    want = {n: p.grad.clone() for n, p in model.named_parameters() if not n.startswith("final.")}
    for p in model.parameters():
        p.grad.zero_()
    # created BEFORE the forward pass -> lower sequence number -> its backward runs after the trunk's
    w = torch.ones(4, device=device, requires_grad=True)
    z = Poison.apply(w)
    loss = _run(model, coords4, feats, device, (1, 1, 1, 1, 1))[2] + z.sum()
    with pytest.raises(RuntimeError, match="poisoned"):
        loss.backward()
    torch.cuda.synchronize()
    for p in model.parameters():
        p.grad.zero_()
    model.load_state_dict({k: v for k, v in model.state_dict().items()})
    n0 = len(joins)
    # the batch-norm running statistics moved twice; the gradients depend on the batch statistics only
    _run(model, coords4, feats, device, (1, 1, 1, 1, 1))[2].backward()
    assert len(joins) > n0, "the pass after a failed one queued no lane join"
    torch.cuda.synchronize()
    for n, g in want.items():
        assert torch.equal(g, dict(model.named_parameters())[n].grad), n
