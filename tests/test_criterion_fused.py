"""The device criterion (csrc/criterion.hip: cost matrices, usc_lsap_batch, label / mask / dice losses and their
gradients in a handful of launches) against the torch-operator path of the same SetCriterion (which the golden vectors
of tests/test_golden_host.py pin to the reference, models/criterion.py + models/matcher.py) and against the golden
vectors directly."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(device, B, S_list, T_list, L=13, Q=100, C=3, seed=0, padded=True):
    from unscene3d_amd.models.criterion import SetCriterion
    from unscene3d_amd.models.matcher import HungarianMatcher
    g = torch.Generator().manual_seed(seed)
    matcher = HungarianMatcher(cost_class=2.0, cost_mask=5.0, cost_dice=2.0, cost_noise_robust=0.0, num_points=-1)
    wd = {"loss_ce": 2.0, "loss_mask": 5.0, "loss_dice": 2.0, "loss_noise_robust": 0.0}
    wd.update({f"{k}_{i}": v for i in range(L - 1) for k, v in list(wd.items())})
    crit = SetCriterion(num_classes=C, matcher=matcher, weight_dict=wd, eos_coef=0.1, losses=["labels", "masks"],
                        num_points=-1, oversample_ratio=3.0, importance_sample_ratio=0.75, class_weights=-1).to(device)
    logits = [(torch.randn(B, Q, C, generator=g) * 2).to(device).requires_grad_(True) for _ in range(L)]
    tables = [[(torch.randn(S, 128 if padded else Q, generator=g) * 3).to(device).requires_grad_(True) for S in S_list]
              for _ in range(L)]
    targets = []
    for S, T in zip(S_list, T_list):
        tm = torch.rand(T, S, generator=g) < 0.2
        tm[:, 0] = True
        labels = torch.ones(T, dtype=torch.int64)
        if T > 2:
            labels[1] = 0
        targets.append({"labels": labels.to(device), "segment_mask": tm.to(device)})
    return crit, logits, tables, targets, wd


def _outputs(logits, tables, Q, attach):
    def view(t):
        v = t[:, :Q]
        if attach:
            v._usc_padded = t
        return v
    levels = [{"pred_logits": lg, "pred_masks": [view(t) for t in tabs]} for lg, tabs in zip(logits, tables)]
    return {"pred_logits": levels[0]["pred_logits"], "pred_masks": levels[0]["pred_masks"], "aux_outputs": levels[1:]}


@pytest.mark.parametrize("B,S_list,T_list,padded", [(1, [609], [17], True), (2, [300, 1500], [5, 25], True),
                                                    (1, [64], [1], True), (1, [333], [32], False)])
def test_fused_criterion_equals_the_operator_path(device, monkeypatch, B, S_list, T_list, padded):
    import unscene3d_amd.models.criterion as CR
    crit, logits, tables, targets, wd = _case(device, B, S_list, T_list, padded=padded, seed=B + len(S_list))
    Q = 100

    def run(fused):
        monkeypatch.setattr(CR, "FUSED", fused)
        for t in logits + [x for tabs in tables for x in tabs]:
            t.grad = None
        losses = crit(_outputs(logits, tables, Q, attach=fused and padded), targets, mask_type="segment_mask")
        total = sum(v * wd[k] for k, v in losses.items())
        total.backward()
        return ({k: float(v) for k, v in losses.items()}, [lg.grad.clone() for lg in logits],
                [[x.grad.clone() for x in tabs] for tabs in tables])
    lf, glf, gtf = run(True)
    assert hasattr(crit, "last_indices")                      # the fused path really ran
    fused_idx = crit.last_indices
    lo, glo, gto = run(False)
    assert sorted(lf) == sorted(lo) and len(lf) == 52
    for k in lo:
        assert abs(lf[k] - lo[k]) <= 1e-5 * max(abs(lo[k]), 1e-3), (k, lf[k], lo[k])
    # the assignments: device solver == scipy on the operator path's cost matrices
    levels = [{"pred_logits": lg, "pred_masks": [t[:, :Q] for t in tabs]} for lg, tabs in zip(logits, tables)]
    ref_idx = crit.match_all_levels(levels, targets, "segment_mask")
    for l in range(len(levels)):
        for b in range(B):
            assert np.array_equal(fused_idx[l][b][0].cpu().numpy(), ref_idx[l][b][0].numpy()), (l, b)
            assert np.array_equal(fused_idx[l][b][1].cpu().numpy(), ref_idx[l][b][1].numpy()), (l, b)
    for a, b_ in zip(glf, glo):
        assert float((a - b_).norm()) <= 1e-4 * float(b_.norm()) + 1e-9
    for ta, tb in zip(gtf, gto):
        for a, b_ in zip(ta, tb):
            assert float((a - b_).norm()) <= 1e-4 * float(b_.norm()) + 1e-9
            if padded:
                assert float(a[:, Q:].abs().max()) == 0.0        # nothing leaks into the padding columns


def test_fused_criterion_matches_the_reference_golden(device):
    """tests/golden/criterion.npz (the reference's SetCriterion + HungarianMatcher run in the build container):
    losses 1e-5, matched indices exact, gradients 1e-4 — through the device criterion."""
    from test_golden_host import _criterion_case
    import unscene3d_amd.models.criterion as CR
    assert CR.FUSED
    z, crit, matcher, outputs, targets, wd, logits, masks = _criterion_case(str(device))
    losses = crit(outputs, targets, mask_type="segment_mask")
    assert getattr(losses, "flat", None) is not None and hasattr(crit, "last_indices")
    for k in losses:
        ref = float(z[f"loss/{k}"])
        assert abs(float(losses[k]) - ref) <= 1e-5 * max(abs(ref), 1e-3), (k, float(losses[k]), ref)
    for b in range(2):
        assert np.array_equal(crit.last_indices[0][b][0].cpu().numpy(), z[f"match_q_{b}"])
        assert np.array_equal(crit.last_indices[0][b][1].cpu().numpy(), z[f"match_t_{b}"])


@pytest.mark.parametrize("fault", ["nan_logit", "inf_mask", "label_out_of_range", "label_negative"])
def test_infeasible_assignment_raises_like_scipy(device, fault):
    """The reference stops a diverged run: scipy.optimize.linear_sum_assignment raises ValueError on NaN / inf costs
    (models/matcher.py:163).  The device solver reports status != 0 and the call itself carries on (identity
    assignment, no host wait); the status travels through pinned memory and the ValueError surfaces at the next
    criterion call or at check_lsap_status(wait=True).  An out-of-range target label takes the same route (it turns its
    cost column into NaN) instead of reading past the class logits."""
    crit, logits, tables, targets, wd = _case(device, 2, [300, 500], [5, 7], seed=9)
    with torch.no_grad():
        if fault == "nan_logit":
            logits[4][1, 17, 0] = float("nan")
        elif fault == "inf_mask":
            tables[2][0][0, 3] = float("inf")          # row 0 is in every target: inf - inf = NaN, as in the reference
        elif fault == "label_out_of_range":
            targets[1]["labels"][2] = 7
        else:
            targets[0]["labels"][0] = -1
    crit.check_lsap_status(wait=True)                                   # nothing pending
    losses = crit(_outputs(logits, tables, 100, attach=True), targets, mask_type="segment_mask")   # does not raise itself
    assert getattr(losses, "flat", None) is not None
    with pytest.raises(ValueError, match="invalid numeric entries"):
        crit.check_lsap_status(wait=True)
    crit.check_lsap_status(wait=True)                                   # reported once
    # a healthy call afterwards is clean, and a pending fault surfaces at the NEXT call without an explicit check
    good = _case(device, 2, [300, 500], [5, 7], seed=10)
    crit(_outputs(good[1], good[2], 100, attach=True), good[3], mask_type="segment_mask")
    crit.check_lsap_status(wait=True)
    crit(_outputs(logits, tables, 100, attach=True), targets, mask_type="segment_mask")
    torch.cuda.synchronize()
    with pytest.raises(ValueError):
        crit(_outputs(good[1], good[2], 100, attach=True), good[3], mask_type="segment_mask")


def test_host_labels_take_the_operator_path(device):
    """`labels` on the host (or of the wrong length) are not dereferenced by the kernels: the operator path runs."""
    crit, logits, tables, targets, wd = _case(device, 1, [200], [4], seed=3)
    targets[0]["labels"] = targets[0]["labels"].cpu()
    assert crit._fused_tables([{"pred_logits": logits[0], "pred_masks": [tables[0][0]]}], targets, "segment_mask") is None
