"""oracle/criterion_ref.py (the checker behind the "oracle step" of tests/test_gpu_step_parity.py) pinned by the
reference's own matcher + criterion: tests/golden/criterion.npz was recorded by importing models/matcher.py and
models/criterion.py in the build container (tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

from oracle import criterion_ref as CR

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "criterion.npz")
WD = {"loss_ce": 2.0, "loss_mask": 5.0, "loss_dice": 2.0, "loss_noise_robust": 0.0}


def _case():
    z = np.load(GOLD)
    n_aux, B = int(z["n_aux"]), 2
    logits = [torch.from_numpy(z[f"logits_{i}"]).requires_grad_() for i in range(n_aux + 1)]
    masks = [[torch.from_numpy(z[f"masks_{i}_{b}"]).requires_grad_() for b in range(B)] for i in range(n_aux + 1)]
    targets = []
    for b in range(B):
        T, S = z[f"tgt_shape_{b}"]
        seg = torch.from_numpy(np.unpackbits(z[f"tgt_mask_{b}"], axis=1)[:, :S].astype(bool))
        targets.append({"labels": torch.ones(int(T), dtype=torch.int64), "segment_mask": seg})
    outputs = {"pred_logits": logits[-1], "pred_masks": masks[-1],
               "aux_outputs": [{"pred_logits": logits[i], "pred_masks": masks[i]} for i in range(n_aux)]}
    return z, outputs, targets, logits, masks


def test_oracle_assignment_matches_reference():
    z, outputs, targets, *_ = _case()
    idx = CR.hungarian_match({k: v for k, v in outputs.items() if k != "aux_outputs"}, targets, "segment_mask")
    for b in range(2):
        assert np.array_equal(idx[b][0].numpy(), z[f"match_q_{b}"])
        assert np.array_equal(idx[b][1].numpy(), z[f"match_t_{b}"])


def test_oracle_losses_and_gradients_match_reference():
    z, outputs, targets, logits, masks = _case()
    info = {}
    losses = CR.set_criterion(outputs, targets, "segment_mask", num_classes=3, eos_coef=0.1, info=info)
    ref_keys = sorted(k[5:] for k in z.files if k.startswith("loss/"))
    assert sorted(losses) == ref_keys
    for k in ref_keys:
        np.testing.assert_allclose(losses[k].detach().numpy(), z["loss/" + k], rtol=1e-5, atol=1e-7)
    wd = dict(WD)
    wd.update({f"{k}_{i}": v for i in range(int(z["n_aux"])) for k, v in WD.items()})
    total = sum(losses[k] * wd[k] for k in losses)
    np.testing.assert_allclose(total.detach().numpy(), z["total"], rtol=1e-5)
    total.backward()
    for i in range(len(logits)):
        np.testing.assert_allclose(logits[i].grad.numpy(), z[f"logits_grad_{i}"], rtol=1e-4, atol=1e-7)
        for b in range(2):
            np.testing.assert_allclose(masks[i][b].grad.numpy(), z[f"masks_grad_{i}_{b}"], rtol=1e-4, atol=1e-8)
    # forced assignments reproduce the same numbers, and `info` carried the oracle's own
    again = CR.set_criterion(outputs, targets, "segment_mask", forced_indices=info["indices"])
    for k in ref_keys:
        assert float(again[k]) == float(losses[k])


def test_oracle_is_independent_of_the_package():
    src = open(CR.__file__).read()
    assert "unscene3d_amd" not in src.split('"""', 2)[2]         # outside the header docstring
