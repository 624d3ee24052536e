"""CPU: host-side mirrors of the reference interface (matcher, criterion, decoder layers, GenericMLP)
against golden vectors produced by importing the reference itself (tests/golden/make_golden.py).
These modules are plain torch (they stay on PyTorch-ROCm on the GPU box), so the check runs anywhere."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name))


def _criterion_case():
    from unscene3d_amd.models.criterion import SetCriterion
    from unscene3d_amd.models.matcher import HungarianMatcher

    z = _load("criterion.npz")
    n_aux, B = int(z["n_aux"]), 2
    logits = [torch.from_numpy(z[f"logits_{i}"]).requires_grad_() for i in range(n_aux + 1)]
    masks = [[torch.from_numpy(z[f"masks_{i}_{b}"]).requires_grad_() for b in range(B)] for i in range(n_aux + 1)]
    targets = []
    for b in range(B):
        T, S = z[f"tgt_shape_{b}"]
        seg = torch.from_numpy(np.unpackbits(z[f"tgt_mask_{b}"], axis=1)[:, :S].astype(bool))
        targets.append({"labels": torch.ones(int(T), dtype=torch.int64), "segment_mask": seg})
    matcher = HungarianMatcher(cost_class=2.0, cost_mask=5.0, cost_dice=2.0, cost_noise_robust=0.0, num_points=-1)
    wd = {"loss_ce": 2.0, "loss_mask": 5.0, "loss_dice": 2.0, "loss_noise_robust": 0.0}
    wd.update({f"{k}_{i}": v for i in range(n_aux) for k, v in list(wd.items())})
    crit = SetCriterion(num_classes=3, matcher=matcher, weight_dict=wd, eos_coef=0.1, losses=["labels", "masks"],
                        num_points=-1, oversample_ratio=3.0, importance_sample_ratio=0.75, class_weights=-1)
    outputs = {"pred_logits": logits[-1], "pred_masks": masks[-1],
               "aux_outputs": [{"pred_logits": logits[i], "pred_masks": masks[i]} for i in range(n_aux)]}
    return z, crit, matcher, outputs, targets, wd, logits, masks


def test_matcher_assignment_matches_reference():
    z, crit, matcher, outputs, targets, *_ = _criterion_case()
    idx = matcher(outputs, targets, "segment_mask")
    for b in range(2):
        assert np.array_equal(idx[b][0].numpy(), z[f"match_q_{b}"])       # integer assignment: exact
        assert np.array_equal(idx[b][1].numpy(), z[f"match_t_{b}"])


def test_criterion_losses_and_grads_match_reference():
    z, crit, matcher, outputs, targets, wd, logits, masks = _criterion_case()
    losses = crit(outputs, targets, mask_type="segment_mask")
    ref_keys = sorted(k[5:] for k in z.files if k.startswith("loss/"))
    assert sorted(losses) == ref_keys                                     # 4 scalars x (1 + n_aux) levels
    for k in ref_keys:
        np.testing.assert_allclose(losses[k].detach().numpy(), z["loss/" + k], rtol=1e-5, atol=1e-7)
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    np.testing.assert_allclose(total.detach().numpy(), z["total"], rtol=1e-5)
    total.backward()
    for i in range(len(logits)):
        np.testing.assert_allclose(logits[i].grad.numpy(), z[f"logits_grad_{i}"], rtol=1e-4, atol=1e-7)
        for b in range(2):
            np.testing.assert_allclose(masks[i][b].grad.numpy(), z[f"masks_grad_{i}_{b}"], rtol=1e-4, atol=1e-8)


def test_decoder_layers_match_reference():
    from unscene3d_amd.models.mask3d import CrossAttentionLayer, FFNLayer, SelfAttentionLayer
    from unscene3d_amd.models.modules.helpers_3detr import GenericMLP

    z = _load("decoder_layers.npz")
    d, H = 128, 8
    ca, sa, ffn = CrossAttentionLayer(d, H), SelfAttentionLayer(d, H), FFNLayer(d, 256)
    mlp = GenericMLP(input_dim=d, hidden_dims=[d], output_dim=d, use_conv=True, output_use_activation=True,
                     hidden_use_bias=True)
    for name, mod in (("ca", ca), ("sa", sa), ("ffn", ffn), ("mlp", mlp)):
        sd = {k[len(name) + 1:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + "/")}
        assert sorted(sd) == sorted(mod.state_dict())                     # checkpoint-key contract
        mod.load_state_dict(sd)
    t = {k: torch.from_numpy(z[k]) for k in ("tgt", "mem", "pos", "qpos", "qp")}
    K = t["mem"].shape[0]
    mask = torch.from_numpy(np.unpackbits(z["mask"], axis=2)[:, :, :K].astype(bool))
    o1 = ca(t["tgt"], t["mem"], memory_mask=mask, memory_key_padding_mask=None, pos=t["pos"], query_pos=t["qpos"])
    o2 = sa(o1, tgt_mask=None, tgt_key_padding_mask=None, query_pos=t["qpos"])
    o3 = ffn(o2)
    for got, key in ((o1, "o1"), (o2, "o2"), (o3, "o3"), (mlp(t["qp"]), "o4")):
        np.testing.assert_allclose(got.detach().numpy(), z[key], rtol=1e-4, atol=1e-5)
