"""Configuration values of the canonical self-training run, without hydra.

Defaults mirror the reference's YAML tree for the overrides used by
scripts/unsupervised/train_unscene3d.sh:10-24: conf/config_base_instance_segmentation.yaml,
conf/model/mask3d.yaml, conf/matcher/hungarian_matcher.yaml, conf/loss/set_criterion.yaml,
conf/optimizer/adamw.yaml, conf/trainer/trainer600.yaml, conf/data/indoor.yaml.
`apply_overrides(cfg, ["general.num_targets=3", …])` accepts the same `a.b=value` grammar."""
from __future__ import annotations

import ast
from types import SimpleNamespace


def _ns(d):
    return SimpleNamespace(**{k: _ns(v) if isinstance(v, dict) else v for k, v in d.items()})


def default_config():
    num_targets = 3
    return _ns({
        "general": {"num_targets": num_targets, "train_on_segments": True, "eval_on_segments": True,
                    "ignore_mask_idx": [], "max_batch_size": 99999999, "gpus": 1, "freeze_backbone": False,
                    "decoder_id": -1, "use_dbscan": False, "dbscan_eps": 0.95, "dbscan_min_points": 1,
                    # export between self-training rounds (conf/config_base_instance_segmentation.yaml:12-44)
                    "filter_out_instances": False, "scores_threshold": 0.1, "iou_threshold": 0.66,
                    "topk_per_image": 100, "save_for_freemask": False, "save_dir": "saved"},
        "data": {"voxel_size": 0.02, "in_channels": 3, "num_labels": 20, "add_raw_coordinates": True,
                 "add_colors": True, "add_normals": False, "ignore_label": 255, "batch_size": 8,
                 "test_mode": "validation"},          # conf/data/indoor.yaml:9 ("test": eval_step skips the criterion)
        "model": {"hidden_dim": 128, "dim_feedforward": 1024, "num_queries": 100, "num_heads": 8, "num_decoders": 3,
                  "dropout": 0.0, "pre_norm": False, "use_level_embed": False, "normalize_pos_enc": True,
                  "positional_encoding_type": "fourier", "gauss_scale": 1.0, "hlevels": [0, 1, 2, 3],
                  "non_parametric_queries": True, "random_query_both": False, "random_normal": False,
                  "random_queries": False, "use_np_features": False, "sample_sizes": [200, 800, 3200, 12800, 51200],
                  "max_sample_size": False, "shared_decoder": True, "num_classes": num_targets,
                  "train_on_segments": True, "scatter_type": "mean", "voxel_size": 0.02,
                  "backbone": {"name": "Res16UNet34C", "bn_momentum": 0.02, "conv1_kernel_size": 3,
                               "dialations": [1, 1, 1, 1], "out_fpn": True}},
        "matcher": {"cost_class": 2.0, "cost_mask": 5.0, "cost_dice": 2.0, "cost_noise_robust": 0.0, "num_points": -1},
        "loss": {"num_classes": num_targets, "eos_coef": 0.1, "losses": ["labels", "masks"], "num_points": -1,
                 "oversample_ratio": 3.0, "importance_sample_ratio": 0.75, "class_weights": -1, "directions": "xyz",
                 "use_droploss": False, "droploss_iou_thresh": 0.1},
        "optimizer": {"lr": 1e-4},
        "trainer": {"max_epochs": 601, "check_val_every_n_epoch": 5},
    })


def apply_overrides(cfg, overrides):
    for ov in overrides:
        key, _, val = ov.partition("=")
        try:
            val = ast.literal_eval(val)
        except Exception:
            val = {"true": True, "false": False, "null": None}.get(val.lower(), val)
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = getattr(node, p)
        setattr(node, parts[-1], val)
    # interpolations of the reference YAML (${general.num_targets}, ${general.train_on_segments}, …)
    cfg.model.num_classes = cfg.general.num_targets
    cfg.loss.num_classes = cfg.general.num_targets
    cfg.model.train_on_segments = cfg.general.train_on_segments
    cfg.loss.num_points = cfg.matcher.num_points
    cfg.model.voxel_size = cfg.data.voxel_size
    return cfg


def instantiate_model(cfg):
    """hydra.utils.instantiate(config.model) of the reference (trainer/trainer.py:60)."""
    from .models import res16unet
    from .models.mask3d import Mask3D

    m, b = cfg.model, cfg.model.backbone
    bb_cfg = SimpleNamespace(bn_momentum=b.bn_momentum, conv1_kernel_size=b.conv1_kernel_size, dilations=b.dialations)
    backbone = getattr(res16unet, b.name)(cfg.data.in_channels, cfg.data.num_labels, bb_cfg, out_fpn=b.out_fpn)
    return Mask3D(config=SimpleNamespace(backbone=backbone), hidden_dim=m.hidden_dim, num_queries=m.num_queries,
                  num_heads=m.num_heads, dim_feedforward=m.dim_feedforward, sample_sizes=m.sample_sizes,
                  shared_decoder=m.shared_decoder, num_classes=m.num_classes, num_decoders=m.num_decoders,
                  dropout=m.dropout, pre_norm=m.pre_norm, positional_encoding_type=m.positional_encoding_type,
                  non_parametric_queries=m.non_parametric_queries, train_on_segments=m.train_on_segments,
                  normalize_pos_enc=m.normalize_pos_enc, use_level_embed=m.use_level_embed,
                  scatter_type=m.scatter_type, hlevels=m.hlevels, use_np_features=m.use_np_features,
                  voxel_size=m.voxel_size, max_sample_size=m.max_sample_size, random_queries=m.random_queries,
                  gauss_scale=m.gauss_scale, random_query_both=m.random_query_both, random_normal=m.random_normal)
