"""Reader side of the self-training loop: the files the previous round exported
(`freemasks/{scene}_cloud.npy`, `{scene}_masks.npy`, written by trainer/postprocess.py::save_for_freemask) are merged
into a scene's pseudo masks, filtered, and turned into the 9-tuple the collate function takes.

Mirrors reference datasets/freemask_semseg.py — `load_self_train_masks` (:224-265) and `__getitem__` (:267-437) in
validation mode and in TRAIN mode: centring + random shift, axis flips, two elastic distortions, the volume and colour
augmentation pipelines, colour drop and normalisation run on the device (datasets/augment.py), drawing from numpy's and
python's global generators with the reference's own calls and in its order, so a seeded run consumes the same streams
(pinned by tests/golden/dataset.npz for the reference's own steps; the two third-party pipelines are restated from
their published definitions).  Random cuboid cuts / point resampling (point_per_cut, resample_points, noise_rate: 0 in
every shipped config) are not built.

The per-point work (1-NN transfer of the exported masks, the greedy merge, the extent filter) runs on the device."""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import torch

from .. import ops

SCANNET_COLOR_MEAN_STD = ((0.47793125906962, 0.4303257521323044, 0.3749598901421883),
                          (0.2834475483823543, 0.27566157565723015, 0.27018971370874995))


def load_self_train_masks(points, freemasks, self_train_cloud, self_train_masks, num_self_train_data=5,
                          drop_original=False, device="cuda"):
    """points [N,>=3] scene points, freemasks [N,K] the scene's pseudo masks; self_train_cloud [M,>=3] /
    self_train_masks bool[M,K'] the previous round's export, most confident first.
    -> freemasks with up to `num_self_train_data` new columns: an exported instance is added when more than half of
    it is not yet covered, and only its uncovered part is added (freemask_semseg.py:246-265).  Same dtype as the
    input, like the reference's np.concatenate."""
    dev = torch.device(device)
    xyz = torch.as_tensor(np.ascontiguousarray(np.asarray(points)[:, :3]), dtype=torch.float32, device=dev)
    cloud = torch.as_tensor(np.ascontiguousarray(np.asarray(self_train_cloud)[:, :3]), dtype=torch.float32, device=dev)
    st = torch.as_tensor(np.asarray(self_train_masks), device=dev).bool()
    same = xyz.shape[0] == cloud.shape[0] and bool(torch.isclose(xyz, cloud, rtol=1e-5, atol=1e-8).all())
    if not same:                                   # np.allclose(points[:, :3], cloud[:, :3]) failed: 1-NN transfer
        _, ind = ops.knn1(xyz.contiguous(), cloud.contiguous())
        st = st[ind]
    fm = np.asarray(freemasks)
    if drop_original:
        fm = np.zeros((fm.shape[0], 0), dtype=fm.dtype)
    fg = torch.as_tensor(fm, device=dev).bool().any(1) if fm.shape[1] else torch.zeros(xyz.shape[0], dtype=torch.bool,
                                                                                        device=dev)
    totals = st.sum(0).tolist()
    added, j = [], 0
    while len(added) < num_self_train_data and j < st.shape[1]:
        fresh = st[:, j] & ~fg
        # useful_iou = |new & ~foreground| / |new| > 0.5, in integers (an empty instance gives nan > 0.5 == False)
        if totals[j] > 0 and 2 * int(fresh.sum()) > totals[j]:
            added.append(fresh)
            fg = fg | fresh
        j += 1
    if not added:
        return fm
    new = torch.stack(added, 1).cpu().numpy()
    return np.concatenate([fm, new], axis=1)


def filter_by_extent(coordinates, freemasks, hard_threshold=0.5, extent_max_ratio=0.8, device="cuda"):
    """Column indices of the masks that are non-empty and whose XY extent does not exceed `extent_max_ratio` of the
    scene's in either direction (freemask_semseg.py:300-310)."""
    dev = torch.device(device)
    c = torch.as_tensor(np.asarray(coordinates), device=dev)[:, :2]
    m = torch.as_tensor(np.asarray(freemasks), device=dev) > hard_threshold
    scene = (c.amax(0) - c.amin(0)) * extent_max_ratio
    inf = torch.finfo(c.dtype).max
    cc, mm = c[:, None, :], m[:, :, None]
    ext = torch.where(mm, cc, torch.full_like(cc, -inf)).amax(0) - torch.where(mm, cc, torch.full_like(cc, inf)).amin(0)
    keep = m.any(0) & ~(ext > scene).any(1)
    return keep.nonzero().flatten().tolist()


class FreeMaskSceneReader:
    """Validation-mode `SemanticSegmentationFreeDataset.__getitem__`: `{scene}.npy` ([N,12]: xyz, rgb, normal,
    segment id, 2 unused) + `{scene}_freemasks.npy` ([N,K] soft masks) -> the 9-tuple
    (coordinates, features, freemasks+segments, scene_name, raw_color, raw_normals, raw_coordinates, idx, [])."""

    def __init__(self, entries, color_mean_std=SCANNET_COLOR_MEAN_STD, add_colors=True, add_normals=True,
                 add_raw_coordinates=False, freemask_hard_threshold=0.5, freemask_extent_max_ratio=0.8,
                 max_num_gt_instances=-1, load_self_train_data=False, self_train_data_dir=None, num_self_train_data=5,
                 device="cuda", mode="validation", volume_augmentations=None, image_augmentations=None,
                 is_elastic_distortion=True, flip_in_center=False, color_drop=0.0, point_per_cut=0, resample_points=0,
                 noise_rate=0):
        self.data = list(entries)                      # dicts with "filepath" and "raw_filepath" (the database yaml)
        self.color_mean = np.asarray(color_mean_std[0], np.float32) * 255.0
        self.color_den = np.reciprocal(np.asarray(color_mean_std[1], np.float32) * 255.0)
        self.add_colors, self.add_normals, self.add_raw_coordinates = add_colors, add_normals, add_raw_coordinates
        self.freemask_hard_threshold = freemask_hard_threshold
        self.freemask_extent_max_ratio = freemask_extent_max_ratio
        self.max_num_gt_instances = max_num_gt_instances
        self.load_self_train_data = load_self_train_data
        self.self_train_data_dir = self_train_data_dir
        self.num_self_train_data = num_self_train_data
        self.device = device
        self.mode, self.is_elastic_distortion, self.color_drop = mode, is_elastic_distortion, color_drop
        # objects with a `transforms` attribute, called like the reference's (volumentations / albumentations Compose);
        # datasets.augment.VolumeAugmentations() / ColorAugmentations() are the shipped YAML pipelines
        self.volume_augmentations, self.image_augmentations = volume_augmentations, image_augmentations
        if flip_in_center or point_per_cut or resample_points or noise_rate:
            raise NotImplementedError("flip_in_center / point_per_cut / resample_points / noise_rate are off in every "
                                      "shipped config (conf/data/datasets/*.yaml) and not built")

    def __len__(self):
        return len(self.data)

    def _self_train(self, idx, points, freemasks):
        scene_id = Path(self.data[idx]["filepath"]).stem
        name = "masks" if self.load_self_train_data != "refined" else "masks_refined"
        base = os.path.join(self.self_train_data_dir, "freemasks")
        try:
            cloud = np.load(os.path.join(base, f"scene{scene_id}_cloud.npy"))
            masks = np.load(os.path.join(base, f"scene{scene_id}_{name}.npy"))
        except FileNotFoundError:
            print(f"Could not load self training data for scene{scene_id}")
            return freemasks
        return load_self_train_masks(points, freemasks, cloud, masks, self.num_self_train_data, device=self.device)

    def __getitem__(self, idx):
        idx = idx % len(self.data)
        path = self.data[idx]["filepath"].replace("../../", "")
        points = np.load(path)
        freemasks = np.load(path.replace(".npy", "_freemasks.npy"))
        if self.load_self_train_data:
            freemasks = self._self_train(idx, points, freemasks)
        if self.max_num_gt_instances > 0:
            freemasks = freemasks[:, :self.max_num_gt_instances]
        coordinates, color, normals, segments = points[:, :3], points[:, 3:6], points[:, 6:9], points[:, 9]
        keep = filter_by_extent(coordinates, freemasks, self.freemask_hard_threshold, self.freemask_extent_max_ratio,
                                self.device)
        if not keep:
            raise LookupError(f"{path}: no usable pseudo mask")   # the reference resamples a random other scene
        freemasks = freemasks[:, keep]
        hard = freemasks > self.freemask_hard_threshold
        freemasks = np.concatenate([hard.any(1).astype(np.int32).reshape(-1, 1), hard.astype(np.int32)], axis=1)
        raw_coordinates, raw_color, raw_normals = coordinates.copy(), color, normals
        if not self.add_colors:
            color = np.ones((len(color), 3))
        if "train" in self.mode and hasattr(self.volume_augmentations, "transforms"):
            return self._train_item(idx, coordinates, color, normals, segments, freemasks, raw_coordinates, raw_color,
                                    raw_normals)
        # albumentations.Normalize on the uint8-truncated colours (freemask_semseg.py:408-409)
        features = (color.astype(np.uint8).astype(np.float32) - self.color_mean) * self.color_den
        if self.add_normals:
            features = np.hstack((features, normals))
        if self.add_raw_coordinates:
            features = np.hstack((features, coordinates))
        freemasks = np.hstack((freemasks, segments[..., None].astype(freemasks.dtype))).astype(np.int32)
        raw = self.data[idx]["raw_filepath"]
        scene_name = f"scene{raw.split('/')[-1].split('_')[0]}" if "arkit" in raw.lower() else raw.split("/")[-2]
        return coordinates, features, freemasks, scene_name, raw_color, raw_normals, raw_coordinates, idx, []

    def _scene_name(self, idx):
        raw = self.data[idx]["raw_filepath"]
        return f"scene{raw.split('/')[-1].split('_')[0]}" if "arkit" in raw.lower() else raw.split("/")[-2]

    def _train_item(self, idx, coordinates, color, normals, segments, freemasks, raw_coordinates, raw_color, raw_normals):
        """freemask_semseg.py:333-437 with the per-point work on the device; coordinates / features come back as device
        tensors (the collate takes them as they are).  Random numbers: numpy's global generator for the shift and the
        elastic noise, python's `random` for the flips, the elastic gate and the colour drop — same calls, same order."""
        from random import random

        from . import augment as A

        dev = torch.device(self.device)
        c = torch.as_tensor(np.ascontiguousarray(coordinates), device=dev).contiguous()
        nrm = torch.as_tensor(np.ascontiguousarray(normals), device=dev).contiguous()
        col = torch.as_tensor(np.ascontiguousarray(color), dtype=torch.float32, device=dev).contiguous()
        A.center_and_shift(c)
        for i in (0, 1):
            if random() < 0.5:
                A.flip_axis(c, i)
        if random() < 0.95 and self.is_elastic_distortion:
            for granularity, magnitude in ((0.2, 0.4), (0.8, 1.6)):
                A.elastic_distortion(c, granularity, magnitude)
        aug = self.volume_augmentations(points=c, normals=nrm, features=col, labels=freemasks)
        c, col, nrm, freemasks = aug["points"], aug["features"], aug["normals"], aug["labels"]
        tables = self.image_augmentations.tables() if self.image_augmentations is not None else None
        if random() < self.color_drop:
            tables = np.full((3, 256), 255, np.uint8)                  # color[:] = 255
        features = A.color_tables_to_features(col, tables, self.color_mean, self.color_den)
        if self.add_normals:
            features = torch.cat([features, nrm.to(features.dtype)], 1)
        if self.add_raw_coordinates:
            features = torch.cat([features, c.to(features.dtype)], 1)
        freemasks = np.hstack((freemasks, segments[..., None].astype(freemasks.dtype))).astype(np.int32)
        return c, features, freemasks, self._scene_name(idx), raw_color, raw_normals, raw_coordinates, idx, []
