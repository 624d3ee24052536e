"""Synthetic stand-in for SemanticSegmentationFreeDataset (reference datasets/freemask_semseg.py):
returns the same 9-tuple (:434) built from `unscene3d_amd.synthetic.make_scene`."""
import numpy as np

from ..synthetic import make_scene


class SyntheticFreeMaskDataset:
    def __init__(self, n_scenes=8, target_voxels=150_000, seed=3000):
        self.n, self.target_voxels, self.seed = n_scenes, target_voxels, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        sc = make_scene(self.seed + i, self.target_voxels)
        xyz = sc["xyz"]
        feats = np.hstack([sc["colors"], xyz.astype(np.float32)])          # colour | raw xyz (add_raw_coordinates)
        labels = np.ones((xyz.shape[0], 1), np.int32)
        table = np.hstack([labels, sc["masks"].astype(np.int32), sc["segment_ids"][:, None].astype(np.int32)])
        return (xyz, feats, table, f"scene{self.seed + i:04d}_00", sc["colors"], np.zeros_like(sc["colors"]),
                xyz.astype(np.float32), i, sc["segment_connectivity"])
