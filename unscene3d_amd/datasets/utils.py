"""Collate for the self-training path (reference datasets/utils.py:181-219, :370-527, :670-687):
`FreeMaskVoxelizeCollate` / `freemask_voxelize` turn a list of dataset 9-tuples
(coordinates, features, freemasks[labels | K mask columns | segment id], scene, raw_color, raw_normals,
raw_coordinates, idx, segment_connectivity — reference datasets/freemask_semseg.py:434) into
(NoGpu(coordinates i32[N,4], features f32[N,C], …), targets, scene names).

MI355X differences: the 2 cm voxelisation (`np.floor(xyz/voxel)` + ME.utils.sparse_quantize, reference
:403-408) runs on the device through the hash-unique kernel instead of in CPU DataLoader workers, and
the voxel rows may optionally be re-ordered into z-order cells (`spatial_sort`: True = 8^3-voxel cells, or the cell
size as a power of two, e.g. 5 = 32^3 voxels), a consistent permutation of every per-voxel array (inverse maps are
remapped accordingly)."""
from __future__ import annotations

import numpy as np
import torch

from .. import MinkowskiEngine as ME
from .. import ops


def _dev(x, dev, dtype):
    """numpy array or tensor (possibly already resident in HBM) -> contiguous device tensor."""
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    return t.to(device=dev, dtype=dtype).contiguous()


class NoGpu:
    """Plain container (reference :670-687)."""

    def __init__(self, coordinates, features, original_labels=None, inverse_maps=None, full_res_coords=None,
                 target_full=None, original_colors=None, original_normals=None, original_coordinates=None, idx=None,
                 segment_connectivity=None):
        self.coordinates, self.features = coordinates, features
        self.original_labels, self.inverse_maps, self.full_res_coords = original_labels, inverse_maps, full_res_coords
        self.target_full, self.original_colors, self.original_normals = target_full, original_colors, original_normals
        self.original_coordinates, self.idx, self.segment_connectivity = original_coordinates, idx, segment_connectivity


def get_instance_freemasks(list_freemasks, list_segments=None):
    """Targets from [labels | K mask columns | segment id] tables (reference :480-527).

    Every non-empty mask column becomes one foreground target (label 1).  With segments, the segment
    mask of a target is set at `unique(ALL values of the rows inside the mask)` — the reference takes
    the unique over whole table rows (label column and the 0/1 mask values included, :503), which is
    reproduced here.  Vectorised over the K columns (one host sync per scene instead of K)."""
    target = []
    for b, table in enumerate(list_freemasks):
        colsT = (table[:, 1:-1] != 0).T.contiguous()                # [K, N] hard masks (row reductions are fast)
        keep = torch.nonzero(colsT.any(1)).reshape(-1)              # non-empty columns (host sync)
        if keep.numel() == 0:
            return []
        masks = colsT[keep]                                         # [T, N]
        entry = {"labels": torch.ones(keep.numel(), dtype=torch.int64, device=table.device), "masks": masks}
        if list_segments:
            S = list_segments[b].shape[0]
            tt, rr = torch.nonzero(masks, as_tuple=True)            # (target, row) pairs
            vals = table[rr]                                        # [P, K+2] every value of those rows
            sm = torch.zeros((keep.numel(), S), dtype=torch.bool, device=table.device)
            sm[tt[:, None].expand_as(vals).reshape(-1), vals.reshape(-1)] = True
            entry["segment_mask"] = sm
        target.append(entry)
    return target


def freemask_voxelize(batch, ignore_label, voxel_size, mode, ignore_class_threshold, device="cuda",
                      spatial_sort=False):
    dev = torch.device(device)
    coords_l, feats_l, tables, inverse_maps = [], [], [], []
    full_res_coords, original_freemasks, colors, normals, raw_coords, idx, seg_conn = [], [], [], [], [], [], []
    for sample in batch:
        full_res_coords.append(sample[0])
        original_freemasks.append(sample[2])
        colors.append(sample[4])
        normals.append(sample[5])
        raw_coords.append(sample[6])
        idx.append(sample[7])
        seg_conn.append(sample[8])
        xyz = _dev(sample[0], dev, torch.float64)
        c3, unique_map, inverse_map = ME.utils.sparse_quantize(xyz, quantization_size=voxel_size, return_index=True,
                                                               return_inverse=True, device=str(dev))
        if spatial_sort:
            c4 = torch.cat([torch.zeros((c3.shape[0], 1), dtype=torch.int32, device=dev), c3], 1).contiguous()
            order = ops.spatial_order(c4, shift=int(spatial_sort) if spatial_sort is not True else 3)
            rank = torch.empty_like(order)
            rank[order] = torch.arange(order.shape[0], device=dev)
            c3, unique_map, inverse_map = c3[order], unique_map[order], rank[inverse_map]
        inverse_maps.append(inverse_map)
        coords_l.append(c3.int())
        feats_l.append(_dev(sample[1], dev, torch.float32)[unique_map])
        if len(sample[2]) > 0:
            tables.append(_dev(sample[2], dev, torch.int64)[unique_map])

    if tables:   # pad the mask columns to a common width, keeping the segment id as the last column
        width = max(t.shape[1] for t in tables)
        tables = [torch.cat([t[:, :-1], t.new_zeros(t.shape[0], width - t.shape[1]), t[:, -1:]], dim=1) for t in tables]
        coordinates, features, _ = ME.utils.sparse_collate(coords_l, feats_l, tables)
    else:
        coordinates, features = ME.utils.sparse_collate(coords_l, feats_l)

    target, target_full = [], []
    if tables:
        segment2label, n_segments = [], []
        for t in tables:
            seg = t[:, -1]
            uniq, inv = torch.unique(seg, return_inverse=True)
            n_segments.append(int(uniq.shape[0]))           # known on the host here: saves the model a read-back
            first = torch.full((uniq.shape[0],), seg.shape[0], dtype=torch.long, device=dev)
            first.scatter_reduce_(0, inv, torch.arange(seg.shape[0], device=dev), reduce="amin")
            t[:, -1] = inv                                   # contiguous segment ids (np.unique return_inverse)
            segment2label.append(t[first][:, :-1])
        target = get_instance_freemasks(tables, list_segments=segment2label)
        for i in range(len(target)):
            target[i]["point2segment"] = tables[i][:, -1].contiguous()   # a row-gather index: the kernels take dense i64
            target[i]["num_segments"] = torch.tensor(n_segments[i])     # host scalar (a tensor like the other entries)
        full = [m if isinstance(m, torch.Tensor) else torch.as_tensor(np.asarray(m)) for m in original_freemasks]
        target_full = get_instance_freemasks(full)
        for i in range(len(target_full)):
            target_full[i]["point2segment"] = full[i][:, -1].long()
    else:
        coordinates, features = [], []
    return (NoGpu(coordinates, features, original_freemasks, inverse_maps, full_res_coords, target_full, colors,
                  normals, raw_coords, idx, seg_conn), target, [sample[3] for sample in batch])


class FreeMaskVoxelizeCollate:
    def __init__(self, ignore_label=255, voxel_size=1, mode="test", small_crops=False, very_small_crops=False,
                 batch_instance=False, probing=False, task="instance_segmentation", ignore_class_threshold=100,
                 filter_out_classes=(), label_offset=0, num_queries=None, device="cuda", spatial_sort=False):
        assert task in ["instance_segmentation"], "task not known"
        if small_crops or very_small_crops:
            raise NotImplementedError("crop collates are outside the accelerated path (SURVEY.md §2.1)")
        self.ignore_label, self.voxel_size, self.mode = ignore_label, voxel_size, mode
        self.ignore_class_threshold, self.device, self.spatial_sort = ignore_class_threshold, device, spatial_sort

    def __call__(self, batch):
        return freemask_voxelize(batch, self.ignore_label, self.voxel_size, self.mode, self.ignore_class_threshold,
                                 device=self.device, spatial_sort=self.spatial_sort)
