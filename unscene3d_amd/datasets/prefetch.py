"""The next batch's coordinate work on a side stream, while the current step's backward runs.

Voxelisation, the coordinate hash, the strided maps of the U-Net pyramid, the neighbour tables and their row orders
depend only on the scene's coordinates, and each of them ends in a count read-back.  Issued at the start of a step
on the compute stream, every read-back waits for whatever the previous step still has queued (backward, optimizer)
and the host cannot start issuing the forward pass — with host issue time and device time level that
serialises the two.  The reference hides this work in DataLoader workers (datasets/utils.py:181-219 runs on the CPU,
concurrently with the GPU step); here it runs on the GPU, on its own HIP stream: its read-backs only wait for its own
short kernels, and the step that consumes the batch finds the maps cached in the SparseTensor's coordinate manager.

Tensors allocated on the side stream are handed to the compute stream with `record_stream` (the caching allocator
then keeps a freed block away from the side stream until the compute stream is past it), and every prefetched batch
is kept referenced until the batch after it has been consumed."""
from __future__ import annotations

import collections

import torch

from .. import MinkowskiEngine as ME


def _record_streams(obj, stream, seen):
    """record_stream(stream) on every HIP tensor reachable from obj (containers, object attributes, attributes hung
    on tensors such as the cached row orders of a neighbour table)."""
    if obj is None or isinstance(obj, (int, float, str, bool, bytes, slice)) or id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
        d = getattr(obj, "__dict__", None)
        if d:
            _record_streams(d, stream, seen)
        return
    if isinstance(obj, dict):
        for v in obj.values():
            _record_streams(v, stream, seen)
        return
    if isinstance(obj, (list, tuple, set, collections.deque)):
        for v in obj:
            _record_streams(v, stream, seen)
        return
    d = getattr(obj, "__dict__", None)
    if d:
        _record_streams(d, stream, seen)
    for name in getattr(type(obj), "__slots__", ()):
        _record_streams(getattr(obj, name, None), stream, seen)


class ScenePrefetcher:
    """prefetch = ScenePrefetcher(collate, add_raw_coordinates, n_down);  prefetch.submit(samples) issues the work for
    the next batch; prefetch.take() -> (data, target, names) with `data.sparse_tensor` (maps prepared) and
    `data.raw_coordinates` attached, ready for `InstanceSegmentation.training_step`."""

    def __init__(self, collate, add_raw_coordinates: bool = True, n_down: int = 4, ksize: int = 3, device="cuda",
                 precompute=None):
        """precompute: optional `Mask3D.precompute_geometry` (bound method): the parameter-free, geometry-only part
        of the model's forward pass is then issued here as well."""
        self.collate, self.add_raw, self.n_down, self.ksize = collate, add_raw_coordinates, n_down, ksize
        self.precompute = precompute
        self.device = torch.device(device)
        self.side = torch.cuda.Stream(device=self.device)
        self._pending = None
        self._keep = collections.deque(maxlen=2)

    def submit(self, samples):
        with torch.cuda.stream(self.side):
            data, target, names = self.collate(samples)
            feats, raw = data.features, None
            if self.add_raw:
                raw = feats[:, -3:].contiguous()
                feats = feats[:, :-3].contiguous()
            x = ME.SparseTensor(coordinates=data.coordinates, features=feats, device=self.device)
            x.coordinate_manager.prepare(x.tensor_stride[0], n_down=self.n_down, ksize=self.ksize)
            if self.precompute is not None and raw is not None and len(target) > 0:
                ns = [t.get("num_segments") for t in target]
                self.precompute(x, raw, [t["point2segment"] for t in target],
                                None if any(n is None or n.is_cuda for n in ns) else ns, n_levels=self.n_down + 1)
            data.sparse_tensor, data.raw_coordinates = x, raw
            done = torch.cuda.Event()
            done.record(self.side)
        self._pending = ((data, target, names), done)

    def take(self):
        if self._pending is None:
            raise RuntimeError("ScenePrefetcher.take() without a submit()")
        batch, done = self._pending
        self._pending = None
        main = torch.cuda.current_stream(self.device)
        main.wait_event(done)
        _record_streams(batch, main, set())
        self._keep.append(batch)
        return batch
