"""Scene-list driver of the pseudo-mask generator (reference pseudo_masks/unscene3d_pseudo_main.py:532-667: one
process walks all scenes of a split one after the other).

SURVEY.md §8e: scenes are independent, so the path scales by REPLICAS — no collective.  Two levels:

* across GPUs: rank r of W takes scenes r, r+W, r+2W, ... (`scene_shard`, static sharding by index mod W; ranks are
  separate processes, e.g. `torchrun --nproc-per-node W tools/pseudo_masks_run.py --scenes DIR --out OUT`, and never talk to each other);
* inside one GPU: `concurrent` scenes at a time, each on its own host thread and its own HIP stream.  One masked-NCut
  loop is a chain of ~20 latency-bound eigen-solves, each a persistent launch of 64 workgroups that keeps a quarter
  of the 256 CUs busy and ends in a small device->host copy (the bipartition logic runs on the host, like the
  reference's); the GIL is released while a thread waits for its copy.  Measured on the 625-segment bench scene
  (`bench.py --mode ncut --scenes K`): 6.0 scenes/s with one scene in flight, 9.7 with two, 6.3 with four, 5.3 with
  six — beyond two, the persistent launches (64 spinning workgroups each) starve the short kernels between them and
  the host threads queue on the interpreter lock.  Default: two.
"""
from __future__ import annotations

import threading
from concurrent.futures import ThreadPoolExecutor

import torch

from . import ncut


def scene_shard(n_scenes: int, rank: int = 0, world: int = 1):
    """Indices of the scenes rank `rank` of `world` processes (static, index mod W)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside [0, {world})")
    return list(range(rank, n_scenes, world))


def default_scene_fn(scene, **kw):
    """scene: dict with `features` (tensor [S,d] or a tuple of two), `unique_segments`, `seg_connectivity` (directed
    pairs) -> bool[K, S] masks over segments (ncut.unscene3d with the published settings unless overridden)."""
    args = dict(affinity_tau=0.6, max_number_of_instances=20, min_segment_size=4, separation_mode="max",
                max_extent_ratio=0.8)
    args.update(kw)
    return ncut.unscene3d(scene["features"], scene["unique_segments"], scene["seg_connectivity"], **args)


class PseudoMaskDriver:
    def __init__(self, device="cuda", concurrent: int = 2, rank: int = 0, world: int = 1, scene_fn=default_scene_fn):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the pseudo-mask generator runs on the HIP device; there is no CPU path")
        self.concurrent, self.rank, self.world, self.scene_fn = max(1, int(concurrent)), rank, world, scene_fn
        self._streams = [torch.cuda.Stream(device=self.device) for _ in range(self.concurrent)]
        self._free = list(range(self.concurrent))
        self._lock = threading.Lock()

    def _run_one(self, scene, kw):
        with self._lock:
            slot = self._free.pop()
        try:
            torch.cuda.set_device(self.device)                 # the current device is per host thread
            with torch.cuda.stream(self._streams[slot]):
                out = self.scene_fn(scene, **kw)
            self._streams[slot].synchronize()
            return out
        finally:
            with self._lock:
                self._free.append(slot)

    def run(self, scenes, **kw):
        """scenes: sequence (the WHOLE split; this rank takes its shard).  -> {scene index: masks} for this rank."""
        mine = scene_shard(len(scenes), self.rank, self.world)
        if not mine:
            return {}
        main = torch.cuda.current_stream(self.device)
        for st in self._streams:                               # inputs may have been produced on the caller's stream
            st.wait_stream(main)
        if self.concurrent == 1:
            return {i: self._run_one(scenes[i], kw) for i in mine}
        with ThreadPoolExecutor(max_workers=self.concurrent) as pool:
            futures = {i: pool.submit(self._run_one, scenes[i], kw) for i in mine}
            return {i: f.result() for i, f in futures.items()}
