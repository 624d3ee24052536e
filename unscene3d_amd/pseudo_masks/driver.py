"""Scene-list driver of the pseudo-mask generator (reference pseudo_masks/unscene3d_pseudo_main.py:532-667: one
process walks all scenes of a split one after the other).

SURVEY.md §8e: scenes are independent, so the path scales by REPLICAS — no collective.  Two levels:

* across GPUs: rank r of W takes scenes r, r+W, r+2W, ... (`scene_shard`, static sharding by index mod W; ranks are
  separate processes, e.g. `torchrun --nproc-per-node W tools/pseudo_masks_run.py --scenes DIR --out OUT`, and never talk to each other);
* inside one GPU: `concurrent` scenes in flight, each on its own HIP stream, driven by ONE host thread.  A scene's
  masked-NCut loop is a chain of ~20 latency-bound eigen-solves (each a persistent launch of 64 workgroups = a quarter
  of the 256 CUs) separated by a small device->host copy and some S-sized host logic; `ncut.unscene3d_steps` is that
  loop as a generator which yields at the copy, so the scheduler here resumes whichever scene's eigenvector has
  arrived, does its host logic, queues its next iteration and moves on — the other scenes' solves keep the GPU busy
  meanwhile.  (Round 2 gave every scene its own host THREAD: 6.0 scenes/s with one scene in flight, 9.7 with two,
  6.3 with four — the threads queued on the interpreter lock.  Round 3: the per-iteration host logic lost its 6 ms
  rebuild of the neighbour sets, and the scenes share one thread.)  Measured on the 625-segment bench scene
  (`bench.py --mode ncut --scenes K`): 8.1 scenes/s with one scene in flight (eigen-solve bound: 5.6 ms x 20
  iterations), 11.1 / 14.5 / 16.1 / 21.0 / 23.5 / 25.0 / 26.7 with 2 / 3 / 4 / 6 / 12 / 16 / 24.  A solve's 64 workgroups hold
  only 20 KB of LDS each, so the solves of many scenes share the CUs; smaller grids per solve (USC3D_TRI_G = 48, 32)
  were measured slower at every K.  Default: 16 in flight.
"""
from __future__ import annotations

import torch

from . import ncut


def scene_shard(n_scenes: int, rank: int = 0, world: int = 1):
    """Indices of the scenes rank `rank` of `world` processes (static, index mod W)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside [0, {world})")
    return list(range(rank, n_scenes, world))


NCUT_DEFAULTS = dict(affinity_tau=0.6, max_number_of_instances=20, min_segment_size=4, separation_mode="max",
                     max_extent_ratio=0.8)


def default_scene_fn(scene, **kw):
    """scene: dict with `features` (tensor [S,d] or a tuple of two), `unique_segments`, `seg_connectivity` (directed
    pairs) -> bool[K, S] masks over segments (ncut.unscene3d with the published settings unless overridden)."""
    args = dict(NCUT_DEFAULTS)
    args.update(kw)
    return ncut.unscene3d(scene["features"], scene["unique_segments"], scene["seg_connectivity"], **args)


def default_scene_steps(scene, **kw):
    """The same as a generator (ncut.unscene3d_steps): yields the event of every eigenvector copy."""
    args = dict(NCUT_DEFAULTS)
    args.update(kw)
    return ncut.unscene3d_steps(scene["features"], scene["unique_segments"], scene["seg_connectivity"], **args)


class PseudoMaskDriver:
    def __init__(self, device="cuda", concurrent: int = 16, rank: int = 0, world: int = 1, scene_fn=None,
                 scene_steps=default_scene_steps):
        """scene_steps(scene, **kw) -> generator yielding torch.cuda.Event (the default: the masked-NCut loop);
        scene_fn(scene, **kw) -> masks: a plain function instead (then every scene runs to completion on its stream,
        `concurrent` has no effect)."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the pseudo-mask generator runs on the HIP device; there is no CPU path")
        self.concurrent, self.rank, self.world = max(1, int(concurrent)), rank, world
        self.scene_fn, self.scene_steps = scene_fn, scene_steps
        self._streams = [torch.cuda.Stream(device=self.device) for _ in range(self.concurrent)]

    def run(self, scenes, **kw):
        """scenes: sequence (the WHOLE split; this rank takes its shard).  -> {scene index: masks} for this rank."""
        mine = scene_shard(len(scenes), self.rank, self.world)
        if not mine:
            return {}
        torch.cuda.set_device(self.device)
        main = torch.cuda.current_stream(self.device)
        for st in self._streams:                               # inputs may have been produced on the caller's stream
            st.wait_stream(main)
        out = {}
        if self.scene_fn is not None:
            for n, i in enumerate(mine):
                with torch.cuda.stream(self._streams[n % self.concurrent]):
                    out[i] = self.scene_fn(scenes[i], **kw)
            for st in self._streams:
                st.synchronize()
            return out
        todo = list(mine)
        slots = {}                                             # stream slot -> (scene index, generator, pending event)

        def advance(slot, i, gen):
            """Resume scene i on its stream until its next copy event (or its end); a scene that ends hands the slot to
            the next one of the list.  A loop, not a recursion: a long run of scenes that finish on their first step
            (fewer than 3 segments return at once) must not grow the Python stack."""
            with torch.cuda.stream(self._streams[slot]):
                while True:
                    try:
                        slots[slot] = (i, gen, next(gen))
                        return
                    except StopIteration as done:
                        out[i] = done.value
                        slots.pop(slot, None)
                        if not todo:
                            return
                        i = todo.pop(0)
                        gen = self.scene_steps(scenes[i], **kw)

        for slot in range(self.concurrent):
            if not todo:                                       # (advance() itself takes further scenes off the list)
                break
            j = todo.pop(0)
            advance(slot, j, self.scene_steps(scenes[j], **kw))
        while slots:
            ready = [sl for sl, (_, _, ev) in slots.items() if ev.query()]
            if not ready:                                      # nothing has arrived: wait for the oldest request
                sl = next(iter(slots))
                slots[sl][2].synchronize()
                ready = [sl]
            for sl in ready:
                i, gen, _ = slots[sl]
                advance(sl, i, gen)
        main.wait_stream(self._streams[0])
        return out
