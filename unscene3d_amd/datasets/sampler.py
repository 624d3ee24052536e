"""Scene sampler of the data-parallel self-training step.

The reference gets its per-rank scene stream from `pl.Trainer(gpus=N)` (main_instance_segmentation.py:86-92), i.e.
`torch.utils.data.DistributedSampler` over the training set (rank r takes indices r, r+W, … of the epoch's
permutation) in front of a DataLoader with `batch_size` scenes per rank (conf/data/indoor.yaml:24-25).  Scenes differ
by up to 3x in voxel count, every rank's backward ends in the same gradient all-reduce, so a step lasts as long as its
LARGEST rank batch: on the bench's own size spread (119 k … 178 k voxels) a step is p10 / p90 = 23.9 / 29.4 ms.

`BucketedDistributedSampler` keeps DistributedSampler's contract — every scene exactly once per epoch over all ranks
(padded by wrap-around unless `drop_last`), the same number of steps on every rank, a plan that is a pure function of
(seed, epoch) and therefore identical on every rank without communication — and adds size bucketing: the epoch's
permutation is cut into windows of `window` steps, the scenes of a window are sorted by size and consecutive groups
of W·B of them form a step (dealt to the ranks in snake order when B > 1, so that the per-rank SUMS balance), then the
steps of the window are shuffled again.  `window = 1` is exactly DistributedSampler + DataLoader batching.
"""
from __future__ import annotations

import numpy as np
import torch


class BucketedDistributedSampler:
    """sizes: one number per scene (points or voxels — anything monotone in the step's cost, known without loading the
    scene: the preprocessing database of the reference stores the point count of every scene file).
    Iterating yields, per step, the list of `batch_size` scene indices of THIS rank."""

    def __init__(self, sizes, num_replicas: int, rank: int, batch_size: int = 1, window: int = 8, shuffle: bool = True,
                 seed: int = 0, drop_last: bool = False):
        if not (0 <= rank < num_replicas):
            raise ValueError(f"rank {rank} outside [0, {num_replicas})")
        if batch_size < 1 or window < 1:
            raise ValueError("batch_size and window must be >= 1")
        self.sizes = np.asarray(sizes, dtype=np.float64).reshape(-1)
        self.n = int(self.sizes.shape[0])
        self.world, self.rank, self.batch_size, self.window = int(num_replicas), int(rank), int(batch_size), int(window)
        self.shuffle, self.seed, self.drop_last, self.epoch = bool(shuffle), int(seed), bool(drop_last), 0
        per_step = self.world * self.batch_size
        if self.n == 0 or (self.drop_last and self.n < per_step):
            raise ValueError("not enough scenes for one step")
        self.steps = self.n // per_step if self.drop_last else -(-self.n // per_step)

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def __len__(self):
        return self.steps

    def _permutation(self, epoch):
        if self.shuffle:                       # the generator and call DistributedSampler uses: same order at window 1
            g = torch.Generator()
            g.manual_seed(self.seed + epoch)
            perm = torch.randperm(self.n, generator=g).numpy()
        else:
            perm = np.arange(self.n)
        total = self.steps * self.world * self.batch_size
        if total <= self.n:
            return perm[:total]
        reps = -(-total // self.n)
        return np.concatenate([perm] * reps)[:total]          # wrap-around padding (DistributedSampler's rule)

    def plan(self, epoch: int | None = None) -> np.ndarray:
        """i64[steps, world, batch_size]: the scene of every (step, rank, slot) of one epoch — the same array on every
        rank."""
        epoch = self.epoch if epoch is None else int(epoch)
        W, B = self.world, self.batch_size
        perm = self._permutation(epoch)
        if self.window == 1:
            # DistributedSampler: rank r's list is perm[r::W]; the DataLoader batches B consecutive entries of it
            return perm.reshape(self.steps, B, W).transpose(0, 2, 1).copy()
        rng = np.random.default_rng([self.seed, epoch, 0x5CE4E])
        out = np.empty((self.steps, W, B), dtype=np.int64)
        s0 = 0
        while s0 < self.steps:
            ns = min(self.window, self.steps - s0)
            if self.steps - (s0 + ns) < self.window:                           # a short tail window would be as uneven
                ns = self.steps - s0                                           # as no bucketing: merge it into this one
            idx = perm[s0 * W * B:(s0 + ns) * W * B]
            order = idx[np.argsort(-self.sizes[idx], kind="stable")]          # largest first
            groups = order.reshape(ns, B, W)                                   # a step = W*B scenes of adjacent size
            for j, s in enumerate(rng.permutation(ns) if self.shuffle else range(ns)):
                grp = groups[s].copy()
                grp[1::2] = grp[1::2, ::-1]                                    # snake: round b deals in reverse order
                shift = int(rng.integers(W)) if self.shuffle else 0            # no rank always holds the largest scene
                out[s0 + j] = np.roll(grp.T, shift, axis=0)
            s0 += ns
        return out

    def __iter__(self):
        for step in self.plan()[:, self.rank, :]:
            yield [int(i) for i in step]

    def imbalance(self, epoch: int | None = None) -> dict:
        """Per step max-over-ranks / mean-over-ranks of the summed sizes (1.0 = perfectly even): what the gradient
        all-reduce makes every rank wait for."""
        per_rank = self.sizes[self.plan(epoch)].sum(axis=2)                    # [steps, world]
        ratio = per_rank.max(axis=1) / per_rank.mean(axis=1)
        return {"mean": float(ratio.mean()), "max": float(ratio.max()), "p90": float(np.quantile(ratio, 0.9))}
