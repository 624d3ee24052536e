#!/usr/bin/env python
"""Soak runs of the two loops around the training step: the pseudo-mask driver (R rounds of K scenes in flight) and the
validation / export step (`eval_step`) — device memory with the cyclic collector OFF must stay flat.
Usage (GPU box): python tools/soak_aux.py"""
import gc
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def mb():
    torch.cuda.synchronize()
    return round(torch.cuda.memory_allocated() / 2**20, 1)


def ncut_soak(dev, K=16, rounds=6):
    from unscene3d_amd.pseudo_masks.driver import PseudoMaskDriver
    from unscene3d_amd.synthetic import make_segment_scene

    feats, conn, _ = make_segment_scene(75, side=25, dims=(384, 96), n_objects=16)
    S = feats[0].shape[0]
    dfe = tuple(torch.from_numpy(f).to(dev) for f in feats)
    driver = PseudoMaskDriver(device=dev, concurrent=K)

    def run():
        scenes = [{"features": (dfe[0].clone(), dfe[1].clone()), "unique_segments": torch.arange(S),
                   "seg_connectivity": torch.from_numpy(conn)} for _ in range(K)]
        return driver.run(scenes)[0]
    run()
    gc.collect()
    gc.disable()
    out = [mb()]
    for _ in range(rounds):
        run()
        out.append(mb())
    gc.enable()
    n = gc.collect()
    return {"scenes_per_round": K, "allocated_MB_per_round": out, "unreachable_after": n, "allocated_MB_after_collect": mb()}


def eval_soak(dev, steps=12):
    import bench
    args = bench.parse(["--no-cpu-baseline", "--voxels", "60000", "--rotate", "0"])
    step = bench.make_mask3d_step(args, dev, 0, 1)
    module = step.module
    from unscene3d_amd.datasets.synthetic import SyntheticFreeMaskDataset
    from unscene3d_amd.datasets.utils import FreeMaskVoxelizeCollate
    for _ in range(3):
        step(1)
    module.eval()
    coll = FreeMaskVoxelizeCollate(ignore_label=255, voxel_size=0.02, mode="validation", device=str(dev))
    ds = SyntheticFreeMaskDataset(n_scenes=2, target_voxels=60000, seed=4242)
    with torch.no_grad():
        module.eval_step(coll([ds[0]]), 0)
    gc.collect()
    gc.disable()
    out = [mb()]
    with torch.no_grad():
        for k in range(steps):
            module.eval_step(coll([ds[k % 2]]), k)
            if k % 3 == 2:
                out.append(mb())
    gc.enable()
    n = gc.collect()
    step.close()
    return {"eval_steps": steps, "allocated_MB_every_3": out, "unreachable_after": n, "allocated_MB_after_collect": mb()}


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    res = {"ncut_driver": ncut_soak(dev)}
    try:
        res["eval_step"] = eval_soak(dev)
    except Exception as e:      # report, do not hide
        res["eval_step"] = {"error": repr(e)}
    print(json.dumps(res))
