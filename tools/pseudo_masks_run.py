#!/usr/bin/env python
"""Scene-list entry of the pseudo-mask generator (reference pseudo_masks/unscene3d_pseudo_main.py:532-667 `main`):
for every scene of a list: per-voxel features -> per-segment features (aggregate_features) -> masked NCut
(unscene3d) -> voxel level (inverse segment mapping, :591-599) -> full-resolution lift by 1-NN (:649-655) ->
`{scene}_cloud.npy` / `{scene}_masks.npy` (:666-667), skipping scenes whose cloud file exists (:550-551).

Replicas only: rank r of W takes the scenes r, r+W, ... and never talks to the others:

    python tools/pseudo_masks_run.py --scenes DIR --out OUT                         # one GPU
    python -m torch.distributed.run --nproc-per-node 8 tools/pseudo_masks_run.py --scenes DIR --out OUT
    python tools/pseudo_masks_run.py --synthetic 6 --out /tmp/pm                   # self-contained demo scenes

Scene files (`DIR/*.npz`, one per scene; the 2D/3D encoders — DINO ViT, CSC Res16UNet — are stock networks whose
weights this offline build cannot fetch, so their per-voxel outputs are an input here; `encode_scene_feats_2d/_3d`
of unscene3d_amd.pseudo_masks.pipeline produce them from frames / a backbone when the weights are at hand):
    coords            int[N,3] or [N,4]   voxel coordinates (b,x,y,z or x,y,z)
    feats_2d, feats_3d f32[N,d]            per-voxel features of one or both modalities (all-zero rows = not seen)
    segment_ids       int[N]              over-segmentation id per voxel (felzenszwalb_cpp.segment_mesh + 1-NN)
    seg_connectivity  int[E,2]            directed segment adjacency
    full_res_coords   f32[M,3]            the scene's full-resolution points (metres)
    voxel_size        float               default 0.02
"""
import argparse
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def synthetic_scene_files(n, out_dir, seed=500):
    """Small self-contained scenes in the file format above (room voxels, block-structured two-modality features)."""
    from unscene3d_amd.synthetic import make_scene
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for k in range(n):
        sc = make_scene(seed + k, target_voxels=6000 + 1500 * k, tol=0.1)
        vox = np.floor(sc["xyz"] / 0.02).astype(np.int64)
        uniq, first = np.unique(vox, axis=0, return_index=True)
        first.sort()
        coords = vox[first].astype(np.int32)
        seg = sc["segment_ids"][first].astype(np.int64)
        obj = sc["masks"][first].argmax(1)                          # object id per voxel -> feature cluster
        rng = np.random.default_rng(seed + k)
        n_obj = int(obj.max()) + 1
        feats = []
        for d in (96, 48):
            centres = rng.standard_normal((n_obj, d)).astype(np.float32)
            f = centres[obj] + rng.standard_normal((len(obj), d)).astype(np.float32) * 0.3
            f[rng.random(len(obj)) < 0.05] = 0.0                  # voxels no frame saw
            feats.append(f.astype(np.float32))
        path = os.path.join(out_dir, f"scene{seed + k:04d}_00.npz")
        np.savez(path, coords=coords, feats_2d=feats[0], feats_3d=feats[1], segment_ids=seg,
                 seg_connectivity=np.asarray(sc["segment_connectivity"], np.int64),
                 full_res_coords=sc["xyz"].astype(np.float32), voxel_size=0.02)
        paths.append(path)
    return paths


def process_scene(path, out_dir, device, ncut_args):
    """-> (scene name, number of masks, seconds) or None when the scene was already processed."""
    from unscene3d_amd.pseudo_masks import ncut, pipeline
    name = os.path.splitext(os.path.basename(path))[0]
    cloud_file = os.path.join(out_dir, f"{name}_cloud.npy")
    if os.path.exists(cloud_file):                                  # reference :550-551
        return None
    t0 = time.perf_counter()
    z = np.load(path)
    coords = torch.from_numpy(z["coords"].astype(np.int32)).to(device)
    segment_ids = torch.from_numpy(z["segment_ids"].astype(np.int64)).to(device)
    conn = torch.from_numpy(z["seg_connectivity"].astype(np.int64))
    voxel_size = float(z["voxel_size"]) if "voxel_size" in z.files else 0.02
    mods = [torch.from_numpy(z[k]).to(device) for k in ("feats_2d", "feats_3d") if k in z.files]
    if not mods:
        raise RuntimeError(f"{path}: needs feats_2d and/or feats_3d")
    agg, unique_segments = [], None
    for f in mods:                                                  # reference :350-402, per modality
        a, unique_segments = ncut.aggregate_features(f, segment_ids, conn, aggregation_mode="mean")
        agg.append(a)
    feats = tuple(agg) if len(agg) == 2 else agg[0]
    bip = ncut.unscene3d(feats, unique_segments.cpu(), conn, **ncut_args)            # bool[K, S]
    # segment level -> voxel level (reference :591-599) -> full resolution (reference :649-655)
    inv = torch.unique(segment_ids, return_inverse=True)[1].cpu().numpy()
    voxel_masks = np.asarray(bip).T[inv] if len(bip) else np.zeros((len(inv), 0), bool)
    full = z["full_res_coords"].astype(np.float32)
    _, all_masks = pipeline.masks_to_full_resolution(coords, full, voxel_size, segment_ids.cpu().numpy(), voxel_masks)
    os.makedirs(out_dir, exist_ok=True)
    np.save(os.path.join(out_dir, f"{name}_masks.npy"), all_masks)
    np.save(cloud_file, full.astype(np.single))                     # written last: its presence marks "done"
    return name, int(all_masks.shape[1]), time.perf_counter() - t0


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--scenes", help="directory with one .npz per scene")
    ap.add_argument("--synthetic", type=int, default=0, help="write N synthetic scene files to OUT/scenes and process them")
    ap.add_argument("--out", required=True)
    ap.add_argument("--affinity-tau", type=float, default=0.6)
    ap.add_argument("--max-instances", type=int, default=20)
    ap.add_argument("--min-segment-size", type=int, default=4)
    ap.add_argument("--separation-mode", default="max")
    ap.add_argument("--max-extent-ratio", type=float, default=0.8)
    args = ap.parse_args(argv)
    from unscene3d_amd.pseudo_masks.driver import scene_shard

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if args.synthetic:
        sdir = os.path.join(args.out, "scenes")
        if rank == 0:
            synthetic_scene_files(args.synthetic, sdir)
        else:
            while len(glob.glob(os.path.join(sdir, "*.npz"))) < args.synthetic:
                time.sleep(0.2)
        args.scenes = sdir
    if not args.scenes:
        ap.error("--scenes DIR or --synthetic N")
    files = sorted(glob.glob(os.path.join(args.scenes, "*.npz")))
    mine = scene_shard(len(files), rank, world)
    ncut_args = dict(affinity_tau=args.affinity_tau, max_number_of_instances=args.max_instances,
                     min_segment_size=args.min_segment_size, separation_mode=args.separation_mode,
                     max_extent_ratio=args.max_extent_ratio)
    done = 0
    for i in mine:
        r = process_scene(files[i], args.out, device, ncut_args)
        if r is None:
            print(f"[rank {rank}] scene already processed: {os.path.basename(files[i])}", flush=True)
        else:
            done += 1
            print(f"[rank {rank}] {r[0]}: {r[1]} masks in {r[2]:.2f} s", flush=True)
    print(f"[rank {rank}/{world}] {done} of {len(mine)} scenes of this shard processed ({len(files)} in the list)", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
