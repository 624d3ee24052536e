"""Distributed helpers (reference models/misc.py:106-111, detectron2.utils.comm.get_world_size)."""
import torch.distributed as dist


def is_dist_avail_and_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1
