"""Distributed helpers (mirror of the hot-path subset of maskrcnn_benchmark/utils/comm.py:11-45)."""
import torch.distributed as dist


def get_world_size():
    if not dist.is_available() or not dist.is_initialized():
        return 1
    return dist.get_world_size()


def get_rank():
    if not dist.is_available() or not dist.is_initialized():
        return 0
    return dist.get_rank()


def is_main_process():
    return get_rank() == 0


def synchronize():
    if get_world_size() > 1:
        dist.barrier()
