"""Data-parallel inference over the GPUs of one node: one process per GPU, ``torch.distributed``
(backend ``nccl`` = RCCL over xGMI on ROCm; ``gloo`` on CPU for the tests).

The reference has no multi-GPU inference; images are independent end to end
(pipeline.py:62-75), so the batch is split into contiguous blocks, each rank runs the whole
detect -> crop -> recognise chain on its block, and only the results (a few kB per image: boxes
+ strings) cross ranks.  The one cross-image coupling — ``pad`` to the batch's maximum size
(pipeline.py:48-57) — is resolved BEFORE sharding so that every rank sees the same detector
input size as the single-process call would.  No all-reduce, no data-path collective.
"""
import os

import numpy as np


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun's contract)."""
    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def shard_bounds(n_items, world_size, rank):
    """Contiguous block [start, end) of ceil(n/world) items for ``rank``."""
    per = -(-n_items // world_size) if n_items else 0
    start = min(rank * per, n_items)
    return start, min(start + per, n_items)


def gather_lists(local, group=None):
    """All-gather per-rank result lists (arbitrary picklable objects) and concatenate in rank order."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(local)
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, list(local), group=group)
    return [item for part in out for item in part]


class ShardedPipeline:
    """``Pipeline.recognize`` over all ranks: same arguments, same return value on every rank."""

    def __init__(self, pipeline, group=None):
        self.pipeline = pipeline
        self.group = group

    def _rank_world(self):
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.group), dist.get_world_size(self.group)
        return 0, 1

    def recognize(self, images, detection_kwargs=None, recognition_kwargs=None):
        from . import tools

        if not isinstance(images, np.ndarray):
            images = [tools.read(image) for image in images]
        images = list(images)
        rank, world = self._rank_world()
        # the padded size comes from the WHOLE batch (pipeline.py:48-57), not from the shard
        _, dhs, dws, hmax, wmax = self.pipeline._plan([im.shape for im in images])  # pylint: disable=protected-access
        start, end = shard_bounds(len(images), world, rank)
        local = self.pipeline.recognize_padded(images[start:end], hmax, wmax, detection_kwargs, recognition_kwargs) \
            if end > start else []
        return gather_lists(local, self.group)
