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


LABEL_WIDTH = 48  # CTC rows: 50 time-steps minus the 2 discarded (recognition.py:175-182, 328)


def init_from_env(backend=None, force=False):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun's contract).
    ``force`` creates the process group even for a single rank, so that the RCCL path is the one that
    runs (and is exercised) at N = 1 too."""
    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not force:
        return 0, 1
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def shard_bounds(n_items, world_size, rank):
    """Contiguous block [start, end) of ceil(n/world) items for ``rank``."""
    per = -(-n_items // world_size) if n_items else 0
    start = min(rank * per, n_items)
    return start, min(start + per, n_items)


def _comm_device(group=None):
    """Tensors handed to a collective live where the backend wants them: HBM for nccl (= RCCL), host for gloo."""
    import torch
    import torch.distributed as dist

    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def ranks_seen(group=None):
    """All-reduce of 1 over the group: the number of ranks that really took part (1 without a group)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return 1
    one = torch.ones(1, dtype=torch.int32, device=_comm_device(group))
    dist.all_reduce(one, op=dist.ReduceOp.SUM, group=group)
    return int(one.item())


def packed_payload_bytes(per_rank_images, cap):
    """Bytes ONE rank contributes to the three all-gathers of `gather_packed` (counts, boxes, label rows)."""
    cap = max(int(cap), 1)
    return {"counts": 4 * (per_rank_images + 2), "boxes": cap * 8 * 4, "labels": cap * LABEL_WIDTH * 4,
            "total": 4 * (per_rank_images + 2) + cap * 8 * 4 + cap * LABEL_WIDTH * 4}


class ShardError(RuntimeError):
    """Raised on EVERY rank when the local chain of at least one rank failed.  The message names the failing rank(s) and
    the KIND of exception each raised (exchanged as a small code in the status slot of the counts all-gather: the
    reference's IndexError at detection.py:272 / ZeroDivisionError at tools.py:95, an AssertionError, a libkocr error, or
    "Exception"), also available as ``failed_ranks`` / ``kinds`` ({rank: name}); the failing rank chains its own
    exception as ``__cause__``.  A sharded caller therefore sees ShardError where the single-process call raises the
    reference's own exception type."""
    failed_ranks = ()
    kinds = {}


_STATUS_NAMES = {1: "Exception", 2: "IndexError", 3: "ZeroDivisionError", 4: "AssertionError", 5: "KocrError", 6: "ValueError",
                 7: "TypeError"}


def _status_of(error):
    """0 = ok, else a code naming the exception type (anything unlisted travels as 1 = "Exception")."""
    if error is None:
        return 0
    for code, name in _STATUS_NAMES.items():
        if code != 1 and any(c.__name__ == name for c in type(error).__mro__):
            return code
    return 1


class _DeviceArray:
    """A device buffer libkocr owns, as something ``torch.as_tensor(..., device="cuda")`` takes without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _pack_tensors(counts, boxes, labels, cap, dev):
    """(cap, 8) box tensor and (cap, 48) label tensor from the per-image form kocr_pipeline produces -- counts [n], boxes
    [n][cap_local][8] (rows >= counts[i] of image i undefined), label rows [m][48] -- on whatever device the inputs live:
    a boolean-mask gather by the counts, image-major, which is the order of the label rows."""
    import torch

    n, cl = boxes.shape[0], boxes.shape[1]
    m = int(labels.shape[0]) if labels is not None else 0
    b = torch.zeros((cap, 8), dtype=torch.float32, device=dev)
    l = torch.full((cap, LABEL_WIDTH), -1, dtype=torch.int32, device=dev)
    if m:
        mask = torch.arange(cl, device=dev)[None, :] < counts[:, None]
        b[:m] = boxes.reshape(n, cl, 8)[mask]
        l[:m] = labels
    return b, l


def _pack_on_device(dev_res, cap, dev):
    """The packed tensors of gather_packed built IN HBM from what kocr_pipeline left there: the payload RCCL moves never
    visits the host."""
    import torch

    n, cl, m = dev_res["n"], dev_res["cap"], dev_res["m"]
    counts = torch.as_tensor(_DeviceArray(dev_res["counts"], (n,), "<i4"), device=dev)
    boxes = torch.as_tensor(_DeviceArray(dev_res["boxes"], (n, cl, 8), "<f4"), device=dev)
    labels = torch.as_tensor(_DeviceArray(dev_res["labels"], (m, LABEL_WIDTH), "<i4"), device=dev) if m else None
    return _pack_tensors(counts, boxes, labels, cap, dev)


def gather_packed(box_groups, labels, per_rank_images, group=None, error=None, timing=None, device_results=None,
                  device_scale=None):
    """all-gather of the per-image box counts, then of fixed-capacity packed results.

    Every rank contributes ``per_rank_images`` count slots (its shard, zero padded), a (cap, 8) float32
    box tensor and a (cap, 48) int32 label tensor, where cap = the largest crop count of any rank (known
    after the counts exchange).  Three collectives, no pickling; on the nccl backend the tensors stay in
    HBM and travel over xGMI.  Returns (box_groups, labels) of the whole batch in image order.

    ``device_scale`` (nccl backend, the SAME value on every rank or None on every rank) switches the box / label gathers to
    device-originated payloads: a rank that had work passes ``device_results`` (``Pipeline.recognize_device_raw``: where its
    results still lie in HBM, boxes in detector-input pixels) and its packed tensors are built on the device from those
    buffers (`_pack_on_device`) -- nothing is copied up from the host; every rank then divides the gathered boxes by the
    scale exactly as the host path does before its gather (``tools.adjust_boxes``: the same float32 product), so the result
    is identical bit for bit.

    ``error``: the exception the local chain raised, if any.  The counts exchange carries a status slot, so a
    data-dependent failure on one rank (the reference raises IndexError at detection.py:272 on an empty contour
    list, ZeroDivisionError at tools.py:95) makes EVERY rank raise ``ShardError`` after the first collective instead
    of leaving the others blocked in the box / label gathers.  ``timing`` (a dict) receives ``gather_s``.
    """
    import time
    import torch
    import torch.distributed as dist

    if error is not None:
        box_groups, labels = [], np.zeros((0, LABEL_WIDTH), np.int32)
    counts_local = [len(b) for b in box_groups]
    m_local = int(sum(counts_local))
    labels = np.asarray(labels, np.int32).reshape(m_local, LABEL_WIDTH)
    boxes_local = (np.concatenate([np.asarray(b, np.float32).reshape(-1, 8) for b in box_groups if len(b)])
                   if m_local else np.zeros((0, 8), np.float32))
    if not (dist.is_available() and dist.is_initialized()):
        if error is not None:
            raise error
        return [np.asarray(b) for b in box_groups], labels
    t0 = time.perf_counter()
    world = dist.get_world_size(group)
    dev = _comm_device(group)
    if error is None and device_scale is not None and dev.type == "cuda" and m_local and \
            (not device_results or device_results.get("m") != m_local):
        # found BEFORE the first collective and sent through the status slot: raised locally after the counts exchange it would
        # leave the other ranks blocked in the box / label gathers
        error = RuntimeError("gather_packed: device-originated gather without this rank's device results")
        box_groups, labels = [], np.zeros((0, LABEL_WIDTH), np.int32)
        counts_local, m_local, boxes_local = [], 0, np.zeros((0, 8), np.float32)
    # 1. counts: per_rank_images + 2 ints per rank (slot -2 = number of images this rank really had, slot -1 = status)
    c = torch.zeros(per_rank_images + 2, dtype=torch.int32)
    c[:len(counts_local)] = torch.tensor(counts_local, dtype=torch.int32)
    c[-2] = len(counts_local)
    c[-1] = _status_of(error)
    c = c.to(dev)
    all_c = torch.empty(world * (per_rank_images + 2), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(all_c, c, group=group)
    all_c = all_c.cpu().numpy().reshape(world, per_rank_images + 2)
    failed = [r for r in range(world) if all_c[r, -1] != 0]
    if failed:
        kinds = sorted({_STATUS_NAMES.get(int(all_c[r, -1]), "Exception") for r in failed})
        msg = f"sharded recognize failed on rank(s) {failed} ({', '.join(kinds)})"
        exc = ShardError(f"{msg}: {type(error).__name__}: {error}" if error is not None else msg)
        exc.failed_ranks = failed
        exc.kinds = {r: _STATUS_NAMES.get(int(all_c[r, -1]), "Exception") for r in failed}
        if error is not None:
            raise exc from error
        raise exc
    cap = max(int(all_c[:, :-2].sum(axis=1).max()), 1)
    # 2. + 3. packed boxes and label rows, capacity = the busiest rank's crop count
    # every rank must take the same branch below: the scale travels with the decision (all ranks of a device-resident batch
    # have it or none has: recognize_device / recognize_scattered pass it on every rank that had work)
    on_device = device_scale is not None and dev.type == "cuda"
    if on_device and m_local:
        b, l = _pack_on_device(device_results, cap, dev)
    elif on_device:
        b = torch.zeros((cap, 8), dtype=torch.float32, device=dev)
        l = torch.full((cap, LABEL_WIDTH), -1, dtype=torch.int32, device=dev)
    else:
        b = torch.zeros((cap, 8), dtype=torch.float32)
        b[:m_local] = torch.from_numpy(boxes_local)
        l = torch.full((cap, LABEL_WIDTH), -1, dtype=torch.int32)
        l[:m_local] = torch.from_numpy(labels)
        b, l = b.to(dev), l.to(dev)
    all_b = torch.empty((world * cap, 8), dtype=torch.float32, device=dev)
    all_l = torch.empty((world * cap, LABEL_WIDTH), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(all_b, b, group=group)
    dist.all_gather_into_tensor(all_l, l, group=group)
    all_b = all_b.cpu().numpy().reshape(world, cap, 4, 2)
    all_l = all_l.cpu().numpy().reshape(world, cap, LABEL_WIDTH)
    if on_device and device_scale != 1:
        all_b = all_b * (1 / device_scale)  # tools.adjust_boxes (pipeline.py:66-71), as Pipeline._adjust does on the host path
    if timing is not None:
        timing["gather_packed_on_device"] = on_device
    if timing is not None:
        timing["gather_s"] = timing.get("gather_s", 0.0) + (time.perf_counter() - t0)
        timing["gather_payload_bytes_per_rank"] = packed_payload_bytes(per_rank_images, cap)["total"]
    out_boxes, out_labels = [], []
    for r in range(world):
        pos = 0
        for i in range(int(all_c[r, -2])):
            n = int(all_c[r, i])
            # an image without boxes is np.array([]) in the reference (detection.py:286)
            out_boxes.append(all_b[r, pos:pos + n].copy() if n else np.zeros((0,), np.float32))
            pos += n
        out_labels.append(all_l[r, :pos])
    return out_boxes, np.concatenate(out_labels) if out_labels else np.zeros((0, LABEL_WIDTH), np.int32)


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

    def recognize(self, images, detection_kwargs=None, recognition_kwargs=None, timing=None):
        """``timing`` (optional dict) receives ``gather_s``: the time spent in the three result all-gathers."""
        from . import tools

        if not isinstance(images, np.ndarray):
            images = [tools.read(image) for image in images]
        images = list(images)
        if not images:
            return []
        rank, world = self._rank_world()
        # the padded size comes from the WHOLE batch (pipeline.py:48-57), not from the shard
        _, _, _, hmax, wmax = self.pipeline._plan([im.shape for im in images])  # pylint: disable=protected-access
        start, end = shard_bounds(len(images), world, rank)
        return self._run_shard(lambda: self.pipeline.recognize_raw(images[start:end], hmax, wmax, detection_kwargs,
                                                                   recognition_kwargs),
                               end > start, -(-len(images) // world), timing)

    def recognize_device(self, d_ptr, n_total, h, w, detection_kwargs=None, timing=None):
        """The same for a batch whose images are already resident in THIS rank's HBM (``d_ptr`` = device pointer of this
        rank's contiguous block of images of the (n_total, h, w, 3) uint8 batch): every image has the same size, so the
        whole batch's padded size is the shard's."""
        rank, world = self._rank_world()
        start, end = shard_bounds(n_total, world, rank)
        dev_res, scale = self._device_gather(n_total, h, w)
        return self._run_shard(lambda: self.pipeline.recognize_device_raw(d_ptr, end - start, h, w, detection_kwargs, *(() if dev_res is None else (dev_res,))),
                               end > start, -(-n_total // world), timing, dev_res, scale)

    def _device_gather(self, n_total, h, w):
        """(dict to receive the device results, scale) when the result gathers can be device-originated -- RCCL group, the
        real Pipeline (one image size, hence one scale, known on every rank from the plan) -- else (None, None)."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized() and dist.get_backend(self.group) == "nccl" and n_total > 0 and
                hasattr(self.pipeline, "recognize_device_raw") and getattr(getattr(self.pipeline, "detector", None), "_ctx", None) is not None):
            return None, None
        scales = self.pipeline._plan([(h, w, 3)] * n_total)[0]  # pylint: disable=protected-access
        return {}, scales[0]

    def recognize_scattered(self, batch, n_total, h, w, src_rank=0, detection_kwargs=None, timing=None):
        """the WHOLE (n_total, h, w, 3) uint8 batch starts on ONE rank -- ``batch`` is a torch tensor
        in ``src_rank``'s HBM (a host tensor under gloo) and is ignored (may be None) on the other ranks.  The raw pages
        are scattered BEFORE the resize (the smallest form of the data: 1.81 GB for BASELINE configs[4], 7/8 of it leaves
        rank 0 over its seven xGMI links) with one ``torch.distributed.scatter`` of equal ceil(n/world) blocks (RCCL
        implements it as grouped send/recv; the last block is zero padded), then every rank runs the local chain on its
        block and the results are all-gathered as in ``recognize_device``.  ``timing`` receives ``scatter_s`` (the
        collective plus the wait for it) and ``scatter_bytes_sent`` next to ``gather_s``."""
        import time
        import torch
        import torch.distributed as dist

        rank, world = self._rank_world()
        per = -(-n_total // world) if n_total else 0
        start, end = shard_bounds(n_total, world, rank)
        distributed = dist.is_available() and dist.is_initialized()
        t0 = time.perf_counter()
        src_error = None
        if not distributed:
            mine = batch
            dev = None
        else:
            dev = _comm_device(self.group)
            # The source rank may fail BEFORE the collective (a missing / mis-shaped batch, an allocation error while it
            # cuts the blocks): every other rank would already sit in the scatter and hang until the backend's timeout.
            # So the source first tells everybody whether the scatter will happen (one int32 broadcast); on failure no
            # rank enters it, and the error travels through the status slot of the counts all-gather like any failure of a
            # local chain: ShardError on every rank.
            chunks = None
            if rank == src_rank:
                try:
                    if batch is None or tuple(batch.shape) != (n_total, h, w, 3) or batch.dtype != torch.uint8:
                        raise ValueError("recognize_scattered: the source rank needs the (n_total, h, w, 3) uint8 batch")
                    batch = batch.to(dev)
                    chunks = []
                    for r in range(world):
                        c = batch[r * per:min((r + 1) * per, n_total)]
                        if c.shape[0] < per:  # equal blocks for the collective: the tail is zero padded (and never processed)
                            c = torch.cat([c, torch.zeros((per - c.shape[0], h, w, 3), dtype=torch.uint8, device=dev)])
                        chunks.append(c.contiguous())
                except Exception as e:  # noqa: BLE001 -- exchanged below, never raised on this rank alone
                    src_error, chunks = e, None
            go = torch.tensor([0 if src_error is not None else 1], dtype=torch.int32, device=dev)
            dist.broadcast(go, src=src_rank, group=self.group)
            mine = None
            if int(go.item()) == 1:
                mine = torch.empty((per, h, w, 3), dtype=torch.uint8, device=dev)
                dist.scatter(mine, scatter_list=chunks, src=src_rank, group=self.group)
                if dev.type == "cuda":
                    torch.cuda.synchronize()
        if timing is not None:
            timing["scatter_s"] = timing.get("scatter_s", 0.0) + (time.perf_counter() - t0)
            timing["scatter_bytes_sent"] = (per * (world - 1) * h * w * 3) if (distributed and rank == src_rank and src_error is None) else 0
        n_mine = end - start

        dev_res, scale = self._device_gather(n_total, h, w)

        def run():
            if src_error is not None:
                raise src_error
            if mine.device.type == "cuda":
                return self.pipeline.recognize_device_raw(mine.data_ptr(), n_mine, h, w, detection_kwargs, *(() if dev_res is None else (dev_res,)))
            # host tensors (gloo): the same call as recognize(); every image has the batch's size
            _, _, _, hmax, wmax = self.pipeline._plan([(h, w, 3)] * n_total)  # pylint: disable=protected-access
            return self.pipeline.recognize_raw(mine[:n_mine].numpy(), hmax, wmax, detection_kwargs, None)

        if distributed and mine is None:  # the source failed: nobody has a block; the source reports why
            return self._run_shard(run, rank == src_rank, per, timing, dev_res, scale)
        return self._run_shard(run, n_mine > 0, per, timing, dev_res, scale)

    def _run_shard(self, run, has_work, per, timing, device_results=None, device_scale=None):
        box_groups, labels, err = [], np.zeros((0, LABEL_WIDTH), np.int32), None
        try:
            if has_work:
                box_groups, labels = run()
                # the packing of gather_packed's preamble, done here so that a malformed local result is exchanged as a
                # status too instead of raising on one rank before the first collective
                np.asarray(labels, np.int32).reshape(sum(len(b) for b in box_groups), LABEL_WIDTH)
                for b in box_groups:
                    if len(b):
                        np.asarray(b, np.float32).reshape(-1, 8)
        except Exception as e:  # noqa: BLE001 -- exchanged as a status flag so that no rank is left in a collective
            box_groups, labels, err = [], np.zeros((0, LABEL_WIDTH), np.int32), e
        box_groups, labels = gather_packed(box_groups, labels, per, self.group, error=err, timing=timing,
                                           device_results=device_results, device_scale=device_scale)
        return self.pipeline.assemble(box_groups, labels)
