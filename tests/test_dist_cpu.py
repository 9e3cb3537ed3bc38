"""CPU suite, part 4: the multi-GPU sharding logic under torch.distributed (gloo, world_size 2)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakePipeline:
    """Duck-types Pipeline for the sharding logic (no GPU): 'recognises' an image as its shape and
    the padded size it was given."""

    scale, max_size = 2, 2048

    def _plan(self, shapes):
        import keras_ocr_amd

        return keras_ocr_amd.pipeline.Pipeline._plan(self, shapes)

    def recognize_raw(self, images, hmax, wmax, detection_kwargs=None, recognition_kwargs=None):
        # image i yields (i % 3) boxes whose coordinates and label rows encode what the rank saw
        groups, rows = [], []
        for im in images:
            n = im.shape[0] % 3
            groups.append(np.full((n, 4, 2), im.shape[0], np.float32) if n else np.array([]))
            for j in range(n):
                row = np.full(48, -1, np.int32)
                row[:4] = [im.shape[0], im.shape[1], hmax, wmax]
                row[4] = j
                rows.append(row)
        return groups, (np.array(rows, np.int32) if rows else np.zeros((0, 48), np.int32))

    def assemble(self, box_groups, labels):
        out, pos = [], 0
        for boxes in box_groups:
            out.append([(tuple(int(v) for v in labels[pos + j, :5]), boxes[j]) for j in range(len(boxes))])
            pos += len(boxes)
        return out


def _worker(rank, world, port, n_images, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    import keras_ocr_amd

    r, w = keras_ocr_amd.dist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    images = [np.zeros((10 + i, 20 + 2 * i, 3), np.uint8) for i in range(n_images)]
    sp = keras_ocr_amd.dist.ShardedPipeline(_FakePipeline())
    out = sp.recognize(images)
    q.put((rank, [[(t, float(b[0, 0])) for t, b in o] for o in out], keras_ocr_amd.dist.shard_bounds(n_images, world, rank),
           keras_ocr_amd.dist.ranks_seen()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [5, 2, 1, 7])
def test_sharded_recognize_gloo_world2(n_images):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_images, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    # every rank returns the SAME full list, in input order, padded to the WHOLE batch's size
    hmax, wmax = 2 * (10 + n_images - 1), 2 * (20 + 2 * (n_images - 1))
    want = [[((10 + i, 20 + 2 * i, hmax, wmax, j), float(10 + i)) for j in range((10 + i) % 3)] for i in range(n_images)]
    assert res[0][1] == want and res[1][1] == want
    assert res[0][3] == 2 and res[1][3] == 2  # both ranks took part in the collectives
    # contiguous blocks of ceil(n/2)
    per = -(-n_images // 2)
    assert res[0][2] == (0, min(per, n_images)) and res[1][2] == (min(per, n_images), n_images)


def test_shard_bounds_cover_everything():
    import keras_ocr_amd

    for n in range(0, 40):
        for w in (1, 2, 3, 8):
            spans = [keras_ocr_amd.dist.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


# ---------------------------------------------------------------------------------------------------
# the REAL Pipeline / Detector / Recognizer classes over a context mocked at the C-ABI seam (Context.pipeline)
# ---------------------------------------------------------------------------------------------------
class _MockContext:
    """Stands in for keras_ocr_amd._lib.Context: `pipeline()` is the ctypes wrapper of kocr_pipeline.  It returns what
    the library would: per image (n_i,4,2) float32 boxes in DETECTOR-INPUT pixels and (sum n_i, 48) label rows -- here
    derived from the arguments, so that the test can tell which rank processed which image with which padded size."""

    def pipeline(self, images, hs, ws, dhs, dws, hmax, wmax, micro_batch=0, on_device=False, **kw):
        groups, rows = [], []
        for im, h, w, dh, dw in zip(images, hs, ws, dhs, dws):
            n = int(h) % 3
            boxes = np.zeros((n, 4, 2), np.float32)
            for j in range(n):
                boxes[j] = np.array([[0, 0], [dw, 0], [dw, dh], [0, dh]], np.float32) + j
                row = np.full(48, -1, np.int32)
                row[:3] = [int(h) % 36, hmax % 36, j]   # decoded with the default alphabet below
                rows.append(row)
            groups.append(boxes if n else np.array([]))
        return groups, (np.array(rows, np.int32) if rows else np.zeros((0, 48), np.int32))


def _real_pipeline():
    import keras_ocr_amd as k

    ctx = _MockContext()
    det = object.__new__(k.detection.Detector)
    rec = object.__new__(k.recognition.Recognizer)
    det._ctx = rec._ctx = ctx  # pylint: disable=protected-access
    rec.alphabet = k.recognition.DEFAULT_ALPHABET
    rec.blank_label_idx = len(rec.alphabet)
    return k.pipeline.Pipeline(detector=det, recognizer=rec, scale=2, max_size=2048)


def _worker_real(rank, world, port, n_images, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    import keras_ocr_amd

    keras_ocr_amd.dist.init_from_env(backend="gloo")
    images = [np.zeros((10 + i, 20 + 2 * i, 3), np.uint8) for i in range(n_images)]
    out = keras_ocr_amd.dist.ShardedPipeline(_real_pipeline()).recognize(images)
    q.put((rank, [[(t, b.tolist()) for t, b in page] for page in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_real_pipeline_classes_over_mocked_abi_gloo_world2():
    """ShardedPipeline(real Pipeline).recognize on 2 ranks == the same Pipeline.recognize in one process: scale rule,
    adjust_boxes, string assembly and the packed all-gather all run for real; only kocr_pipeline is mocked."""
    n_images = 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real, args=(r, 2, port, n_images, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    images = [np.zeros((10 + i, 20 + 2 * i, 3), np.uint8) for i in range(n_images)]
    single = [[(t, b.tolist()) for t, b in page] for page in _real_pipeline().recognize(images)]
    assert res[0][1] == single and res[1][1] == single
    # image 1 (height 11 -> 2 boxes): boxes come back in INPUT pixels (detector-input / scale 2), strings decoded
    assert len(single[1]) == 2 and single[1][0][1][2] == [20 + 2, 11.0]
    alphabet = "0123456789abcdefghijklmnopqrstuvwxyz"
    hmax = 2 * (10 + n_images - 1)
    assert single[1][0][0] == alphabet[11] + alphabet[hmax % 36] + alphabet[0]


def test_packed_payload_bytes():
    """Payload of the three result all-gathers: counts + cap x 8 f32 + cap x 48 i32 per rank."""
    from keras_ocr_amd import dist as kd

    p = kd.packed_payload_bytes(32, 1128)
    assert p == {"counts": 136, "boxes": 1128 * 32, "labels": 1128 * 192, "total": 136 + 1128 * 224}
    assert kd.packed_payload_bytes(4, 0)["total"] == 24 + 224


class _FailingPipeline(_FakePipeline):
    """recognize_raw raises on the rank whose shard contains the image of height 13 (a data-dependent failure such as the
    reference's IndexError at detection.py:272)."""

    def recognize_raw(self, images, hmax, wmax, detection_kwargs=None, recognition_kwargs=None):
        if any(im.shape[0] == 13 for im in images):
            raise IndexError("list index out of range (empty contour list)")
        return super().recognize_raw(images, hmax, wmax, detection_kwargs, recognition_kwargs)


def _worker_fail(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    import keras_ocr_amd

    keras_ocr_amd.dist.init_from_env(backend="gloo")
    images = [np.zeros((10 + i, 20, 3), np.uint8) for i in range(6)]   # height 13 = image 3 -> rank 1's shard
    sp = keras_ocr_amd.dist.ShardedPipeline(_FailingPipeline())
    try:
        sp.recognize(images)
        q.put((rank, "no error", None))
    except keras_ocr_amd.dist.ShardError as e:
        q.put((rank, str(e), type(e.__cause__).__name__ if e.__cause__ is not None else None))
    # the group is still usable: no rank was left behind in a collective
    ok = keras_ocr_amd.dist.ShardedPipeline(_FakePipeline()).recognize(images[:3])
    q.put((rank, "after", len(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_failure_on_one_rank_raises_on_every_rank_gloo_world2():
    """a data-dependent failure of one rank's shard must not leave the other ranks blocked in the box / label
    all-gathers: the counts exchange carries a status flag and every rank raises ShardError after the FIRST collective."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fail, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(4))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    errs = {r: (m, c) for r, m, c in res if m != "after"}
    assert "rank(s) [1]" in errs[0][0] and errs[0][1] is None            # rank 0 learns of it through the status slot
    assert "IndexError" in errs[0][0]                                    # ... including WHAT was raised (status code)
    assert "rank(s) [1]" in errs[1][0] and "IndexError" in errs[1][0] and errs[1][1] == "IndexError"
    assert [(r, n) for r, m, n in res if m == "after"] == [(0, 3), (1, 3)]


def test_stagewise_path_pads_to_the_imposed_size(monkeypatch):
    """duck-typed stages (no shared libkocr context) take the stage-wise path; a sharded call must still pad
    to the WHOLE batch's size, not to the shard's own maximum."""
    import keras_ocr_amd as k

    seen = {}

    class Det:
        def detect(self, images, **kw):
            seen["shape"] = images.shape
            return [np.array([])] * len(images)

    class Rec:
        alphabet = k.recognition.DEFAULT_ALPHABET

        def recognize_from_boxes(self, images, box_groups, **kw):
            return [[] for _ in images]

    # tools.resize_image runs on the GPU; the padding rule under test does not depend on its pixels
    monkeypatch.setattr(k.tools, "resize_image", lambda image, max_scale, max_size: (
        np.zeros((image.shape[0] * max_scale, image.shape[1] * max_scale, 3), np.uint8), max_scale))
    pipe = k.pipeline.Pipeline(detector=Det(), recognizer=Rec(), scale=2)
    pipe.recognize_raw([np.zeros((10, 12, 3), np.uint8)], hmax=64, wmax=80)
    assert seen["shape"] == (1, 64, 80, 3)
    pipe.recognize_raw([np.zeros((10, 12, 3), np.uint8)])
    assert seen["shape"] == (1, 20, 24, 3)


# ---------------------------------------------------------------------------------------------------
# the batch starts on ONE rank and is scattered before the local chains (recognize_scattered)
# ---------------------------------------------------------------------------------------------------
def _worker_scatter(rank, world, port, n_images, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import keras_ocr_amd

    keras_ocr_amd.dist.init_from_env(backend="gloo")
    h, w = 13, 24
    batch = None
    if rank == 1:   # the source is NOT rank 0: nothing may depend on that
        arr = np.zeros((n_images, h, w, 3), np.uint8)
        for i in range(n_images):
            arr[i] = i + 1
        batch = torch.from_numpy(arr)
    timing = {}
    seen = []

    class Spy(_FakePipeline):
        def recognize_raw(self, images, hmax, wmax, detection_kwargs=None, recognition_kwargs=None):
            seen.extend(int(im[0, 0, 0]) for im in images)   # which pages reached this rank (their fill value)
            return super().recognize_raw(images, hmax, wmax, detection_kwargs, recognition_kwargs)

    out = keras_ocr_amd.dist.ShardedPipeline(Spy()).recognize_scattered(batch, n_images, h, w, src_rank=1, timing=timing)
    q.put((rank, seen, len(out), timing.get("scatter_bytes_sent"), timing.get("scatter_s", 0) > 0, "gather_s" in timing))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [5, 4, 1])
def test_scatter_from_one_rank_gloo_world2(n_images):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_scatter, args=(r, 2, port, n_images, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    per = -(-n_images // 2)
    assert res[0][1] == list(range(1, min(per, n_images) + 1))                 # rank 0 got pages 1..per from rank 1
    assert res[1][1] == list(range(per + 1, n_images + 1))                     # rank 1 kept the tail (zero padding unused)
    assert res[0][2] == n_images and res[1][2] == n_images                     # everybody ends with the whole result
    assert res[0][3] == 0 and res[1][3] == per * 13 * 24 * 3                   # only the source sends: one block per peer
    assert all(r[4] and r[5] for r in res)


def _worker_scatter_bad_source(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import keras_ocr_amd

    keras_ocr_amd.dist.init_from_env(backend="gloo")
    sp = keras_ocr_amd.dist.ShardedPipeline(_FakePipeline())
    # the source (rank 0) holds a batch of the WRONG shape: it must not raise alone while rank 1 sits in the scatter
    batch = torch.zeros((3, 9, 9, 3), dtype=torch.uint8) if rank == 0 else None
    try:
        sp.recognize_scattered(batch, 4, 13, 24, src_rank=0)
        q.put((rank, "no error", None))
    except keras_ocr_amd.dist.ShardError as e:
        q.put((rank, str(e), type(e.__cause__).__name__ if e.__cause__ else None))
    # the group is still usable: a well-formed call right after it
    good = torch.zeros((4, 13, 24, 3), dtype=torch.uint8) if rank == 0 else None
    out = sp.recognize_scattered(good, 4, 13, 24, src_rank=0)
    q.put((rank, "after", len(out)))
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_source_failure_raises_on_every_rank_gloo_world2():
    """a bad / missing batch on the SOURCE rank of recognize_scattered used to raise there before the scatter,
    leaving every other rank blocked in it.  Now the source announces the failure (one int broadcast), nobody enters the
    scatter, and every rank raises ShardError naming rank 0 and the ValueError."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_scatter_bad_source, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    errs = {r: (m, c) for r, m, c in res if m != "after"}
    assert set(errs) == {0, 1}
    for r in (0, 1):
        assert "rank(s) [0]" in errs[r][0] and "ValueError" in errs[r][0], errs
    assert errs[0][1] == "ValueError" and errs[1][1] is None
    assert sorted((r, n) for r, m, n in res if m == "after") == [(0, 4), (1, 4)]


def test_device_side_packing_equals_host_packing():
    """dist._pack_tensors (what the RCCL path runs on HBM tensors, round 5) on host tensors: from the per-image form
    kocr_pipeline leaves behind -- counts, boxes [n][cap][8] with undefined rows behind each image's count, label rows -- to
    the packed (cap, 8) / (cap, 48) payload of gather_packed, identical to the host path's concatenation."""
    import torch
    from keras_ocr_amd import dist as kd

    rng = np.random.default_rng(3)
    counts = np.array([3, 0, 5, 1], np.int32)
    cap_local, m = 6, int(counts.sum())
    boxes = rng.random((4, cap_local, 8)).astype(np.float32)          # rows >= counts[i]: garbage that must not travel
    labels = rng.integers(-1, 36, (m, 48)).astype(np.int32)
    b, l = kd._pack_tensors(torch.from_numpy(counts), torch.from_numpy(boxes), torch.from_numpy(labels), 12, torch.device("cpu"))
    want = np.concatenate([boxes[i, :c] for i, c in enumerate(counts) if c])
    assert b.shape == (12, 8) and l.shape == (12, 48)
    assert np.array_equal(b[:m].numpy(), want) and not b[m:].any()
    assert np.array_equal(l[:m].numpy(), labels) and (l[m:].numpy() == -1).all()
    b0, l0 = kd._pack_tensors(torch.zeros(2, dtype=torch.int32), torch.zeros((2, 4, 8)), None, 1, torch.device("cpu"))
    assert not b0.any() and (l0.numpy() == -1).all()                   # a rank without boxes
