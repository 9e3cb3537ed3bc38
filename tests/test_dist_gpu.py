"""The RCCL (backend "nccl") branch of the result gather on a real GPU, world size 1 (the GPU box has one device; the
8-GPU run is the driver's): ``ShardedPipeline(real Pipeline).recognize`` must equal ``Pipeline.recognize`` with the three
``all_gather_into_tensor`` collectives running on HBM tensors (dist.gather_packed), for host arrays and for a batch that
is already resident in HBM (the bench's cfg5_sharded leg).  VERDICT r02 item 2 / SURVEY.md 8(e).1-3."""
import os
import socket

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group():
    import torch
    import torch.distributed as dist
    import keras_ocr_amd

    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    rank, world = keras_ocr_amd.dist.init_from_env(backend="nccl", force=True)
    assert (rank, world) == (0, 1) and dist.get_backend() == "nccl"
    yield
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def pipe(ctx, craft_weights, crnn_weights):
    import keras_ocr_amd
    from oracle import craft as ocraft, tools as otools

    page = synth.text_page(96, 128, 5, seed=21)[None]
    big = np.stack([otools.resize_image(p, 2, 2048)[0] for p in page])
    heat = ocraft.detector_predict(craft_weights, big)
    w = keras_ocr_amd.weights.calibrate_craft_head(craft_weights, heat, text_frac=0.10, link_frac=0.04)
    det = keras_ocr_amd.detection.Detector(weights=w, ctx=ctx)
    rec = keras_ocr_amd.recognition.Recognizer(weights=crnn_weights, ctx=ctx)
    return keras_ocr_amd.pipeline.Pipeline(detector=det, recognizer=rec)


def _same(a, b):
    assert len(a) == len(b)
    for pa, pb in zip(a, b):
        assert [t for t, _ in pa] == [t for t, _ in pb]
        for (_, ba), (_, bb) in zip(pa, pb):
            assert np.array_equal(np.asarray(ba), np.asarray(bb))


def test_sharded_equals_single_process_over_rccl(nccl_group, pipe):
    import keras_ocr_amd

    assert keras_ocr_amd.dist.ranks_seen() == 1  # all-reduce over RCCL
    pages = [synth.text_page(96, 128, 5, seed=s) for s in (21, 22)] + [np.full((64, 80, 3), 255, np.uint8),
                                                                       synth.text_page(80, 112, 4, seed=23)]
    want = pipe.recognize(pages)
    assert sum(len(p) for p in want) > 0
    timing = {}
    got = keras_ocr_amd.dist.ShardedPipeline(pipe).recognize(pages, timing=timing)
    _same(got, want)
    assert timing["gather_s"] > 0 and timing["gather_payload_bytes_per_rank"] > 0


def test_sharded_device_resident_batch_over_rccl(nccl_group, pipe):
    import torch
    import keras_ocr_amd

    pages = np.stack([synth.text_page(96, 128, 5, seed=s) for s in (31, 32, 33)])
    want = pipe.recognize(pages)
    d = torch.from_numpy(pages).cuda()
    timing = {}
    got = keras_ocr_amd.dist.ShardedPipeline(pipe).recognize_device(d.data_ptr(), 3, 96, 128, timing=timing)
    _same(got, want)
    # round 5: the packed tensors RCCL moved were built in HBM from the buffers kocr_pipeline left there (no host staging),
    # and the result is bit-identical to the host-packed one
    assert timing["gather_packed_on_device"] is True


def test_device_results_of_the_last_pipeline_call(nccl_group, pipe, ctx):
    """kocr_pipeline_device_results: the same boxes / counts / label rows kocr_pipeline returned, still resident in HBM."""
    import torch
    from keras_ocr_amd.dist import _DeviceArray

    pages = np.stack([synth.text_page(96, 128, 5, seed=s) for s in (31, 32, 33)])
    d = torch.from_numpy(pages).cuda()
    res = {}
    boxes, labels = pipe.recognize_device_raw(d.data_ptr(), 3, 96, 128, device_results=res)
    assert res["n"] == 3 and res["m"] == labels.shape[0] == sum(len(b) for b in boxes) > 0 and res["scale"] == 2
    counts = torch.as_tensor(_DeviceArray(res["counts"], (3,), "<i4"), device="cuda").cpu().numpy()
    assert list(counts) == [len(b) for b in boxes]
    lab = torch.as_tensor(_DeviceArray(res["labels"], (res["m"], 48), "<i4"), device="cuda").cpu().numpy()
    assert np.array_equal(lab, labels)
    bx = torch.as_tensor(_DeviceArray(res["boxes"], (3, res["cap"], 4, 2), "<f4"), device="cuda").cpu().numpy()
    for i, b in enumerate(boxes):
        if len(b):
            assert np.array_equal(bx[i, :len(b)] * np.float32(0.5), np.asarray(b))
    # nothing resident after a call that reuses the arenas
    ctx.craft_forward(pages[:1])
    with pytest.raises(Exception, match="no kocr_pipeline result"):
        ctx.pipeline_device_results()


def test_gather_packed_on_hbm_tensors_empty_rank(nccl_group):
    """a rank without any box still takes part in all three collectives (cap = 1)"""
    from keras_ocr_amd import dist as kd

    boxes, labels = kd.gather_packed([np.array([]), np.array([])], np.zeros((0, 48), np.int32), 2)
    assert len(boxes) == 2 and all(len(b) == 0 for b in boxes) and labels.shape == (0, 48)


def test_scattered_batch_over_rccl(nccl_group, pipe):
    """the batch starts in ONE rank's HBM and goes through torch.distributed.scatter (RCCL) before the local
    chains; at world 1 the collective degenerates to a device copy but runs through the same code path."""
    import torch
    import keras_ocr_amd

    pages = np.stack([synth.text_page(96, 128, 5, seed=s) for s in (41, 42, 43)])
    want = pipe.recognize(pages)
    timing = {}
    got = keras_ocr_amd.dist.ShardedPipeline(pipe).recognize_scattered(torch.from_numpy(pages).cuda(), 3, 96, 128, src_rank=0,
                                                                       timing=timing)
    _same(got, want)
    assert timing["scatter_s"] > 0 and timing["scatter_bytes_sent"] == 0 and timing["gather_s"] > 0
    assert timing["gather_packed_on_device"] is True
    # a missing batch on the source rank: announced to every rank before the scatter, raised as ShardError everywhere (round 5)
    with pytest.raises(keras_ocr_amd.dist.ShardError, match="ValueError") as ei:
        keras_ocr_amd.dist.ShardedPipeline(pipe).recognize_scattered(None, 3, 96, 128, src_rank=0)
    assert isinstance(ei.value.__cause__, ValueError) and "source rank needs" in str(ei.value.__cause__)
