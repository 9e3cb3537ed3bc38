"""CPU suite: weight ingestion -- the readers behind ``Detector(weights=...)`` /
``Recognizer(weights=...)`` and the cache semantics of ``tools.download_and_verify``.

  * ``.pth``: ``torch.save`` of a ``module.``-prefixed state dict with ``num_batches_tracked`` entries, as
    craft_mlt_25k.pth is (detection.py:428-468), read back by ``load_torch_state_dict``;
  * ``.h5``: files written by the REAL h5py (image's second interpreter, tests/golden/make_keras_h5.py) in the layout
    Keras ``save_weights`` produces, read back by ``weights.read_keras_h5`` through the built-in HDF5 reader
    (``keras_ocr_amd/h5lite.py``; h5py is not importable in the main interpreter);
  * cache + sha256 semantics (tools.py:501-530) with a temporary ``KERAS_OCR_CACHE_DIR`` and ``file://`` URLs.
The GPU half (heat-maps / labels identical whichever way the weights arrive) is tests/test_weights_gpu.py.
"""
import os
import subprocess

import numpy as np
import pytest

CONDA_PY = "/opt/conda/bin/python3.9"
HERE = os.path.dirname(os.path.abspath(__file__))


def write_keras_h5(weights, path, kind):
    """npz -> Keras-layout .h5 through real h5py; skips the test when the second interpreter is not there."""
    if not os.path.isfile(CONDA_PY):
        pytest.skip("no /opt/conda interpreter with h5py in this environment")
    npz = str(path) + ".npz"
    np.savez(npz, **weights)
    r = subprocess.run([CONDA_PY, os.path.join(HERE, "golden", "make_keras_h5.py"), npz, str(path), kind],
                       capture_output=True, text=True, check=False)
    if r.returncode != 0 and "No module named 'h5py'" in r.stderr:
        pytest.skip("h5py missing in /opt/conda")
    assert r.returncode == 0, r.stderr[-2000:]
    os.remove(npz)


def write_craft_pth(weights, path):
    import torch

    sd = {}
    for k, v in weights.items():
        sd["module." + k] = torch.from_numpy(np.array(v))
        if k.endswith(".running_var"):
            sd["module." + k[:-len("running_var")] + "num_batches_tracked"] = torch.tensor(25000)
    torch.save(sd, str(path))


def test_pth_state_dict_round_trip(tmp_path, craft_weights):
    import keras_ocr_amd

    p = tmp_path / "craft_mlt_25k.pth"
    write_craft_pth(craft_weights, p)
    got = keras_ocr_amd.detection.load_torch_state_dict(str(p))
    assert not any(k.endswith("num_batches_tracked") for k in got)
    assert {k[len("module."):] for k in got} == set(craft_weights)  # kocr_load_craft strips the prefix (craft.cpp)
    for k, v in craft_weights.items():
        assert got["module." + k].dtype == np.float32 and np.array_equal(got["module." + k], v)


@pytest.mark.parametrize("kind", ["craft", "crnn", "crnn_notop"])
def test_keras_h5_round_trip_through_builtin_reader(tmp_path, kind, craft_weights, crnn_weights):
    import keras_ocr_amd

    w = craft_weights if kind == "craft" else dict(crnn_weights)
    if kind == "crnn_notop":
        w = {k: v for k, v in w.items() if not k.startswith("fc_12")}
    p = tmp_path / f"{kind}.h5"
    write_keras_h5(w, p, kind)
    got = keras_ocr_amd.weights.read_keras_h5(str(p), "craft" if kind == "craft" else "crnn")
    assert set(got) == set(w), set(got) ^ set(w)
    for k, v in w.items():
        assert got[k].shape == v.shape and got[k].dtype == np.float32 and np.array_equal(got[k], v), k


GENERIC_VARIANTS = ["fresh", "busy", "tf20", "fullmodel"]


@pytest.mark.parametrize("variant", GENERIC_VARIANTS)
@pytest.mark.parametrize("kind", ["craft", "crnn", "crnn_notop"])
def test_keras_h5_from_a_generic_keras_saver(tmp_path, kind, variant, craft_weights, crnn_weights):
    """files from a saver that is NOT make_keras_h5.py -- tests/golden/make_keras_h5_generic.py replays the
    reference's layer constructor calls against a stand-in of Keras's automatic naming (process-wide counters: conv2d_7,
    dense_3, model_2, lstm_cell_11 ...; TF 2.0 without the LSTM cell scope; `model.save()` with /model_weights and
    /optimizer_weights) and of `save_weights_to_hdf5_group`.  The reader must not depend on any of those names."""
    import keras_ocr_amd

    if not os.path.isfile(CONDA_PY):
        pytest.skip("no /opt/conda interpreter with h5py in this environment")
    w = craft_weights if kind == "craft" else dict(crnn_weights)
    if kind == "crnn_notop":
        w = {k: v for k, v in w.items() if not k.startswith("fc_12")}
    npz, p = str(tmp_path / "w.npz"), str(tmp_path / f"{kind}_{variant}.h5")
    np.savez(npz, **w)
    r = subprocess.run([CONDA_PY, os.path.join(HERE, "golden", "make_keras_h5_generic.py"), npz, p, kind, variant],
                       capture_output=True, text=True, check=False)
    if r.returncode != 0 and "No module named 'h5py'" in r.stderr:
        pytest.skip("h5py missing in /opt/conda")
    assert r.returncode == 0, r.stderr[-2000:]
    got = keras_ocr_amd.weights.read_keras_h5(p, "craft" if kind == "craft" else "crnn")
    assert set(got) == set(w), set(got) ^ set(w)
    for k, v in w.items():
        assert got[k].shape == v.shape and got[k].dtype == np.float32 and np.array_equal(got[k], v), k


def test_stn_placement_errors_are_explicit(tmp_path, crnn_weights):
    """The unnamed localisation layers are placed by shape: a second tensor of a shape that identifies one of them, or
    an unnamed tensor of an unknown shape, must raise a clear error instead of loading the wrong thing."""
    import keras_ocr_amd
    from keras_ocr_amd import weights as kw

    base = {f"{k.split('/')[0]}/{k}:0": v for k, v in crnn_weights.items() if k.startswith(("conv_1", "fc_9"))}
    stn = {"model/conv2d/kernel:0": crnn_weights["stn_conv_1/kernel"], "model/conv2d/bias:0": crnn_weights["stn_conv_1/bias"]}

    def run(extra, monkey):
        monkey.setattr(kw, "_h5_datasets", lambda path: {**base, **stn, **extra})
        return keras_ocr_amd.weights.read_keras_h5("x.h5", "crnn")

    mp = pytest.MonkeyPatch()
    try:
        assert "stn_conv_1/bias" in run({}, mp)
        with pytest.raises(ValueError, match="ambiguous localisation network"):
            run({"model/conv2d_9/bias:0": np.zeros(16, np.float32)}, mp)
        with pytest.raises(ValueError, match="non-default build_params"):
            run({"model/conv2d_9/bias:0": np.zeros(17, np.float32)}, mp)
        with pytest.raises(ValueError, match="two groups"):
            run({"model_5/dense/bias:0": np.zeros(6, np.float32)}, mp)
    finally:
        mp.undo()


def test_h5lite_reads_the_checked_in_h5py_fixture():
    """tests/golden/tiny_h5py_fixture.h5 was written by the real h5py 3.3 (generator: the heredoc in the commit that added
    it; values are formulas re-evaluated here), so the built-in reader is exercised even where no second interpreter
    exists: scoped variable names, a nested model group, an empty group, big-endian float64, a scalar int64."""
    from keras_ocr_amd import h5lite

    d = h5lite.read_datasets(os.path.join(HERE, "golden", "tiny_h5py_fixture.h5"))
    assert set(d) == {"conv_1/conv_1/kernel:0", "conv_1/conv_1/bias:0", "model_1/conv2d_7/kernel:0", "model_1/dense_3/bias:0",
                      "lstm_10/lstm_10/lstm_cell_11/recurrent_kernel:0", "optimizer_weights/Adam/iter:0"}
    assert np.array_equal(d["conv_1/conv_1/kernel:0"], np.arange(72, dtype=np.float32).reshape(3, 3, 2, 4) / 8 - 1)
    assert np.array_equal(d["conv_1/conv_1/bias:0"], np.linspace(-1, 1, 4, dtype=np.float32))
    assert np.array_equal(d["model_1/conv2d_7/kernel:0"], np.full((5, 5, 1, 2), 0.25, np.float32))
    assert np.array_equal(d["model_1/dense_3/bias:0"], np.array([0.9, 0, 0, 0, 0.9, 0], np.float32))
    rk = d["lstm_10/lstm_10/lstm_cell_11/recurrent_kernel:0"]
    assert rk.dtype == np.float64 and np.array_equal(rk, np.arange(24, dtype=np.float64).reshape(2, 12))
    assert d["optimizer_weights/Adam/iter:0"].shape == () and int(d["optimizer_weights/Adam/iter:0"]) == 25000


def test_h5lite_rejects_what_it_does_not_understand(tmp_path):
    from keras_ocr_amd import h5lite

    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file at all")
    with pytest.raises(ValueError):
        h5lite.read_datasets(str(p))
    if os.path.isfile(CONDA_PY):
        code = ("import h5py, numpy as np, sys\n"
                "f = h5py.File(sys.argv[1], 'w')\n"
                "f.create_dataset('a/b', data=np.arange(100000, dtype='f4'), chunks=(1000,), compression='gzip')\n"
                "f.close()\n")
        r = subprocess.run([CONDA_PY, "-c", code, str(p)], capture_output=True, text=True, check=False)
        if r.returncode == 0:
            with pytest.raises(NotImplementedError):
                h5lite.read_datasets(str(p))
        # a dataset whose datatype message is a reference to a committed (shared) datatype: flag bit 1 of the message
        code = ("import h5py, numpy as np, sys\n"
                "f = h5py.File(sys.argv[1], 'w')\n"
                "f['t'] = np.dtype('<f4')\n"
                "f.create_dataset('a', data=np.arange(10, dtype='f4'), dtype=f['t'])\n"
                "f.close()\n")
        r = subprocess.run([CONDA_PY, "-c", code, str(p)], capture_output=True, text=True, check=False)
        if r.returncode == 0:
            with pytest.raises(NotImplementedError):
                h5lite.read_datasets(str(p))


def test_download_and_verify_cache_semantics(tmp_path, monkeypatch):
    """tools.py:501-530: cached + matching hash -> no fetch; missing or mismatching -> fetch; then verify."""
    import keras_ocr_amd

    tools = keras_ocr_amd.tools
    src = tmp_path / "remote" / "weights.bin"
    src.parent.mkdir()
    src.write_bytes(b"abc" * 1000)
    sha = tools.sha256sum(str(src))
    assert sha == __import__("hashlib").sha256(b"abc" * 1000).hexdigest()
    cache = tmp_path / "cache"
    monkeypatch.setenv("KERAS_OCR_CACHE_DIR", str(cache))
    assert tools.get_default_cache_dir() == str(cache)
    url = "file://" + str(src)
    # 1. not cached: fetched into $KERAS_OCR_CACHE_DIR/<basename>
    p = tools.download_and_verify(url, sha256=sha, verbose=False)
    assert p == str(cache / "weights.bin") and open(p, "rb").read() == b"abc" * 1000
    # 2. cached with the right hash: the URL is not touched (it no longer exists)
    src.rename(tmp_path / "remote" / "moved.bin")
    assert tools.download_and_verify(url, sha256=sha, verbose=False) == p
    # 3. cached copy corrupted: fetched again (the URL exists again)
    (tmp_path / "remote" / "moved.bin").rename(src)
    open(p, "wb").write(b"corrupted")
    tools.download_and_verify(url, sha256=sha, verbose=False)
    assert open(p, "rb").read() == b"abc" * 1000
    # 4. the fetched file does not have the promised hash
    with pytest.raises(AssertionError):
        tools.download_and_verify(url, sha256="0" * 64, verbose=False)
    # 5. explicit filename and cache_dir; no hash -> whatever is cached is accepted
    q = tools.download_and_verify(url, cache_dir=str(tmp_path / "c2"), filename="sub/name.bin", verbose=False)
    assert q == str(tmp_path / "c2" / "sub" / "name.bin") and os.path.isfile(q)
    open(q, "wb").write(b"edited")
    assert tools.download_and_verify(url, cache_dir=str(tmp_path / "c2"), filename="sub/name.bin", verbose=False) == q
    assert open(q, "rb").read() == b"edited"


def test_registry_matches_reference():
    """detection.py:647-658, recognition.py:27-44 (Appendix E of SURVEY.md)."""
    import keras_ocr_amd as k

    d = k.detection.PRETRAINED_WEIGHTS
    assert d[("clovaai_general", True)]["filename"] == "craft_mlt_25k.pth"
    assert d[("clovaai_general", False)]["sha256"] == "7283ce2ff05a0617e9740c316175ff3bacdd7215dbdf1a726890d5099431f899"
    r = k.recognition.PRETRAINED_WEIGHTS["kurapan"]["weights"]
    assert r["top"]["sha256"] == "a7d8086ac8f5c3d6a0a828f7d6fbabcaf815415dd125c32533013f85603be46d"
    assert r["notop"]["sha256"] == "027fd2cced3cbea0c4f5894bb8e9e85bac04f11daf96b8fdcf1e4ee95dcf51b9"
    assert all(v["url"].startswith("https://github.com/faustomorales/keras-ocr/releases/download/v0.8.4/") for v in d.values())
