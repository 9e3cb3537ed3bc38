"""CPU suite: the host-side pieces of bench.py (no GPU): synthetic pages, the parity object, the work model behind
the roofline object, and the refusal to report a multi-GPU number from fewer devices."""
import subprocess
import sys
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pages_are_seeded_and_hold_the_requested_words():
    sys.path.insert(0, ROOT)
    import bench

    a = bench.make_pages(3, 768, seed=4)
    b = bench.make_pages(3, 768, seed=4)
    c = bench.make_pages(3, 768, seed=5)
    assert a.shape == (3, 768, 768, 3) and a.dtype == np.uint8
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    # white background with dark glyphs in about WORDS_PER_PAGE grid cells
    dark = (a[0].min(-1) < 128)
    assert 0.002 < dark.mean() < 0.2


def test_parity_object():
    import bench

    box = np.array([[0, 0], [10, 0], [10, 5], [0, 5]], np.float32)
    same = bench.parity_of([("ab", box), ("c", box + 1)], [("ab", box), ("c", box + 1)])
    assert same["ok"] and same["strings_equal"] and same["boxes_max_abs_diff_px"] == 0.0
    moved = bench.parity_of([("ab", box)], [("ab", box + 0.5)])
    assert not moved["ok"] and moved["boxes_max_abs_diff_px"] == 0.5
    text = bench.parity_of([("ab", box)], [("ax", box)])
    assert not text["ok"] and not text["strings_equal"]
    count = bench.parity_of([("ab", box)], [])
    assert not count["ok"] and count["boxes_max_abs_diff_px"] is None
    assert bench.parity_of([], [])["ok"]
    # a box the GPU does not have is accepted only next to a flipped threshold pixel (heat-map px = input px * scale / 2)
    far = bench.parity_of([("ab", box)], [("ab", box), ("c", box + 300)], flipped=np.array([[2, 3]]))
    assert not far["ok"] and far["flip_accounting"]["unexplained"] == 1
    near = bench.parity_of([("ab", box)], [("ab", box), ("c", box + 300)], flipped=np.array([[301, 305]]))
    assert near["ok"] and near["flip_accounting"]["boxes_moved_by_flips"] == 1 and near["flipped_threshold_pixels"] == 1


def test_issued_work_model():
    import keras_ocr_amd as k

    f = k.perfmodel.issued_per_algorithmic
    assert f("conv_w4s_256x128_pool") == {"pipe": "bf16", "factor": 3.0, "why": f("conv_w4s_256x128")["why"]}
    assert f("conv_ws_128x128")["factor"] == 4.0
    assert f("conv_ds_256x128")["factor"] == 6.0 and f("conv_wino_128x32")["pipe"] == "fp32"
    # round 4: F(4,3) on the fp16 cores -- two pieces / three products (h), one piece / one product (q: fast mode)
    assert f("conv_w4hv_256x128_pool") == {"pipe": "fp16", "factor": 1.5, "why": f("conv_w4ht_256x128")["why"]}
    assert f("conv_w4qv_256x128")["factor"] == 0.5 and f("conv_w4qv_256x128")["pipe"] == "fp16"
    for fam in ("conv_w4v_256x128_pool", "conv_w4t_256x128"):   # round 3: vertical-reuse arrangements
        assert f(fam)["factor"] == 3.0 and f(fam)["pipe"] == "bf16"
    assert abs(f("conv_wino_128x32")["factor"] - 2 / 3) < 1e-12 and f("conv_mfma_128x64_m2")["factor"] == 1.0


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 2` with fewer than 2 visible GPUs must fail loudly, not print n_gpus: 1."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=env, timeout=600, check=False)
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stderr + r.stdout)
    assert '"metric"' not in r.stdout


def test_pmc_counter_parsing_and_corrections(tmp_path):
    """keras_ocr_amd.pmc: per-kernel sums over a rocprofv3 counter_collection.csv, the gfx950 FETCH_SIZE correction (KiB,
    doubled) and the profiler-row -> kernel-name mapping bench.py's live traffic measurement relies on."""
    import keras_ocr_amd as k

    d = tmp_path / "fetch"
    d.mkdir()
    rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp"]
    for i, (name, v) in enumerate([("void conv_w43_kernel<0, 0, 0>(W4Params)", 1000.0), ("void conv_w43_kernel<0, 0, 0>(W4Params)", 3000.0),
                                   ("void conv_w43_kernel<1, 0, 0>(W4Params)", 7.0), ("maxpool2x2_kernel(EwParams)", 512.0)]):
        rows.append(f'{i},"{name}",FETCH_SIZE,{v},{100 * i},{100 * i + 40}')
    (d / "p_counter_collection.csv").write_text("\n".join(rows) + "\n")
    agg, n = k.pmc.load_counters(str(tmp_path))
    kern = k.pmc.find_kernel(agg.keys(), "conv_w4s_256x128")
    assert kern.startswith("void conv_w43_kernel<0, 0, 0") and n[kern] == 2
    assert agg[kern]["FETCH_SIZE"] == 4000.0 and agg[kern]["_ns_FETCH_SIZE"] == 80
    assert k.pmc.fetch_bytes(agg[kern]["FETCH_SIZE"]) / n[kern] == 2000 * 1024 * 2
    assert k.pmc.write_bytes(10.0) == 10240
    assert k.pmc.find_kernel(agg.keys(), "conv_w4s_256x128_pool").startswith("void conv_w43_kernel<1")
    assert k.pmc.find_kernel(agg.keys(), "conv_hs_256x32") is None and k.pmc.find_kernel(agg.keys(), "no_such_row") is None
    assert k.perfmodel.issued_per_algorithmic("conv_hs_256x32")["factor"] == 6.0
    assert abs(k.perfmodel.issued_per_algorithmic("conv_hs_256x16")["factor"] - 60.0 / 9.0) < 1e-12
    assert abs(k.perfmodel.issued_per_algorithmic("conv_k5_352x16:stn_conv_1")["factor"] - 6.24) < 1e-12
