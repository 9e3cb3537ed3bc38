"""The cell-grid layout of the recogniser's conv stack (Tensor::cellW, conv_w43vh_kernel MODE 2; keras_ocr_amd/csrc/crnn.cpp)
through its unit-test seam kocr_conv2d_cells: crops side by side in cells with zero gutters, one input scale and one max-|x|
slot per cell, gutters written as zeros, the flipped 'valid' 2x2 pooling fused.

Checked per crop against an fp64 convolution of that crop ALONE (same bound as tests/test_conv_gpu.py:
|err| <= 1e-6 (|x| conv |w|) + 2^-36 max|x| (1 conv |w|)), and bit for bit: a crop's result depends neither on the cell it
sits in nor on what the other cells hold (recognition.py:491-537 recognises every box independently)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# cell height (crop rows + 1), cell pitch, crop width, cells per image, images, Cin, Cout
LEVELS = [
    (32, 208, 200, 4, 2, 64, 128),    # conv_2 class: 31 x 200 crops
    (16, 104, 100, 8, 1, 128, 256),   # conv_4 class: 15 x 100, two cout blocks
    (8, 52, 50, 16, 2, 64, 130),      # conv_6 class: 7 x 50 (the last quad of a crop row has two live columns), ragged couts
]


def _grid(crops, hc, pitch, cn, n_img):
    m, hv, wv, c = crops.shape
    g = np.zeros((n_img, hc, cn * pitch, c), np.float32)
    for i in range(m):
        n, j = divmod(i, cn)
        g[n, 1:1 + hv, j * pitch:j * pitch + wv] = crops[i]
    return g


def _ungrid(g, hv, wv, pitch, cn, m):
    return np.stack([g[i // cn, 1:1 + hv, (i % cn) * pitch:(i % cn) * pitch + wv] for i in range(m)])


def _ref64(crops, wt, pre_b, post_a=None, post_b=None):
    xt = torch.from_numpy(crops).double().permute(0, 3, 1, 2)
    wtt = torch.from_numpy(wt).double().permute(3, 2, 0, 1)
    b = torch.from_numpy(pre_b).double().view(1, -1, 1, 1)
    y = F.relu(F.conv2d(xt, wtt, None, padding=1) + b)
    bound = F.conv2d(xt.abs(), wtt.abs(), None, padding=1) + b.abs()
    ones = F.conv2d(torch.ones_like(xt), wtt.abs(), None, padding=1)
    if post_a is not None:
        a = torch.from_numpy(post_a).double().view(1, -1, 1, 1)
        y = y * a + torch.from_numpy(post_b).double().view(1, -1, 1, 1)
        bound, ones = bound * a.abs(), ones * a.abs()
    perm = lambda t: t.permute(0, 2, 3, 1).numpy()
    return perm(y), perm(bound), perm(ones)


def _need_fp16(ctx):
    if ctx.get_split_mode() == 0:
        pytest.skip("cell grids exist in the fp16 arithmetic modes only")


@pytest.mark.parametrize("level", LEVELS, ids=[str(l) for l in LEVELS])
def test_cells_match_fp64_per_crop_and_gutters_are_zero(ctx, level):
    _need_fp16(ctx)
    hc, pitch, wv, cn, n_img, cin, cout = level
    hv, m = hc - 1, cn * n_img - 1  # the last cell stays empty
    rng = np.random.default_rng(hc * 1000 + cin)
    crops = np.maximum(rng.standard_normal((m, hv, wv, cin)), 0).astype(np.float32)
    crops *= (2.0 ** rng.integers(-6, 7, m)).astype(np.float32)[:, None, None, None]  # every crop its own magnitude
    wt = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32)
    pre_b = rng.uniform(-0.3, 0.3, cout).astype(np.float32)
    ctx.profile_enable(True)
    ctx.profile_reset()
    out, _, amax = ctx.conv2d_cells(_grid(crops, hc, pitch, cn, n_img), wt, pitch, wv, pre_b=pre_b, relu=True)
    rows = ctx.profile_report()
    ctx.profile_enable(False)
    assert sorted(k for k in rows if k.startswith("conv")) == ["conv_w4hv_256x128_cells" if pitch >= 64 else "conv_w4ht_256x128_cells"], sorted(rows)
    got = _ungrid(out, hv, wv, pitch, cn, m).astype(np.float64)
    want, bound, ones = _ref64(crops, wt, pre_b)
    xmax = np.abs(crops).reshape(m, -1).max(1)[:, None, None, None]
    ratio = np.abs(got - want) / np.maximum(1e-6 * bound + 2.0 ** -36 * xmax * ones, 1e-30)
    print(f"cells {level}: max err / bound = {ratio.max():.3f}")
    assert float(ratio.max()) <= 1.0
    # gutters: row 0 of every image, the columns behind the crop width of every cell -- exactly zero
    mask = np.zeros(out.shape[:3], bool)
    mask[:, 0] = True
    for j in range(cn):
        mask[:, :, j * pitch + wv:(j + 1) * pitch] = True
    assert not out[mask].any()
    # the per-cell max |x| the epilogue tracked = the maximum of what it wrote into the cell, exactly
    want_amax = np.stack([np.abs(out[:, :, j * pitch:(j + 1) * pitch]).reshape(n_img, -1).max(1) for j in range(cn)], 1)
    assert np.array_equal(amax[:, :cn].reshape(-1)[:m], want_amax.reshape(-1)[:m])


@pytest.mark.parametrize("level", LEVELS[:2], ids=[str(l) for l in LEVELS[:2]])
def test_cells_fused_pooling_relu_then_bn(ctx, level):
    """conv_3 / conv_5: ReLU, BatchNorm AFTER it (negative values possible), pooling of crop rows (2 i + 1, 2 i + 2) = cell
    rows (2 i + 2, 2 i + 3); the pooled grid has its own zero row and zero columns; the full-resolution tensor is not written."""
    _need_fp16(ctx)
    hc, pitch, wv, cn, n_img, cin, cout = level
    hv, m = hc - 1, cn * n_img
    rng = np.random.default_rng(hc * 77 + cout)
    crops = np.maximum(rng.standard_normal((m, hv, wv, cin)), 0).astype(np.float32)
    wt = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32)
    pre_b = rng.uniform(-0.3, 0.3, cout).astype(np.float32)
    post_a = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    post_b = rng.uniform(-0.6, 0.1, cout).astype(np.float32)
    grid = _grid(crops, hc, pitch, cn, n_img)
    full, _, _ = ctx.conv2d_cells(grid, wt, pitch, wv, pre_b=pre_b, relu=True, post_a=post_a, post_b=post_b)
    none, pooled, amax = ctx.conv2d_cells(grid, wt, pitch, wv, pool=True, need_full=False, pre_b=pre_b, relu=True,
                                          post_a=post_a, post_b=post_b)
    assert none is None and pooled.shape == (n_img, hc // 2, cn * pitch // 2, cout)
    # pooling of what the unfused call wrote, windows of cell rows (2 r, 2 r + 1), r >= 1; row 0 and the gutter columns zero
    want = full.reshape(n_img, hc // 2, 2, cn * pitch // 2, 2, cout).max((2, 4))
    want[:, 0] = 0
    assert np.array_equal(pooled, want)
    assert (full < 0).any()  # the BatchNorm after the ReLU did produce negative values: max(0-gutter, x) would have hidden them
    got = _ungrid(pooled, hv // 2, wv // 2, pitch // 2, cn, m)
    y, _, _ = _ref64(crops, wt, pre_b, post_a, post_b)
    ref = y[:, 1:1 + 2 * (hv // 2)].reshape(m, hv // 2, 2, wv // 2, 2, cout).max((2, 4))
    assert float(np.abs(got - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()))
    want_amax = np.stack([np.abs(full[:, :, j * pitch:(j + 1) * pitch]).reshape(n_img, -1).max(1) for j in range(cn)], 1)
    assert np.array_equal(amax, want_amax)  # the pooled tensor's slots carry the full-resolution maximum (an upper bound)


def test_a_crop_depends_neither_on_its_cell_nor_on_its_neighbours(ctx):
    _need_fp16(ctx)
    hc, pitch, wv, cn, n_img, cin, cout = LEVELS[0]
    hv, m = hc - 1, cn * n_img
    rng = np.random.default_rng(5)
    crops = np.maximum(rng.standard_normal((m, hv, wv, cin)), 0).astype(np.float32)
    wt = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32)
    a, _, _ = ctx.conv2d_cells(_grid(crops, hc, pitch, cn, n_img), wt, pitch, wv, relu=True)
    perm = rng.permutation(m)
    other = crops[perm].copy()
    other[perm.tolist().index(0)] = crops[0]  # (crop 0 moved with the permutation, unchanged)
    loud = other.copy()
    for i in range(m):
        if perm[i] != 0:
            loud[i] *= np.float32(3e4)  # every OTHER crop four decades louder: crop 0's scale must not move
    b, _, _ = ctx.conv2d_cells(_grid(loud, hc, pitch, cn, n_img), wt, pitch, wv, relu=True)
    ga, gb = _ungrid(a, hv, wv, pitch, cn, m), _ungrid(b, hv, wv, pitch, cn, m)
    i0 = perm.tolist().index(0)
    assert np.array_equal(gb[i0], ga[0])
