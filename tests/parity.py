"""Shared end-to-end comparison logic of the GPU parity tests (see tests/test_baseline_sizes_gpu.py for the
reasoning): count the heat-map pixels that fall on the other side of a getBoxes threshold, and require every
oracle box no such pixel touches to be reproduced to 1e-3 px with the identical string."""
import numpy as np


def flips(heat_gpu, heat_ref):
    """pixels whose thresholded text / link value differs between the two heat-maps (one image)"""
    f = ((heat_gpu[..., 0] > np.float32(0.4)) != (heat_ref[..., 0] > np.float32(0.4))) | \
        ((heat_gpu[..., 1] > np.float32(0.4)) != (heat_ref[..., 1] > np.float32(0.4)))
    return np.argwhere(f)  # (y, x) in heat-map pixels


def compare_page(got, want, flips, scale, heat_shape, report):
    """got / want: lists of (text, box) in INPUT-image pixels; flips: heat-map pixels (detector input / 2)."""
    gb = [np.asarray(b, np.float64) for _, b in got]
    used = set()
    unexplained = 0
    for text, box in want:
        box = np.asarray(box, np.float64)
        d = [float(np.abs(box - b).max()) if i not in used else np.inf for i, b in enumerate(gb)]
        j = int(np.argmin(d)) if d else -1
        if j >= 0 and d[j] <= 1e-3:
            used.add(j)
            assert got[j][0] == text, (got[j][0], text)
            report["boxes_equal"] += 1
            continue
        # not reproduced: must be explained by a flipped pixel inside the word's neighbourhood (box in heat-map
        # pixels = input px * scale / 2, grown by the dilation radius bound)
        hb = box * scale / 2.0
        x0, y0, x1, y1 = hb[:, 0].min() - 24, hb[:, 1].min() - 24, hb[:, 0].max() + 24, hb[:, 1].max() + 24
        near = [(y, x) for y, x in flips if x0 <= x <= x1 and y0 <= y <= y1]
        if not near:
            unexplained += 1
        report["boxes_moved_by_flips"] += 1
    report["pixels"] += heat_shape[0] * heat_shape[1]
    report["flipped_pixels"] += len(flips)
    assert unexplained == 0, f"{unexplained} oracle boxes missing on the GPU without a flipped threshold pixel nearby"
    # no extra boxes either, beyond what the flips can explain
    assert abs(len(got) - len(want)) <= len(flips)


