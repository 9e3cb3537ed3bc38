"""Synthetic inputs shared by the tests (seeded; no network, no datasets)."""
import numpy as np


def gaussian(h, w, cx, cy, sx, sy, theta=0.0, amp=1.0):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    c, s = np.cos(theta), np.sin(theta)
    u = (xx - cx) * c + (yy - cy) * s
    v = -(xx - cx) * s + (yy - cy) * c
    return amp * np.exp(-0.5 * ((u / sx) ** 2 + (v / sy) ** 2))


def word_heatmap(h, w, words, noise=0.0, seed=0):
    """words: list of (cx, cy, n_chars, pitch, char_sigma, theta, amp).  Builds the text map as
    one Gaussian per character (the shape get_gaussian_heatmap produces, detection.py:55-62)
    and the link map as one Gaussian between neighbouring characters."""
    text = np.zeros((h, w))
    link = np.zeros((h, w))
    for cx, cy, n, pitch, sig, th, amp in words:
        c, s = np.cos(th), np.sin(th)
        centres = [(cx + (i - (n - 1) / 2) * pitch * c, cy + (i - (n - 1) / 2) * pitch * s) for i in range(n)]
        for x, y in centres:
            text = np.maximum(text, gaussian(h, w, x, y, sig, sig * 1.1, th, amp))
        for (x0, y0), (x1, y1) in zip(centres[:-1], centres[1:]):
            link = np.maximum(link, gaussian(h, w, (x0 + x1) / 2, (y0 + y1) / 2, sig * 0.6, sig * 0.6, th, amp))
    y = np.stack([text, link], -1)
    if noise:
        y = y + np.random.default_rng(seed).normal(0, noise, y.shape)
    return y.astype(np.float32)


def heatmap_batch():
    """A batch exercising every branch of getBoxes (detection.py:207-287)."""
    h, w = 120, 160
    maps = []
    # 0: three horizontal words + one rotated + an isolated character (diamond branch)
    maps.append(word_heatmap(h, w, [(40, 20, 4, 11, 4.5, 0.0, 1.0), (110, 30, 3, 12, 5.0, 0.0, 0.9),
                                    (80, 75, 5, 11, 4.5, 0.45, 1.0), (25, 95, 1, 0, 6.0, 0.0, 1.0)]))
    # 1: nothing above threshold
    maps.append(np.zeros((h, w, 2), np.float32))
    # 2: a weak word (max < 0.7), a tiny blob (< 10 px), a word clipped by the border, steep rotation
    m = word_heatmap(h, w, [(40, 30, 4, 11, 4.5, 0.0, 0.6), (155, 60, 4, 11, 4.5, 0.0, 1.0),
                            (60, 85, 4, 11, 4.0, -1.1, 1.0)])
    m[100:102, 100:103, 0] = 0.9
    maps.append(m)
    # 3: noisy map: many small components, label ordering matters
    maps.append(word_heatmap(h, w, [(50, 40, 6, 10, 4.0, 0.1, 1.0), (100, 90, 4, 12, 5.0, -0.2, 1.0)], noise=0.12,
                             seed=5))
    return np.stack(maps)


def text_page(h, w, n_words, seed, scale=1.0):
    """White RGB page with black rectangles-with-stripes standing in for words (uint8)."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w, 3), 255, np.uint8)
    for _ in range(n_words):
        ww = int(rng.integers(30, 90) * scale)
        hh = int(rng.integers(10, 18) * scale)
        x = int(rng.integers(0, max(1, w - ww)))
        y = int(rng.integers(0, max(1, h - hh)))
        patch = rng.integers(0, 120, (hh, ww, 3), dtype=np.uint8)
        patch[:, :: max(2, int(6 * scale))] = 255
        img[y:y + hh, x:x + ww] = patch
    return img
