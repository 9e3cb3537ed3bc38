"""Host-side mirror of ``keras_ocr.evaluation`` (reference ``keras_ocr/evaluation.py:13-147``):
polygon IoU and precision/recall scoring of pipeline output.  SURVEY.md §8(f) item 4 — off the hot
path, pure numpy/Python: pyclipper, cv2.contourArea and editdistance are re-stated (simple polygons
are triangulated by ear clipping and intersected triangle by triangle with Sutherland-Hodgman)."""
import typing
import warnings

import numpy as np


def _area2(poly):
    x, y = poly[:, 0], poly[:, 1]
    return float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def _ccw(poly):
    return poly if _area2(poly) > 0 else poly[::-1]


def _triangulate(poly):
    """Ear clipping of a simple polygon (counter-clockwise) -> list of (3,2) triangles."""
    pts = [tuple(map(float, p)) for p in _ccw(np.asarray(poly, dtype=np.float64))]
    pts = [p for i, p in enumerate(pts) if p != pts[i - 1]]
    tris = []

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    guard = 0
    while len(pts) > 3 and guard < 10000:
        guard += 1
        n = len(pts)
        for i in range(n):
            a, b, c = pts[i - 1], pts[i], pts[(i + 1) % n]
            if cross(a, b, c) <= 0:
                continue  # reflex or degenerate corner
            if any(cross(a, b, q) >= 0 and cross(b, c, q) >= 0 and cross(c, a, q) >= 0
                   for q in pts if q not in (a, b, c)):
                continue
            tris.append(np.array([a, b, c]))
            del pts[i]
            break
        else:
            break  # numerically degenerate: stop with what is left
    if len(pts) == 3:
        tris.append(np.array(pts))
    return tris


def _clip_convex(subject, clip):
    """Sutherland-Hodgman: subject polygon clipped by a CONVEX counter-clockwise clip polygon."""
    out = [tuple(p) for p in subject]
    n = len(clip)
    for i in range(n):
        a, b = clip[i], clip[(i + 1) % n]
        inp, out = out, []
        if not inp:
            break

        def inside(p):
            return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]) >= 0

        def inter(p, q):
            d1 = (b[0] - a[0], b[1] - a[1])
            d2 = (q[0] - p[0], q[1] - p[1])
            den = d1[0] * d2[1] - d1[1] * d2[0]
            t = ((p[0] - a[0]) * d2[1] - (p[1] - a[1]) * d2[0]) / den
            return (a[0] + t * d1[0], a[1] + t * d1[1])

        s = inp[-1]
        for e in inp:
            if inside(e):
                if not inside(s):
                    out.append(inter(s, e))
                out.append(e)
            elif inside(s):
                out.append(inter(s, e))
            s = e
    return np.array(out, dtype=np.float64) if len(out) >= 3 else None


def iou_score(box1, box2):
    """evaluation.iou_score (evaluation.py:13-53): IoU of two polygons given as lists of (x, y);
    a 2-point box is an axis-aligned (x1,y1),(x2,y2) rectangle; coordinates are truncated to int32
    like the reference does before clipping."""
    if len(box1) == 2:
        (x1, y1), (x2, y2) = box1
        box1 = np.array([[x1, y1], [x2, y1], [x2, y2], [x1, y2]])
    if len(box2) == 2:
        (x1, y1), (x2, y2) = box2
        box2 = np.array([[x1, y1], [x2, y1], [x2, y2], [x1, y2]])
    p1 = np.array(box1, dtype="int32").astype(np.float64)
    p2 = np.array(box2, dtype="int32").astype(np.float64)
    a1, a2 = abs(_area2(p1)) / 2, abs(_area2(p2)) / 2
    if a1 == 0 or a2 == 0:
        warnings.warn("A box with zero area was detected.")
        return 0
    intersection = 0.0
    for t1 in _triangulate(p1):
        for t2 in _triangulate(p2):
            c = _clip_convex(_ccw(t1), _ccw(t2))
            if c is not None:
                intersection += abs(_area2(c)) / 2
    union = a1 + a2 - intersection
    return intersection / union


def _edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def _text_similarity(a, b):
    """1 - normalised Levenshtein distance (evaluation.py:119-127); two empty strings are identical."""
    longest = max(len(a), len(b))
    return 1 if longest == 0 else 1 - _edit_distance(a, b) / longest


def score(true, pred, iou_threshold=0.5, similarity_threshold=0.5, translator=None):
    """evaluation.score (evaluation.py:56-147): detection + recognition precision / recall.

    ``true`` / ``pred``: ``{image_id: [{"text", "vertices"[, "ignore"]}]}`` with the same keys.  Returns
    ``(results, (precision, recall))``; ``results`` lists ``true_positives`` / ``near_true_positives`` (one entry per
    (truth, prediction) pair whose IoU reaches ``iou_threshold``, split by text similarity), ``false_negatives``
    (truths no prediction overlaps) and ``false_positives`` (predictions overlapping no truth).  An ``ignore``d truth
    absorbs the predictions it overlaps and is itself never counted.  Precision and recall count distinct matched
    truths, as the reference does (so two predictions on one truth are one true positive).

    Written as: IoU table per image -> pair classification -> bookkeeping (the reference interleaves the three)."""
    image_ids = sorted(true)
    assert all(a == b for a, b in zip(image_ids, sorted(pred))), "true and pred dictionaries must have the same keys"
    clean = (lambda t: t.translate(translator)) if translator is not None else (lambda t: t)
    results = {"true_positives": [], "false_positives": [], "near_true_positives": [], "false_negatives": []}
    for image_id in image_ids:
        truths, preds = true[image_id], pred[image_id]
        overlaps = [[iou_score(t["vertices"], p["vertices"]) >= iou_threshold for p in preds] for t in truths]
        for ti, (truth, row) in enumerate(zip(truths, overlaps)):
            ignored = bool(truth.get("ignore", False))
            partners = [pi for pi, hit in enumerate(row) if hit]
            if not partners:
                if not ignored:
                    results["false_negatives"].append({"image_id": image_id, "true_idx": ti})
                continue
            if ignored:
                continue
            for pi in partners:
                pair = {"true_idx": ti, "pred_idx": pi, "image_id": image_id}
                good = _text_similarity(clean(truth["text"]), clean(preds[pi]["text"])) >= similarity_threshold
                results["true_positives" if good else "near_true_positives"].append(pair)
        claimed = {pi for row in overlaps for pi, hit in enumerate(row) if hit}
        results["false_positives"] += [{"pred_index": pi, "image_id": image_id} for pi in range(len(preds)) if pi not in claimed]
    n_fn, n_fp = len(results["false_negatives"]), len(results["false_positives"])
    n_tp = len({(m["image_id"], m["true_idx"]) for m in results["true_positives"]})
    return results, (n_tp / (n_tp + n_fp), n_tp / (n_tp + n_fn))
