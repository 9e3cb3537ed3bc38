"""CPU suite, part 5: evaluation / drawing helpers against the reference's own
test values (reference tests/test_evaluation.py:4-10)."""
import numpy as np


def test_iou_score_reference_values():
    import keras_ocr_amd

    box1 = [(0, 0), (100, 0), (100, 100), (0, 100)]
    box2 = [(50, 50), (100, 50), (100, 100), (50, 100)]
    assert keras_ocr_amd.evaluation.iou_score(box1, box2) == 0.25
    box2 = [(100, 100), (200, 100), (200, 200), (100, 200)]
    assert keras_ocr_amd.evaluation.iou_score(box1, box2) == 0.0
    # 2-point form, rotated quad, non-convex polygon
    assert keras_ocr_amd.evaluation.iou_score([(0, 0), (10, 10)], [(0, 0), (10, 0), (10, 10), (0, 10)]) == 1.0
    diamond = [(50, 0), (100, 50), (50, 100), (0, 50)]
    np.testing.assert_allclose(keras_ocr_amd.evaluation.iou_score(box1, diamond), 0.5)
    ell = [(0, 0), (100, 0), (100, 50), (50, 50), (50, 100), (0, 100)]  # L shape, area 7500
    np.testing.assert_allclose(keras_ocr_amd.evaluation.iou_score(box1, ell), 0.75)


def test_score_precision_recall():
    import keras_ocr_amd

    sq = lambda x, y: [(x, y), (x + 10, y), (x + 10, y + 10), (x, y + 10)]  # noqa: E731
    true = {"a": [{"text": "hello", "vertices": sq(0, 0)}, {"text": "world", "vertices": sq(50, 0)},
                  {"text": "skip", "vertices": sq(0, 50), "ignore": True}]}
    pred = {"a": [{"text": "hallo", "vertices": sq(1, 0)}, {"text": "xxxxx", "vertices": sq(50, 1)},
                  {"text": "extra", "vertices": sq(80, 80)}]}
    results, (precision, recall) = keras_ocr_amd.evaluation.score(true, pred)
    assert len(results["true_positives"]) == 1 and len(results["near_true_positives"]) == 1
    assert len(results["false_positives"]) == 1 and len(results["false_negatives"]) == 0
    assert (precision, recall) == (0.5, 1.0)


def test_draw_boxes_marks_pixels():
    import keras_ocr_amd

    img = np.full((60, 80, 3), 255, np.uint8)
    box = np.array([[10, 10], [60, 10], [60, 40], [10, 40]], np.float32)
    out = keras_ocr_amd.tools.drawBoxes(img, [("w", box)], boxes_format="predictions", thickness=3)
    assert out.shape == img.shape and (out[10, 30] == (255, 0, 0)).all() and (out[25, 35] == 255).all()
    assert keras_ocr_amd.tools.drawBoxes(img, []) is img


def test_score_bookkeeping_rules():
    """evaluation.py:86-147: every (truth, prediction) pair over the IoU threshold is listed; an ignored truth absorbs
    its predictions silently; precision / recall count DISTINCT matched truths; translator and empty strings."""
    import string
    import keras_ocr_amd

    sq = lambda x, y: [(x, y), (x + 10, y), (x + 10, y + 10), (x, y + 10)]  # noqa: E731
    true = {"b": [{"text": "Hello!", "vertices": sq(0, 0)}, {"text": "", "vertices": sq(30, 0)},
                  {"text": "ign", "vertices": sq(60, 0), "ignore": True}, {"text": "lost", "vertices": sq(0, 40)},
                  {"text": "ign2", "vertices": sq(60, 60), "ignore": True}],
            "a": []}
    pred = {"a": [{"text": "ghost", "vertices": sq(5, 5)}],
            "b": [{"text": "hello", "vertices": sq(0, 1)}, {"text": "HELLO", "vertices": sq(1, 0)},
                  {"text": "", "vertices": sq(30, 1)}, {"text": "whatever", "vertices": sq(60, 1)}]}
    tr = str.maketrans(string.ascii_uppercase, string.ascii_lowercase, string.punctuation)
    results, (precision, recall) = keras_ocr_amd.evaluation.score(true, pred, translator=tr)
    tps = [(m["image_id"], m["true_idx"], m["pred_idx"]) for m in results["true_positives"]]
    assert tps == [("b", 0, 0), ("b", 0, 1), ("b", 1, 2)]          # two predictions on truth 0; "" == "" is similarity 1
    assert results["near_true_positives"] == []
    assert results["false_negatives"] == [{"image_id": "b", "true_idx": 3}]   # ignored truth 4 is not a false negative
    assert results["false_positives"] == [{"pred_index": 0, "image_id": "a"}]  # prediction 3 was absorbed by truth 2
    assert (precision, recall) == (2 / 3, 2 / 3)
    # without the translator "Hello!" vs "hello": distance 2 of 6 -> 0.67 still passes; at 0.9 it is a near match
    results, pr = keras_ocr_amd.evaluation.score({"b": true["b"][:2]}, {"b": [pred["b"][0], pred["b"][2]]},
                                                 similarity_threshold=0.9)
    assert [m["true_idx"] for m in results["near_true_positives"]] == [0] and [m["true_idx"] for m in results["true_positives"]] == [1]
    assert pr == (1.0, 1.0)  # a near match is neither a false positive nor a false negative
    import pytest
    with pytest.raises(AssertionError):
        keras_ocr_amd.evaluation.score({"x": []}, {"y": []})
