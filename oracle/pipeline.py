"""Oracle: ``Pipeline.recognize`` glue (reference ``keras_ocr/pipeline.py:28-75``,
``detection.py:745-785``, ``recognition.py:491-537``).  TEST INFRASTRUCTURE ONLY."""
import numpy as np

from . import craft, crnn, postproc, tools


def detect(craft_w, images, detection_threshold=0.7, text_threshold=0.4, link_threshold=0.4, size_threshold=10,
           heat_out=None):
    """Detector.detect (detection.py:745-785).  ``heat_out`` (a list) receives the heat-maps (for the parity accounting)."""
    heat = craft.detector_predict(craft_w, images)
    if heat_out is not None:
        heat_out.append(heat)
    return postproc.get_boxes(heat, detection_threshold=detection_threshold, text_threshold=text_threshold,
                              link_threshold=link_threshold, size_threshold=size_threshold)


def recognize_from_boxes(crnn_w, images, box_groups, alphabet=crnn.DEFAULT_ALPHABET):
    """Recognizer.recognize_from_boxes (recognition.py:491-537)."""
    assert len(box_groups) == len(images), "You must provide the same number of box groups as images."
    crops = []
    start_end = []
    for image, boxes in zip(images, box_groups):
        gray = tools.rgb2gray_u8(image)
        for box in boxes:
            crops.append(tools.warp_box(gray, box, target_height=31, target_width=200))
        start = 0 if not start_end else start_end[-1][1]
        start_end.append((start, start + len(boxes)))
    if not crops:
        return [[]] * len(images)
    X = np.array(crops, dtype="float32") / 255
    X = X[..., np.newaxis]
    predictions, _ = crnn.recognize_crops(crnn_w, X, alphabet)
    return [predictions[start:end] for start, end in start_end]


def recognize(craft_w, crnn_w, images, scale=2, max_size=2048, detection_kwargs=None, heat_out=None):
    """Pipeline.recognize (pipeline.py:28-75)."""
    images = [tools.resize_image(image, max_scale=scale, max_size=max_size) for image in images]
    max_height, max_width = np.array([image.shape[:2] for image, _ in images]).max(axis=0)
    scales = [s for _, s in images]
    images = np.array([tools.pad(image, width=max_width, height=max_height) for image, _ in images])
    box_groups = detect(craft_w, images, heat_out=heat_out, **(detection_kwargs or {}))
    prediction_groups = recognize_from_boxes(crnn_w, images, box_groups)
    box_groups = [tools.adjust_boxes(boxes=boxes, scale=1 / s) if s != 1 else boxes for boxes, s in zip(box_groups, scales)]
    return [list(zip(predictions, boxes)) for predictions, boxes in zip(prediction_groups, box_groups)]
