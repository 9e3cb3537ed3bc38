"""Host-side mirror of ``keras_ocr.pipeline.Pipeline`` (reference ``keras_ocr/pipeline.py:7-75``)."""
import numpy as np

from . import detection, recognition, tools


class Pipeline:
    """A wrapper for a combination of detector and recognizer (pipeline.py:7-26).

    Args:
        detector: The detector to use
        recognizer: The recognizer to use
        scale: The scale factor to apply to input images
        max_size: The maximum single-side dimension of images for inference.
    """

    def __init__(self, detector=None, recognizer=None, scale=2, max_size=2048):
        if detector is None:
            detector = detection.Detector()
        if recognizer is None:
            recognizer = recognition.Recognizer()
        self.scale = scale
        self.detector = detector
        self.recognizer = recognizer
        self.max_size = max_size

    def _plan(self, shapes):
        """resize_image's scale rule per image + the batch's padded size (pipeline.py:44-57)."""
        scales = [tools.resize_scale(s, self.scale, self.max_size) for s in shapes]
        dws = [int(s[1] * sc) for s, sc in zip(shapes, scales)]
        dhs = [int(s[0] * sc) for s, sc in zip(shapes, scales)]
        return scales, dhs, dws, max(dhs), max(dws)

    def recognize(self, images, detection_kwargs=None, recognition_kwargs=None):
        """Pipeline.recognize (pipeline.py:28-75): list of images (arrays or file paths) or an
        (N,H,W,3) array -> list (per image) of (text, box) tuples, boxes in input-image pixels."""
        return self.recognize_padded(images, None, None, detection_kwargs, recognition_kwargs)

    def recognize_padded(self, images, hmax, wmax, detection_kwargs=None, recognition_kwargs=None):
        """recognize() with the padded detector-input size imposed by the caller (used when a
        larger batch is sharded across GPUs: every shard pads to the WHOLE batch's size)."""
        if not isinstance(images, np.ndarray):
            images = [tools.read(image) for image in images]
        images = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        if not images:
            return []
        detection_kwargs = dict(detection_kwargs or {})
        del recognition_kwargs  # Keras predict kwargs: no effect on results
        ctx = self.detector._ctx  # pylint: disable=protected-access
        if self.recognizer._ctx is not ctx:  # pylint: disable=protected-access
            raise ValueError("detector and recognizer must share one libkocr context (one GPU)")
        scales, dhs, dws, hmax_, wmax_ = self._plan([im.shape for im in images])
        hmax = hmax_ if hmax is None else max(hmax, hmax_)
        wmax = wmax_ if wmax is None else max(wmax, wmax_)
        micro_batch = detection_kwargs.pop("batch_size", 0) or 0
        box_groups, labels = ctx.pipeline(
            images, [im.shape[0] for im in images], [im.shape[1] for im in images], dhs, dws, hmax, wmax,
            micro_batch=micro_batch, **detection_kwargs)
        return self._assemble(box_groups, labels, scales)

    def recognize_device(self, d_ptr, n, h, w, detection_kwargs=None):
        """Same as recognize() for a batch already resident in HBM: ``d_ptr`` = device pointer of an
        (n,h,w,3) uint8 tensor (e.g. ``torch.Tensor.data_ptr()``)."""
        detection_kwargs = dict(detection_kwargs or {})
        ctx = self.detector._ctx  # pylint: disable=protected-access
        scales, dhs, dws, hmax, wmax = self._plan([(h, w, 3)] * n)
        micro_batch = detection_kwargs.pop("batch_size", 0) or 0
        stride = h * w * 3
        box_groups, labels = ctx.pipeline([int(d_ptr) + i * stride for i in range(n)], [h] * n, [w] * n, dhs, dws,
                                          hmax, wmax, micro_batch=micro_batch, on_device=True, **detection_kwargs)
        return self._assemble(box_groups, labels, scales)

    def _assemble(self, box_groups, labels, scales):
        predictions = self.recognizer._decode(labels)  # pylint: disable=protected-access
        prediction_groups, start = [], 0
        for boxes in box_groups:
            prediction_groups.append(predictions[start:start + len(boxes)])
            start += len(boxes)
        box_groups = [
            tools.adjust_boxes(boxes=boxes, boxes_format="boxes", scale=1 / scale) if scale != 1 else boxes
            for boxes, scale in zip(box_groups, scales)
        ]
        return [list(zip(predictions, boxes)) for predictions, boxes in zip(prediction_groups, box_groups)]
