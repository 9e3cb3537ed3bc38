"""Host-side mirror of ``keras_ocr.pipeline.Pipeline`` (reference ``keras_ocr/pipeline.py:7-75``)."""
import numpy as np

from . import detection, recognition, tools


def decode_labels(alphabet, labels):
    """Label rows -> strings, skipping the blank (= len(alphabet)) and the -1 padding (recognition.py:527-534).

    One table lookup and one UTF-32 decode for the whole batch: iterating 600 x 48 numpy scalars costs 4 ms
    per call, during which the GPU sits idle."""
    labels = np.asarray(labels)
    if labels.size == 0:
        return [""] * len(labels)
    n = len(alphabet)
    if (labels.ndim != 2 or any(len(c) != 1 or c == "\0" or 0xD800 <= ord(c) <= 0xDFFF for c in alphabet)
            or labels.min() < -1 or labels.max() > n):
        # multi-character entries, lone surrogates (no UTF-32 encoding), a single row, or indices the reference's own
        # expression would wrap / reject: evaluate that expression
        skip = (n, -1)
        rows = labels.tolist() if labels.ndim == 2 else [labels.tolist()] if labels.ndim == 1 else labels.reshape(-1, labels.shape[-1]).tolist()
        return ["".join([alphabet[i] for i in row if i not in skip]) for row in rows]
    table = np.array([ord(c) for c in alphabet] + [0], "<u4")
    text = table[np.where(labels < 0, n, labels)].tobytes().decode("utf-32-le")
    w = labels.shape[1]
    return [text[i * w:(i + 1) * w].replace("\0", "") for i in range(labels.shape[0])]


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
        box_groups, labels = self.recognize_raw(images, hmax, wmax, detection_kwargs, recognition_kwargs)
        return self.assemble(box_groups, labels)

    def recognize_raw(self, images, hmax=None, wmax=None, detection_kwargs=None, recognition_kwargs=None):
        """The fused device path up to (but not including) string assembly: returns
        ``(box_groups, label_rows)`` -- per image an (n_i,4,2) float32 array in INPUT-image pixels
        (adjust_boxes already applied, pipeline.py:66-71) and one (sum n_i, 48) int32 array of decoded
        label rows (-1 padded, recognition.py:177-182) in image order.  This fixed-width form is what
        crosses ranks in ``dist.ShardedPipeline``."""
        if not isinstance(images, np.ndarray):
            images = [tools.read(image) for image in images]
        images = [np.ascontiguousarray(im) for im in images]
        if not images:
            return [], np.zeros((0, 48), np.int32)
        detection_kwargs = dict(detection_kwargs or {})
        del recognition_kwargs  # Keras predict kwargs: no effect on results
        ctx = getattr(self.detector, "_ctx", None)
        if any(im.dtype != np.uint8 for im in images):
            # float (or any non-uint8) images: the reference's cv2 calls interpolate them in float (tools.py:394, :107);
            # the stage-wise path does the same with the float kernels (kocr_resize_pad_f32 / kocr_warp_crops_f32, round 5) --
            # off the fused fixed-point path, which is defined for uint8 pixels only
            return self._recognize_stagewise([im.astype(np.float32) for im in images], detection_kwargs, hmax, wmax)
        if ctx is None or getattr(self.recognizer, "_ctx", None) is not ctx:
            # duck-typed / separately-placed stages: the reference's stage-wise path (pipeline.py:44-75)
            return self._recognize_stagewise(images, detection_kwargs, hmax, wmax)
        scales, dhs, dws, hmax_, wmax_ = self._plan([im.shape for im in images])
        hmax = hmax_ if hmax is None else max(hmax, hmax_)
        wmax = wmax_ if wmax is None else max(wmax, wmax_)
        micro_batch = detection_kwargs.pop("batch_size", 0) or 0
        box_groups, labels = ctx.pipeline(
            images, [im.shape[0] for im in images], [im.shape[1] for im in images], dhs, dws, hmax, wmax,
            micro_batch=micro_batch, **detection_kwargs)
        return self._adjust(box_groups, scales), labels

    def _recognize_stagewise(self, images, detection_kwargs, hmax=None, wmax=None):
        """pipeline.py:44-75 with the public stage APIs only (any object with ``detect`` /
        ``recognize_from_boxes``); strings are mapped back to label rows through the recognizer's alphabet.
        ``hmax`` / ``wmax``: padded size imposed by the caller (a sharded batch pads to the WHOLE batch's size)."""
        own = getattr(self.detector, "_ctx", None)  # a libkocr-backed detector: its context also resizes (else the default one)
        resized = [tools.resize_image(image, max_scale=self.scale, max_size=self.max_size, **({"ctx": own} if own is not None else {}))
                   for image in images]
        max_height, max_width = np.array([image.shape[:2] for image, _ in resized]).max(axis=0)
        max_height = max(int(max_height), int(hmax or 0))
        max_width = max(int(max_width), int(wmax or 0))
        scales = [scale for _, scale in resized]
        padded = np.array([tools.pad(image, width=max_width, height=max_height) for image, _ in resized])
        box_groups = self.detector.detect(images=padded, **detection_kwargs)
        texts = self.recognizer.recognize_from_boxes(images=padded, box_groups=box_groups)
        alphabet = self.recognizer.alphabet
        rows = [t for group in texts for t in group]
        labels = np.full((len(rows), max([48] + [len(t) for t in rows])), -1, np.int32)
        for r, t in enumerate(rows):
            labels[r, :len(t)] = [alphabet.index(ch) for ch in t]
        return self._adjust(box_groups, scales), labels

    def recognize_device(self, d_ptr, n, h, w, detection_kwargs=None):
        """Same as recognize() for a batch already resident in HBM: ``d_ptr`` = device pointer of an
        (n,h,w,3) uint8 tensor (e.g. ``torch.Tensor.data_ptr()``)."""
        return self.assemble(*self.recognize_device_raw(d_ptr, n, h, w, detection_kwargs))

    def recognize_device_raw(self, d_ptr, n, h, w, detection_kwargs=None, device_results=None):
        """recognize_device up to (but not including) string assembly: ``(box_groups, label_rows)`` as recognize_raw.
        ``device_results`` (a dict, optional) receives where the same results still lie in HBM (``Context.
        pipeline_device_results``: boxes in DETECTOR-input pixels, before the division by the scale) plus ``scale``, for a
        caller that packs them on the device (``dist.gather_packed``); valid until the next call on the context."""
        detection_kwargs = dict(detection_kwargs or {})
        ctx = self.detector._ctx  # pylint: disable=protected-access
        scales, dhs, dws, hmax, wmax = self._plan([(h, w, 3)] * n)
        micro_batch = detection_kwargs.pop("batch_size", 0) or 0
        stride = h * w * 3
        box_groups, labels = ctx.pipeline([int(d_ptr) + i * stride for i in range(n)], [h] * n, [w] * n, dhs, dws,
                                          hmax, wmax, micro_batch=micro_batch, on_device=True, **detection_kwargs)
        if device_results is not None and n:
            device_results.update(ctx.pipeline_device_results())
            device_results["scale"] = scales[0]  # one size, one scale
        return self._adjust(box_groups, scales), labels

    @staticmethod
    def _adjust(box_groups, scales):
        """pipeline.py:66-71: boxes back to input-image pixels (identity when scale == 1)."""
        return [
            tools.adjust_boxes(boxes=boxes, boxes_format="boxes", scale=1 / scale) if scale != 1 else boxes
            for boxes, scale in zip(box_groups, scales)
        ]

    def assemble(self, box_groups, labels):
        """(box_groups, label rows) -> the reference's return value (pipeline.py:72-75)."""
        # recognition.py:527-534: label rows -> strings, skipping the blank (= len(alphabet)) and the -1 padding
        predictions = decode_labels(self.recognizer.alphabet, labels)
        out, start = [], 0
        for boxes in box_groups:
            out.append(list(zip(predictions[start:start + len(boxes)], boxes)))
            start += len(boxes)
        return out
