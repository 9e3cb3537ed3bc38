"""CPU oracle for the keras-ocr hot path — TEST INFRASTRUCTURE ONLY.

A restatement, on torch-CPU / numpy / scipy, of the algorithm behind
``keras_ocr.pipeline.Pipeline.recognize`` (reference ``keras_ocr/pipeline.py:28-75``).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``keras_ocr_amd/``) never does.

Pinning status (see DESIGN.md "Oracle"):
  * ``oracle.craft``  — pinned against the reference's own PyTorch statement of CRAFT
    (``keras_ocr/detection.py:472-644``) executed from /root/reference with a stub
    ``torchvision`` (``tests/golden/make_golden.py``), fixtures under ``tests/golden/``.
  * ``oracle.tools`` geometry helpers — pinned against the reference's ``tools.py``
    functions that run on numpy/scipy alone (same script).
  * everything that bottoms out in TensorFlow / OpenCV / shapely (CRNN graph, getBoxes,
    warpPerspective, resize, cvtColor) — **parity unpinned**: those libraries, and the
    pretrained weights, are absent here; the restatement follows the reference call sites
    and the published semantics of those libraries.
"""
