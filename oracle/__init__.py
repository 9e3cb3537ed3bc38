"""CPU oracle for the keras-ocr hot path — TEST INFRASTRUCTURE ONLY.

A restatement, on torch-CPU / numpy / scipy, of the algorithm behind
``keras_ocr.pipeline.Pipeline.recognize`` (reference ``keras_ocr/pipeline.py:28-75``).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``keras_ocr_amd/``) never does.

Pinning status (DESIGN.md section 4 has the full table):
  * ``oracle.craft``  -- pinned against the reference's own PyTorch statement of CRAFT
    (``keras_ocr/detection.py:472-644``) executed from /root/reference with a stub
    ``torchvision`` (``tests/golden/make_golden.py``), fixtures under ``tests/golden/``.
  * ``oracle.tools`` geometry helpers -- pinned against the reference's ``tools.py``
    functions that run on numpy/scipy alone (same script); ``oracle.crnn.stn_transform`` -- pinned against
    the reference's own ``recognition._transform`` executed through a numpy stand-in of the TF ops it uses.
  * everything that bottoms out in TensorFlow / OpenCV / shapely (Conv/BN/LSTM/CTC semantics, getBoxes' cv2
    calls, warpPerspective, resize, cvtColor, minimum_rotated_rectangle) -- those libraries and the pretrained
    weights are installed nowhere in this image, so they cannot be executed; each is cross-checked against an
    INDEPENDENT statement of the same operation (scikit-image / scipy / Qhull / Pillow fixtures generated under the
    image's second interpreter by ``tests/golden/make_golden_3p.py``; ``torch.nn`` modules;
    ``tests/test_thirdparty_crosscheck_cpu.py``).  PARITY WITH THOSE LIBRARIES' OWN NUMERICS IS UNPINNED here;
    ``tests/golden/make_golden_real.py`` is the one-command pin for a host that has them (the unmodified reference on this
    repository's synthetic weights -> ``tests/golden/real_golden.npz``; ``tests/test_real_golden.py`` then holds this oracle and
    the GPU path to it), and ``oracle.postproc.min_area_box_cv32`` restates cv2.minAreaRect's float32 rotating calipers next to
    the exact rectangle the GPU reproduces, to measure what the difference costs (``scripts/minarearect_deviation.py``).
"""
