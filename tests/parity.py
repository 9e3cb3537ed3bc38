"""Shared end-to-end comparison logic of the GPU parity tests (see tests/test_baseline_sizes_gpu.py for the
reasoning): count the heat-map pixels that fall on the other side of a getBoxes threshold, and require every
oracle box no such pixel touches to be reproduced to 1e-3 px with the identical string.  The logic itself lives in
oracle/parity.py (bench.py's parity leg uses it too)."""
from oracle.parity import flips, page_report, heat_tolerance, heat_within_tolerance  # noqa: F401  (re-exported)


def compare_page(got, want, flipped, scale, heat_shape, report):
    """asserting form of oracle.parity.page_report; accumulates into `report`"""
    rep = page_report(got, want, flipped, scale)
    assert rep["strings_differ"] == 0, "a box reproduced to 1e-3 px carries a different string"
    report["boxes_equal"] += rep["boxes_equal"]
    report["boxes_moved_by_flips"] += rep["boxes_moved_by_flips"]
    report["pixels"] += heat_shape[0] * heat_shape[1]
    report["flipped_pixels"] += len(flipped)
    assert rep["unexplained"] == 0, f"{rep['unexplained']} oracle boxes missing on the GPU without a flipped threshold pixel nearby"
    # no extra boxes either, beyond what the flips can explain
    assert abs(len(got) - len(want)) <= len(flipped)
