"""Work model of the convolution kernels, shared by bench.py and the perf scripts.

The profiler rows of libkocr are named after the kernel family (csrc/*.hip launchers); this module says
how many matrix-core FLOPs each family ISSUES per ALGORITHMIC (direct-convolution, fp32) FLOP and on
which pipe, so that a roofline line can quote both the algorithmic rate and the
utilisation of the pipe the kernel really runs on.
"""

_FAMILIES = [
    # prefix, pipe, factor, explanation
    ("conv_w4h", "fp16", 0.5 * 3.0, "1/2 (Winograd F(4,3): 6 multiplies per 4 outputs x 3 taps) x 3 fp16x2 split products"),
    ("conv_w4q", "fp16", 0.5 * 1.0, "1/2 (Winograd F(4,3)) x 1 fp16 product (reduced-precision fast mode)"),
    ("conv_w4s", "bf16", 0.5 * 6.0, "1/2 (Winograd F(4,3): 6 multiplies per 4 outputs x 3 taps) x 6 bf16x3 split products"),
    ("conv_w4v", "bf16", 0.5 * 6.0, "1/2 (Winograd F(4,3): 6 multiplies per 4 outputs x 3 taps) x 6 bf16x3 split products"),
    ("conv_w4t", "bf16", 0.5 * 6.0, "1/2 (Winograd F(4,3): 6 multiplies per 4 outputs x 3 taps) x 6 bf16x3 split products"),
    ("conv_ws", "bf16", 2.0 / 3.0 * 6.0, "2/3 (Winograd F(2,3)) x 6 bf16x3 split products"),
    ("conv_ds", "bf16", 6.0, "6 bf16x3 split products (direct 1x1 / dilated convolution)"),
    ("conv_hh", "fp16", 3.0, "3 fp16x2 split products (direct 3x3 convolution, <= 32 output channels)"),
    ("conv_hs_256x16", "bf16", 6.0 * 10.0 / 9.0, "6 bf16x3 split products x 10/9 (3x3 taps issued in 5 pairs; <= 16 output channels)"),
    ("conv_hs", "bf16", 6.0, "6 bf16x3 split products (direct 3x3 convolution, <= 32 output channels)"),
    ("conv_k5", "bf16", 6.0 * 26.0 / 25.0, "6 bf16x3 split products x 26/25 (5x5 taps issued in 13 pairs; 16 output channels, small images)"),
]


def issued_per_algorithmic(prof_name):
    """-> {"pipe": "bf16"|"fp16"|"fp32", "factor": issued FLOPs per algorithmic FLOP, "why": text}."""
    for prefix, pipe, factor, why in _FAMILIES:
        if prof_name.startswith(prefix):
            return {"pipe": pipe, "factor": factor, "why": why}
    return {"pipe": "fp32", "factor": 1.0, "why": "1 (direct implicit GEMM on the fp32 MFMA)"}
