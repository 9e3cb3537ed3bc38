#!/usr/bin/env python
"""Developer check: per kernel of a device assembly listing (hipcc --cuda-device-only -S), where its scratch instructions sit
relative to the MFMA stream -- a scratch access between the first and the last MFMA of the K loop is a spill that costs a
full memory round trip per iteration at one wave per SIMD.  usage: isa_kloop_check.py <file.s> [kernel-substring]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for k, (i, name) in enumerate(starts):
    if want not in name:
        continue
    end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
    body = lines[i:end]
    mf = [j for j, l in enumerate(body) if "v_mfma" in l]
    sc = [j for j, l in enumerate(body) if "scratch_" in l]
    if not mf:
        continue
    # gaps between consecutive MFMAs larger than 400 lines separate the K loop(s) from prologue / epilogue code
    clusters, cur = [], [mf[0]]
    for a, b in zip(mf, mf[1:]):
        if b - a > 400:
            clusters.append(cur)
            cur = []
        cur.append(b)
    clusters.append(cur)
    inside = sum(1 for j in sc for c in clusters if len(c) > 30 and c[0] < j < c[-1])
    valu = sum(1 for l in body[mf[0]:mf[-1]] if re.match(r"\s+v_(?!mfma)", l))
    print(f"{name[:60]:60s} mfma {len(mf):4d} in {len(clusters)} cluster(s) {[len(c) for c in clusters]}, scratch ops {len(sc):3d}, "
          f"inside an MFMA cluster {inside}, non-MFMA VALU between first and last MFMA {valu}")
