#!/usr/bin/env python
"""Build-time check (Makefile, after the objects): kernels whose speed depends on keeping their whole working set in
registers must not have spilled into scratch memory -- a spill changes no result, only the speed, so nothing else would
notice (lstm16_kernel holds 256 floats of U per lane next to its accumulators, at the 512-register ceiling; the
fp16 F(4,3) kernel's K loop was 30 % slower in round 4 while a struct of seven scale constants lived in scratch).
Input: the -Rpass-analysis=kernel-resource-usage remarks hipcc wrote while compiling the translation unit.
usage: check_kernel_resources.py <remarks.txt> <kernel-name-substring>=<max scratch bytes per lane> ..."""
import re
import sys


def main():
    text = open(sys.argv[1]).read()
    limits = dict(a.split("=") for a in sys.argv[2:])
    seen = {}
    for m in re.finditer(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+).*?VGPRs Spill: (\d+)", text, re.S):
        name, scratch, spill = m.group(1), int(m.group(2)), int(m.group(3))
        for key, lim in limits.items():
            if key in name:
                seen.setdefault(key, []).append((name, scratch, spill))
                if scratch > int(lim):
                    sys.exit(f"check_kernel_resources: {name} uses {scratch} bytes/lane of scratch ({spill} spilled VGPRs); "
                             f"limit {lim}: a register spill in this kernel is a large, silent performance regression")
    missing = [k for k in limits if k not in seen]
    if missing:
        sys.exit(f"check_kernel_resources: no resource remarks found for {missing} (kernel renamed?)")
    for key, rows in seen.items():
        print(f"check_kernel_resources: {key}: {len(rows)} instantiation(s), scratch <= {max(r[1] for r in rows)} bytes/lane: ok")


if __name__ == "__main__":
    main()
