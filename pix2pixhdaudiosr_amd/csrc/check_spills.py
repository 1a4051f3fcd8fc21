#!/usr/bin/env python3
"""Fail the build when a gconv / wgrad kernel of conv.hip uses scratch (see the Makefile rule for conv.o)."""
import re
import sys

text = open(sys.argv[1]).read()
bad, seen = [], 0
for m in re.finditer(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+)", text, re.S):
    name, scratch = m.group(1), int(m.group(2))
    if "gconv_kernel" in name or "gconv_pkernel" in name or "wgrad_kernel" in name:
        seen += 1
        if scratch:
            bad.append((name, scratch))
for line in text.splitlines():
    if "warning:" in line or "error:" in line:
        print(line)
if bad or not seen:
    for name, scratch in bad:
        print(f"SPILL: {name}: {scratch} bytes/lane of scratch", file=sys.stderr)
    if not seen:
        print("check_spills: no gconv/wgrad kernel found in the resource remarks", file=sys.stderr)
    sys.exit(1)
print(f"check_spills: {seen} MFMA kernels, no scratch")
