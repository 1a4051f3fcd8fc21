#!/usr/bin/env python3
"""Time the trunk conv (auto tile = 256x192, forced 256x256, forced 256x128) with whatever library P2PHD_LIB selects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pix2pixhdaudiosr_amd import _lib
from bench import time_trunk_conv
L = _lib.lib()
out = []
for bm in (0, 512, 256):
    _lib.check(L.p2phd_set_option(b"gconv_bm", bm))
    time_trunk_conv(32, iters=3)
    sec, flops = time_trunk_conv(32, iters=20)
    out.append(f"bm={bm}: {sec*1e6:.1f} us {flops/sec/1e12:.0f} TF")
    if hasattr(L, "p2phd_debug_probe"):
        import ctypes as C
        buf = (C.c_ulonglong * 4)()
        L.p2phd_debug_probe(buf, 1)                     # drop what the timing loops accumulated
        time_trunk_conv(32, iters=1)                    # (warm-up launch + 1 timed launch)
        torch.cuda.synchronize()
        L.p2phd_debug_probe(buf, 1)
        w, b, c, n = (float(v) for v in buf)
        out.append(f"[probe per wave-step cycles: wait {w/n:.0f} barrier {b/n:.0f} compute {c/n:.0f}]")
print(os.environ.get("P2PHD_LIB", "default").split("/")[-1], " | ".join(out))
