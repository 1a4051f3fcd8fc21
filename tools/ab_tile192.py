#!/usr/bin/env python3
"""Interleaved same-process A/B (HIP events) of the 128 x 192 tile (option tile128x192 = 1) against the 128 x 128 tile (0) on the
trunk conv of the two-scale generator (configs[2]/[3]): Conv3x3 1536 -> 1536 behind ReflectionPad2d(1) on [32, 16, 8], with the
InstanceNorm partial sums, and on two smaller planes of the same kind.

    python tools/ab_tile192.py            (GPU)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pix2pixhdaudiosr_amd import _ops  # noqa: E402

L = _ops.lib()
dt = torch.bfloat16


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, (N, H, W, ch) in (("trunk 1536 @16x8 B=32", (32, 16, 8, 1536)), ("trunk 1536 @16x8 B=16", (16, 16, 8, 1536)), ("768 @16x8 B=64", (64, 16, 8, 768))):
    spec = _ops.ConvSpec(ch, ch, 3, 1, 1, 1, False, 0, True, _ops.ACT_RELU)
    d = spec.desc(N, H, W, dt)
    x = torch.randn(N, H, W, ch, device="cuda").to(dt)
    w = torch.randn(ch, ch, 3, 3, device="cuda") * 0.02
    y = torch.empty_like(x)
    stats = torch.zeros(N, ch, 2, device="cuda")
    wp = spec.packed(w, 0, d)
    ws = _ops.workspace(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), 1 << 20), "cuda")
    fwd = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp), None, 0, _ops.ptr(y), _ops.ptr(stats), _ops.ptr(ws), _ops.stream_ptr()))
    flops = 2.0 * N * H * W * ch * ch * 9
    res, outs, sts = {0: [], 1: []}, {}, {}
    for rnd in range(4):
        for v in (0, 1):
            _ops.check(L.p2phd_set_option(b"tile128x192", v))
            res[v].append(timeit(fn=fwd))
            if rnd == 0:
                outs[v] = y.float().clone(); sts[v] = stats.clone()
    _ops.check(L.p2phd_set_option(b"tile128x192", 1))
    err = float((outs[0] - outs[1]).abs().max())
    serr = float((sts[0] - sts[1]).abs().max() / sts[0].abs().max())
    a, bb = min(res[0]), min(res[1])
    print(f"{name:24s} 128x128 {a:6.1f} us ({flops / a / 1e6:5.0f} TF) | 128x192 {bb:6.1f} us ({flops / bb / 1e6:5.0f} TF)  "
          f"max |dy| {err:.1e} stats rel {serr:.1e}   all: {[round(v) for v in res[0]]} vs {[round(v) for v in res[1]]}", flush=True)
