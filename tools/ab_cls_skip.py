#!/usr/bin/env python3
"""Interleaved A/B of the tap-skipping merged sub-pixel launches (option cls_skip = 1, round 4) against the plain merged launch
(every class multiplies all 2 x 2 taps, 7 of 16 of them packed zeros) on the generator's 3x3 stride-2 layers at configs[1] size
(B = 32, bf16): ConvTranspose2d forward and Conv2d input gradient.  Weights are re-packed after every option change.

    python tools/ab_cls_skip.py            (GPU)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pix2pixhdaudiosr_amd import _ops  # noqa: E402

B = int(os.environ.get("B", "32"))
L = _ops.lib()


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def layer(name, cin, cout, transposed, H, W):
    dt = torch.bfloat16
    spec = _ops.ConvSpec(cin, cout, 3, 2, 1, 0, transposed, 1 if transposed else 0, True, 0)
    d = spec.desc(B, H, W, dt)
    Ho, Wo = spec.out_size(d)
    x = torch.randn(B, H, W, _ops.cpitch(cin), device="cuda").to(dt)
    w = torch.randn((cin, cout, 3, 3) if transposed else (cout, cin, 3, 3), device="cuda") * 0.02
    y = torch.empty(B, Ho, Wo, _ops.cpitch(cout), device="cuda", dtype=dt)
    dy = torch.randn_like(y)
    gx = torch.empty_like(x)
    stats = torch.zeros(B, _ops.cpitch(cout), 2, device="cuda")
    ws = _ops.workspace(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), L.p2phd_conv_dgrad_workspace_bytes(C.byref(d)), 1 << 20), "cuda")
    res, outs = {0: [], 1: []}, {}
    for rnd in range(3):
        for skip in (0, 1):
            _ops.check(L.p2phd_set_option(b"cls_skip", skip))
            _ops.bump_weight_epoch()
            wp = spec.packed(w, 1 if not transposed else 0, d)
            if transposed:
                fn = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp), None, 0, _ops.ptr(y), _ops.ptr(stats), _ops.ptr(ws), _ops.stream_ptr()))
            else:
                fn = lambda: _ops.check(L.p2phd_conv_dgrad(C.byref(d), _ops.ptr(dy), _ops.ptr(wp), None, _ops.ptr(gx), _ops.ptr(ws), _ops.stream_ptr()))
            res[skip].append(timeit(fn))
            if rnd == 0:
                outs[skip] = (y if transposed else gx).float().clone()
    _ops.check(L.p2phd_set_option(b"cls_skip", 1))
    _ops.bump_weight_epoch()
    err = float((outs[0] - outs[1]).norm() / outs[0].norm())
    flops = 2.0 * B * (Ho * Wo if not transposed else H * W * 4) * cin * cout * 9 / (1 if not transposed else 4)
    a, b = min(res[0]), min(res[1])
    print(f"{name:34s} all taps {a:7.1f} us | class taps {b:7.1f} us ({flops / b / 1e6:5.0f} TF)   rel diff of the outputs {err:.1e}   "
          f"all: {[round(v) for v in res[0]]} vs {[round(v) for v in res[1]]}", flush=True)


print(f"B={B}")
layer("G up 768->384 convT fwd @32x16", 768, 384, True, 32, 16)
layer("G up 384->192 convT fwd @64x32", 384, 192, True, 64, 32)
layer("G up 192->96 convT fwd @128x64", 192, 96, True, 128, 64)
layer("G down 96->192 s2 dgrad @256x128", 96, 192, False, 256, 128)
layer("G down 192->384 s2 dgrad @128x64", 192, 384, False, 128, 64)
layer("G down 384->768 s2 dgrad @64x32", 384, 768, False, 64, 32)
