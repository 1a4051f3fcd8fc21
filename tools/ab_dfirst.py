#!/usr/bin/env python3
"""Interleaved same-process A/B (HIP events) of csrc/dfirst.hip (option dfirst = 1) against the generic gather-GEMM path (0) on the
discriminator's first layer at configs[1] size: 4 -> 64, 4 x 4 stride 2, LeakyReLU, both scales, 2B = 64 samples.

    python tools/ab_dfirst.py            (GPU)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pix2pixhdaudiosr_amd import _ops  # noqa: E402

L = _ops.lib()
dt = torch.bfloat16


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, (N, H, W) in (("D s0 4->64 k4s2 (2B)", (64, 512, 256)), ("D s1 4->64 k4s2 (2B)", (64, 256, 128)), ("D s0 4->64 k4s2 (B)", (32, 512, 256))):
    spec = _ops.ConvSpec(4, 64, 4, 2, 2, 0, False, 0, False, _ops.ACT_LRELU)
    d = spec.desc(N, H, W, dt)
    Ho, Wo = spec.out_size(d)
    x = torch.zeros(N, H, W, 8, device="cuda", dtype=dt)
    x[..., :4] = torch.randn(N, H, W, 4, device="cuda").to(dt)
    w = torch.randn(64, 4, 4, 4, device="cuda") * 0.1
    b = torch.randn(64, device="cuda") * 0.1
    y = torch.empty(N, Ho, Wo, 64, device="cuda", dtype=dt)
    wp = spec.packed(w, 0, d)
    ws = _ops.workspace(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), 1 << 20), "cuda")
    fwd = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp), _ops.ptr(b), _ops.ACT_LRELU, _ops.ptr(y), None, _ops.ptr(ws), _ops.stream_ptr()))
    nbytes = x.numel() * 2 + y.numel() * 2
    res, outs = {0: [], 1: []}, {}
    for rnd in range(4):
        for v in (0, 1):
            _ops.check(L.p2phd_set_option(b"dfirst", v))
            res[v].append(timeit(fn=fwd))
            if rnd == 0:
                outs[v] = y.float().clone()
    _ops.check(L.p2phd_set_option(b"dfirst", 1))
    err = float((outs[0] - outs[1]).norm() / outs[0].norm())
    a, bb = min(res[0]), min(res[1])
    print(f"{name:24s} generic gather-GEMM {a:6.1f} us ({nbytes / a / 1e6:5.2f} TB/s) | dfirst {bb:6.1f} us ({nbytes / bb / 1e6:5.2f} TB/s)  "
          f"rel L2 between them {err:.1e}   all: {[round(v) for v in res[0]]} vs {[round(v) for v in res[1]]}", flush=True)
