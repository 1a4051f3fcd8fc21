#!/usr/bin/env python3
"""Interleaved same-process A/B (HIP events) of the K slab order of the generic gather-GEMM loop (option k_chunk_major:
1 = (64-channel chunk, tap, channel), 0 = (tap, channel)) on the layers it applies to at configs[1] size: forward and
input gradient.  The orders sum K differently, so outputs agree to accumulation noise, not bit for bit.

    python tools/ab_korder.py            (GPU)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pix2pixhdaudiosr_amd import _ops  # noqa: E402

L = _ops.lib()
dt = torch.bfloat16
OPT = os.environ.get("AB_OPTION", "k_chunk_major").encode()


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


# name, cin, cout, k, stride, pad, transposed, (N, H, W)
LAYERS = [
    ("D s0 256->512 k4s1 (2B)", 256, 512, 4, 1, 2, False, (64, 65, 33)),
    ("D s0 256->512 k4s1 (B)", 256, 512, 4, 1, 2, False, (32, 65, 33)),
    ("D s1 256->512 k4s1 (2B)", 256, 512, 4, 1, 2, False, (64, 33, 17)),
    ("D s0 128->256 k4s2 (2B)", 128, 256, 4, 2, 2, False, (64, 129, 65)),
    ("D s0 64->128 k4s2 (2B)", 64, 128, 4, 2, 2, False, (64, 257, 129)),
    ("G down 192->384 s2", 192, 384, 3, 2, 1, False, (32, 128, 64)),
    ("G down 384->768 s2", 384, 768, 3, 2, 1, False, (32, 64, 32)),
]
only = os.environ.get("ONLY")
for name, cin, cout, k, stride, pad, tr, (N, H, W) in LAYERS:
    if only and only not in name:
        continue
    spec = _ops.ConvSpec(cin, cout, k, stride, pad, 0, tr, 0, True, _ops.ACT_LRELU if hasattr(_ops, "ACT_LRELU") else 1)
    d = spec.desc(N, H, W, dt)
    Ho, Wo = spec.out_size(d)
    x = torch.randn(N, H, W, cin, device="cuda").to(dt)
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.02
    y = torch.empty(N, Ho, Wo, cout, device="cuda", dtype=dt)
    stats = torch.zeros(N, cout, 2, device="cuda")
    wp0, wp1 = spec.packed(w, 0, d), spec.packed(w, 1, d)
    ws = _ops.workspace(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), L.p2phd_conv_dgrad_workspace_bytes(C.byref(d)), 1 << 20), "cuda")
    dy = torch.randn_like(y)
    gx = torch.empty_like(x)
    fwd = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp0), None, 0, _ops.ptr(y), _ops.ptr(stats), _ops.ptr(ws), _ops.stream_ptr()))
    dgr = lambda: _ops.check(L.p2phd_conv_dgrad(C.byref(d), _ops.ptr(dy), _ops.ptr(wp1), None, _ops.ptr(gx), _ops.ptr(ws), _ops.stream_ptr()))
    flops = 2.0 * N * Ho * Wo * cin * cout * k * k
    for what, fn, out in (("fwd", fwd, y), ("dgrad", dgr, gx)):
        res, outs = {0: [], 1: []}, {}
        for rnd in range(4):
            for v in (0, 1):
                _ops.check(L.p2phd_set_option(OPT, v))
                res[v].append(timeit(fn))
                if rnd == 0:
                    outs[v] = out.float().clone()
        _ops.check(L.p2phd_set_option(OPT, 1))
        err = float((outs[0] - outs[1]).norm() / outs[0].norm())
        a, b = min(res[0]), min(res[1])
        print(f"{name:28s} {what:5s} {OPT.decode()}=0 {a:6.1f} us ({flops / a / 1e6:5.0f} TF) | =1 {b:6.1f} us ({flops / b / 1e6:5.0f} TF)  "
              f"rel L2 between them {err:.1e}   all: {[round(v) for v in res[0]]} vs {[round(v) for v in res[1]]}", flush=True)
