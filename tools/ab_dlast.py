#!/usr/bin/env python3
"""Interleaved same-process A/B (HIP events) of csrc/dlast.hip (option dlast = 1) against the W-fold gather-GEMM launches (0) on the
discriminator's head at configs[1] size: 512 -> 1, 4 x 4 stride 1, both scales; forward (2B), input gradient with the
producer's fused InstanceNorm-backward sums (2B for the discriminator-loss pass, B for the generator-loss pass).

    python tools/ab_dlast.py            (GPU)
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


for name, (N, H, W) in (("D s0 512->1 (2B)", (64, 66, 34)), ("D s0 512->1 (B)", (32, 66, 34)), ("D s1 512->1 (2B)", (64, 34, 18)), ("D s1 512->1 (B)", (32, 34, 18))):
    CH = 512
    spec = _ops.ConvSpec(CH, 1, 4, 1, 2, 0, False, 0, False, _ops.ACT_NONE)
    d = spec.desc(N, H, W, dt)
    Ho, Wo = spec.out_size(d)
    x = torch.randn(N, H, W, CH, device="cuda").to(dt)
    w = torch.randn(1, CH, 4, 4, device="cuda") * 0.05
    b = torch.randn(1, device="cuda")
    dy = torch.zeros(N, Ho, Wo, 8, device="cuda", dtype=dt); dy[..., 0] = torch.randn(N, Ho, Wo, device="cuda").to(dt)
    addend = torch.randn_like(x)
    prev_stats = torch.zeros(N, CH, 2, device="cuda"); prev_stats[..., 1] = H * W
    y = torch.empty(N, Ho, Wo, 8, device="cuda", dtype=dt)
    gx = torch.empty_like(x)
    bst = torch.zeros(N, CH, 2, device="cuda")
    wp0, wp1 = spec.packed(w, 0, d), spec.packed(w, 1, d)
    ws = _ops.workspace(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), L.p2phd_conv_dgrad_workspace_bytes(C.byref(d)),
                            L.p2phd_conv_dgrad_bsum_workspace_bytes(C.byref(d)), 1 << 20), "cuda")
    fwd = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp0), _ops.ptr(b), 0, _ops.ptr(y), None, _ops.ptr(ws), _ops.stream_ptr()))
    dgr = lambda: _ops.check(L.p2phd_conv_dgrad_bsum(C.byref(d), _ops.ptr(dy), _ops.ptr(wp1), _ops.ptr(addend), _ops.ptr(gx), _ops.ptr(x), _ops.ptr(prev_stats),
                                                     _ops.ACT_LRELU, 1e-5, _ops.ptr(bst), _ops.ptr(ws), _ops.stream_ptr()))
    nb = x.numel() * 2
    for what, fn, out, byts in (("fwd", fwd, y, nb), ("dgrad+addend+sums", dgr, gx, 3 * nb)):
        res, outs = {0: [], 1: []}, {}
        for rnd in range(4):
            for v in (0, 1):
                _ops.check(L.p2phd_set_option(b"dlast", v))
                res[v].append(timeit(fn))
                if rnd == 0:
                    outs[v] = out.float().clone()
        _ops.check(L.p2phd_set_option(b"dlast", 1))
        err = float((outs[0] - outs[1]).norm() / outs[0].norm())
        a, bb = min(res[0]), min(res[1])
        print(f"{name:18s} {what:18s} W-fold gather-GEMM {a:6.1f} us ({byts / a / 1e6:5.2f} TB/s) | dlast {bb:6.1f} us ({byts / bb / 1e6:5.2f} TB/s)  "
              f"rel L2 between them {err:.1e}   all: {[round(v) for v in res[0]]} vs {[round(v) for v in res[1]]}", flush=True)
