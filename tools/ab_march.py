#!/usr/bin/env python3
"""Interleaved A/B of the marching kernels (csrc/march.hip, option march = 1) against the generic gather-GEMM (march = 0) on the
generator's outermost stride-2 layers at configs[1] size (B = 32, bf16), stand-alone launches, HIP events, one process.

    python tools/ab_march.py            (GPU)
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


def layer(name, cin, cout, transposed, H, W, which):
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
    prev_stats = torch.zeros(B, _ops.cpitch(cin), 2, device="cuda"); prev_stats[..., 1] = H * W
    bst = torch.empty(B, _ops.cpitch(cin), 2, device="cuda")
    wp0, wp1 = spec.packed(w, 0, d), spec.packed(w, 1, d)
    ws = _ops.workspace(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), L.p2phd_conv_dgrad_workspace_bytes(C.byref(d)),
                            L.p2phd_conv_dgrad_bsum_workspace_bytes(C.byref(d)), 1 << 20), "cuda")
    nbytes = (x.numel() + y.numel()) * 2
    if which == "fwd":
        fn = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp0), None, 0, _ops.ptr(y), _ops.ptr(stats), _ops.ptr(ws), _ops.stream_ptr()))
    elif which == "fwd_lazy":
        fn = lambda: _ops.check(L.p2phd_conv_fwd_lazy(C.byref(d), _ops.ptr(x), _ops.ptr(prev_stats), _ops.ACT_RELU, 1e-5, _ops.ptr(wp0), None, _ops.ptr(y), _ops.ptr(stats), _ops.ptr(ws), _ops.stream_ptr()))
    elif which == "wgrad_lazy":
        gw = torch.empty_like(w)
        wsw = _ops.workspace(max(L.p2phd_conv_wgrad_workspace_bytes(C.byref(d)), 1 << 20), "cuda", "w")
        fn = lambda: _ops.check(L.p2phd_conv_wgrad_lazy(C.byref(d), _ops.ptr(x), _ops.ptr(prev_stats), _ops.ACT_RELU, 1e-5, _ops.ptr(dy), _ops.ptr(gw), None, 0, _ops.ptr(wsw), _ops.stream_ptr()))
    elif which == "wgrad":
        gw = torch.empty_like(w)
        wsw = _ops.workspace(max(L.p2phd_conv_wgrad_workspace_bytes(C.byref(d)), 1 << 20), "cuda", "w")
        fn = lambda: _ops.check(L.p2phd_conv_wgrad(C.byref(d), _ops.ptr(x), _ops.ptr(dy), _ops.ptr(gw), None, _ops.ptr(wsw), _ops.stream_ptr()))
    elif which == "dgrad":
        fn = lambda: _ops.check(L.p2phd_conv_dgrad(C.byref(d), _ops.ptr(dy), _ops.ptr(wp1), None, _ops.ptr(gx), _ops.ptr(ws), _ops.stream_ptr()))
    else:
        fn = lambda: _ops.check(L.p2phd_conv_dgrad_bsum(C.byref(d), _ops.ptr(dy), _ops.ptr(wp1), None, _ops.ptr(gx), _ops.ptr(x), _ops.ptr(prev_stats),
                                                        _ops.ACT_RELU, 1e-5, _ops.ptr(bst), _ops.ptr(ws), _ops.stream_ptr()))
    res = {0: [], 1: []}
    for rnd in range(3):
        for march in ((1,) if "lazy" in which else (0, 1)):
            _ops.check(L.p2phd_set_option(b"march", march))
            res[march].append(timeit(fn))
    _ops.check(L.p2phd_set_option(b"march", 1))
    g, m = min(res[0] or [float("nan")]), min(res[1])
    print(f"{name:34s} {which:10s} generic {g:7.1f} us ({nbytes / g / 1e6:4.2f} TB/s) | marching {m:7.1f} us ({nbytes / m / 1e6:4.2f} TB/s)   "
          f"all: {[round(v) for v in res[0]]} vs {[round(v) for v in res[1]]}", flush=True)


print(f"B={B}")
layer("G down 48->96 s2 @512x256", 48, 96, False, 512, 256, "fwd")
layer("G up 96->48 convT @256x128", 96, 48, True, 256, 128, "dgrad")
layer("G up 96->48 convT @256x128", 96, 48, True, 256, 128, "dgrad+bsum")
layer("G up 96->48 convT @256x128", 96, 48, True, 256, 128, "fwd")
layer("G down 48->96 s2 @512x256", 48, 96, False, 512, 256, "dgrad")
layer("G down 48->96 s2 @512x256", 48, 96, False, 512, 256, "dgrad+bsum")
layer("G down 48->96 s2 @512x256", 48, 96, False, 512, 256, "wgrad")
layer("G down 48->96 s2 @512x256", 48, 96, False, 512, 256, "fwd_lazy")
layer("G down 48->96 s2 @512x256", 48, 96, False, 512, 256, "wgrad_lazy")
layer("G up 96->48 convT @256x128", 96, 48, True, 256, 128, "fwd_lazy")
layer("G up 96->48 convT @256x128", 96, 48, True, 256, 128, "wgrad_lazy")
layer("G up 96->48 convT @256x128", 96, 48, True, 256, 128, "wgrad")
