#!/usr/bin/env python3
"""Per-layer timing of every distinct conv layer of BASELINE configs[1] (forward, input gradient, weight gradient),
with the MFMA rate and the HBM rate each launch sustains.  One process, HIP events."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pix2pixhdaudiosr_amd import _ops, _lib

B = int(os.environ.get("B", "32"))
L = _ops.lib()

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

def layer(name, count, cin, cout, k, stride, pad, pad_mode, transposed, opad, H, W, norm=True, need_dgrad=True):
    dt = torch.bfloat16
    spec = _ops.ConvSpec(cin, cout, k, stride, pad, pad_mode, transposed, opad, norm, 0)
    d = spec.desc(B, H, W, dt)
    Ho, Wo = spec.out_size(d)
    x = torch.randn(B, H, W, _ops.cpitch(cin), device="cuda").to(dt)
    wshape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    w = torch.randn(wshape, device="cuda") * 0.02
    y = torch.empty(B, Ho, Wo, _ops.cpitch(cout), device="cuda", dtype=dt)
    dy = torch.randn_like(y)
    stats = torch.zeros(B, _ops.cpitch(cout), 2, device="cuda")
    wp0 = spec.packed(w, 0, d); wp1 = spec.packed(w, 1, d)
    gw = torch.empty_like(w); gx = torch.empty_like(x)
    wsf = L.p2phd_conv_fwd_workspace_bytes(C.byref(d)); wsd = L.p2phd_conv_dgrad_workspace_bytes(C.byref(d)); wsw = L.p2phd_conv_wgrad_workspace_bytes(C.byref(d))
    ws = _ops.workspace(max(wsf, wsd, wsw, 1), "cuda")
    px_out = B * Ho * Wo
    flops = 2.0 * px_out * cin * cout * k * k / (stride * stride if transposed else 1)
    esz = 2
    bytes_fwd = (x.numel() + y.numel()) * esz
    f = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp0), None, 0, _ops.ptr(y), _ops.ptr(stats if norm else None), _ops.ptr(ws), _ops.stream_ptr()))
    g = lambda: _ops.check(L.p2phd_conv_dgrad(C.byref(d), _ops.ptr(dy), _ops.ptr(wp1), None, _ops.ptr(gx), _ops.ptr(ws), _ops.stream_ptr()))
    n_rx = L.p2phd_conv_reflect_extras_elems(C.byref(d)) if (pad_mode == 1 and norm) else 0
    if n_rx:
        # as in the step (round 4): the gradient comes out of the single-launch InstanceNorm backward with the reflection
        # extras behind it, the input gradient reads them in place (timing only: the extras hold random values here)
        dyx = torch.randn(dy.numel() + n_rx, device="cuda").to(dt)
        g = lambda: _ops.check(L.p2phd_conv_dgrad_rx(C.byref(d), _ops.ptr(dyx), _ops.ptr(wp1), None, _ops.ptr(gx), _ops.stream_ptr()))
    h = lambda: _ops.check(L.p2phd_conv_wgrad(C.byref(d), _ops.ptr(x), _ops.ptr(dy), _ops.ptr(gw), None, _ops.ptr(ws), _ops.stream_ptr()))
    tf = timeit(f); tg = timeit(g) if need_dgrad else 0.0; th = timeit(h)
    def fmt(t): return f"{t*1e6:7.0f}us {flops/t/1e12:5.0f}TF {bytes_fwd/t/1e12:4.1f}TB/s" if t > 0 else " " * 27
    print(f"{name:28s} x{count:2d}  GF {flops/1e9:7.1f} | fwd {fmt(tf)} | dgrad {fmt(tg)} | wgrad {fmt(th)} | total/step {count*(tf+tg+th)*1e3:6.2f} ms", flush=True)
    return count * (tf + tg + th)

tot = 0.0
print(f"B={B}  (GF = forward GFLOP of one launch; TB/s = (input+output bytes)/time)")
# ---- GlobalGenerator ngf48 nd4 nb9 @512x256 ----
tot += layer("G c7 2->48 reflect", 1, 2, 48, 7, 1, 3, 1, False, 0, 512, 256, need_dgrad=False)
ch, H, W = 48, 512, 256
for i in range(4):
    tot += layer(f"G down {ch}->{ch*2} s2", 1, ch, ch * 2, 3, 2, 1, 0, False, 0, H, W)
    ch, H, W = ch * 2, H // 2, W // 2
tot += layer("G trunk 768->768 reflect", 18, 768, 768, 3, 1, 1, 1, False, 0, 32, 16)
for i in range(4):
    tot += layer(f"G up {ch}->{ch//2} convT", 1, ch, ch // 2, 3, 2, 1, 0, True, 1, H, W)
    ch, H, W = ch // 2, H * 2, W * 2
tot += layer("G c7 48->2 reflect tanh", 1, 48, 2, 7, 1, 3, 1, False, 0, 512, 256, norm=False)
print(f"G conv total {tot*1e3:.2f} ms/step")
# ---- MultiscaleDiscriminator, 3 forward passes, dgrad x3 (first layer only in the G pass), wgrad x2 ----
dtot = 0.0
for scale, (H, W) in enumerate(((512, 256), (256, 128))):
    dtot += layer(f"D s{scale} 4->64 k4s2", 3, 4, 64, 4, 2, 2, 0, False, 0, H, W, norm=False)
    h1, w1 = H // 2 + 1, W // 2 + 1
    dtot += layer(f"D s{scale} 64->128 k4s2", 3, 64, 128, 4, 2, 2, 0, False, 0, h1, w1)
    h2, w2 = h1 // 2 + 1, w1 // 2 + 1
    dtot += layer(f"D s{scale} 128->256 k4s2", 3, 128, 256, 4, 2, 2, 0, False, 0, h2, w2)
    h3, w3 = h2 // 2 + 1, w2 // 2 + 1
    dtot += layer(f"D s{scale} 256->512 k4s1", 3, 256, 512, 4, 1, 2, 0, False, 0, h3, w3)
    dtot += layer(f"D s{scale} 512->1 k4s1", 3, 512, 1, 4, 1, 2, 0, False, 0, h3 + 1, w3 + 1, norm=False)
print(f"D conv total (3 passes, upper bound: wgrad counted 3x not 2x) {dtot*1e3:.2f} ms/step")
