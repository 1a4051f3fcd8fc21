#!/usr/bin/env python3
"""Interleaved A/B of the HALO main loop of the 256 x 192 gather-GEMM tile (csrc/gconv_halo.inc, option gconv_halo = 1) against
the generic loop (0) on the residual-trunk layer at configs[1] size ([32, 32 x 16, 768] bf16): forward (ReflectionPad2d(1)) and
input gradient (reflection adjoint through the extras the InstanceNorm backward appends).  The two loops sum K in a different
order, so the outputs agree to fp32 accumulation noise, not bit for bit.

    python tools/ab_halo.py            (GPU)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pix2pixhdaudiosr_amd import _ops  # noqa: E402

B = int(os.environ.get("B", "32"))
H, W, CH = int(os.environ.get("H", "32")), 16, int(os.environ.get("CH", "768"))
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


spec = _ops.ConvSpec(CH, CH, 3, 1, 1, 1, False, 0, True, _ops.ACT_RELU)
d = spec.desc(B, H, W, dt)
x = torch.randn(B, H, W, CH, device="cuda").to(dt)
w = torch.randn(CH, CH, 3, 3, device="cuda") * 0.02
y = torch.empty_like(x)
stats = torch.zeros(B, CH, 2, device="cuda")
wp0, wp1 = spec.packed(w, 0, d), spec.packed(w, 1, d)
ws = _ops.workspace(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), 1 << 20), "cuda")
n_rx = L.p2phd_conv_reflect_extras_elems(C.byref(d))
assert n_rx > 0
g = torch.randn_like(x)
buf = torch.empty(x.numel() + n_rx, device="cuda", dtype=dt)
dy, rx = buf[:x.numel()].view(x.shape), buf[x.numel():]
st = torch.zeros(B, CH, 2, device="cuda"); st[..., 1] = H * W
db = torch.zeros(CH, device="cuda")
_ops.check(L.p2phd_instnorm_act_bwd_rx(d.dtype, _ops.ptr(g), _ops.ptr(x), _ops.ptr(st), _ops.ptr(dy), _ops.ptr(db), 1, B, H, W, CH, 1e-5,
                                       _ops.ACT_RELU, _ops.ptr(rx), _ops.stream_ptr()))
gx = torch.empty_like(x)
fwd = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp0), None, 0, _ops.ptr(y), _ops.ptr(stats), _ops.ptr(ws), _ops.stream_ptr()))
dgr = lambda: _ops.check(L.p2phd_conv_dgrad_rx(C.byref(d), _ops.ptr(dy), _ops.ptr(wp1), None, _ops.ptr(gx), _ops.stream_ptr()))
flops = 2.0 * B * H * W * CH * CH * 9
print(f"B={B} {H}x{W} {CH}->{CH}")
for name, fn, out in (("forward (reflect)", fwd, y), ("input gradient (reflection adjoint, extras)", dgr, gx)):
    res, outs = {0: [], 1: []}, {}
    for rnd in range(4):
        for halo in (0, 1):
            _ops.check(L.p2phd_set_option(b"gconv_halo", halo))
            res[halo].append(timeit(fn))
            if rnd == 0:
                outs[halo] = (out.float().clone(), stats.clone())
    _ops.check(L.p2phd_set_option(b"gconv_halo", 1))
    err = float((outs[0][0] - outs[1][0]).norm() / outs[0][0].norm())
    amax = float((outs[0][0] - outs[1][0]).abs().max())
    serr = float((outs[0][1] - outs[1][1]).norm() / outs[0][1].norm().clamp_min(1e-30)) if name.startswith("forward") else 0.0
    a, b = min(res[0]), min(res[1])
    print(f"{name:46s} generic {a:6.1f} us ({flops / a / 1e6:5.0f} TF) | halo {b:6.1f} us ({flops / b / 1e6:5.0f} TF)   outputs: rel L2 {err:.1e}, max abs {amax:.2e}, "
          f"statistics rel {serr:.1e}   all: {[round(v) for v in res[0]]} vs {[round(v) for v in res[1]]}", flush=True)
