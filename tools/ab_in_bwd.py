#!/usr/bin/env python3
"""A/B of the single-launch InstanceNorm + ReLU backward on the residual-trunk plane ([32, 32 x 16, 768] bf16): workgroup order
plain (option wgrad_xcd = 0) vs XCD-aware (1, default).  Stand-alone launches, HIP events, interleaved rounds.

    python tools/ab_in_bwd.py            (GPU)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pix2pixhdaudiosr_amd import _ops  # noqa: E402

L = _ops.lib()
B, H, W, Cc = 32, 32, 16, 768
dt = torch.bfloat16
y = torch.randn(B, H, W, Cc, device="cuda").to(dt)
g = torch.randn_like(y)
dy = torch.empty_like(y)
stats = torch.zeros(B, Cc, 2, device="cuda"); stats[..., 1] = H * W
bst = torch.empty(B, Cc, 2, device="cuda")
db = torch.zeros(Cc, device="cuda")
ex = 2 * (W + 2) + 2 * H
rx = torch.empty(B, ex, Cc, device="cuda", dtype=dt)


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


plain = lambda: _ops.check(L.p2phd_instnorm_act_bwd_acc(_ops.dt_code(dt), _ops.ptr(g), _ops.ptr(y), _ops.ptr(stats), _ops.ptr(bst), _ops.ptr(dy), _ops.ptr(db), B, H * W, Cc, 1e-5, _ops.ACT_RELU, _ops.stream_ptr()))
with_rx = lambda: _ops.check(L.p2phd_instnorm_act_bwd_rx(_ops.dt_code(dt), _ops.ptr(g), _ops.ptr(y), _ops.ptr(stats), _ops.ptr(dy), _ops.ptr(db), 1, B, H, W, Cc, 1e-5, _ops.ACT_RELU, _ops.ptr(rx), _ops.stream_ptr()))
res = {}
outs = {}
for rnd in range(3):
    for xcd in (0, 1):
        _ops.check(L.p2phd_set_option(b"wgrad_xcd", xcd))
        for name, fn in (("plain", plain), ("with reflection extras", with_rx)):
            res.setdefault((name, xcd), []).append(timeit(fn))
            outs[(name, xcd)] = (dy.clone(), rx.clone())
_ops.check(L.p2phd_set_option(b"wgrad_xcd", 1))
mb = 3 * y.numel() * 2 / 1e6
for name in ("plain", "with reflection extras"):
    a, b = min(res[(name, 0)]), min(res[(name, 1)])
    same = torch.equal(outs[(name, 0)][0], outs[(name, 1)][0]) and (name == "plain" or torch.equal(outs[(name, 0)][1], outs[(name, 1)][1]))
    print(f"in_act_bwd_fused {name:24s}: plain order {a:6.1f} us ({mb / a:.2f} TB/s) | XCD-aware {b:6.1f} us ({mb / b:.2f} TB/s)  bit-identical: {same}")
