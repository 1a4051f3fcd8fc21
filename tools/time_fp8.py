#!/usr/bin/env python3
"""The residual-trunk conv (Conv3x3 768->768 @32x16, B=32) and the discriminator's 256->512 k4 s1 layer: bf16 forward vs
the fp8 (e4m3 operands) forward of BASELINE configs[4], conv kernel + statistics merge, graph-timed; and cfg5's trunk
(1024 channels at 64x32)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pix2pixhdaudiosr_amd import _ops
import bench

L = _ops.lib()
def run(tag, cin, cout, k, pad, pad_mode, H, W, B=32):
    spec = _ops.ConvSpec(cin, cout, k, 1, pad, pad_mode, False, 0, True, _ops.ACT_RELU)
    d = spec.desc(B, H, W, torch.bfloat16)
    Ho, Wo = spec.out_size(d)
    x = torch.randn(B, H, W, cin, device="cuda").to(torch.bfloat16)
    x8 = x.float().clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.02
    y = torch.empty(B, Ho, Wo, cout, device="cuda", dtype=torch.bfloat16)
    y8 = torch.empty_like(y)
    stats = torch.zeros(B, cout, 2, device="cuda")
    ws = torch.empty(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), 256), dtype=torch.uint8, device="cuda")
    wp = spec.packed(w, 0, d); wq = spec.packed_fp8(w, d)
    f16 = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp), None, 0, _ops.ptr(y), _ops.ptr(stats), _ops.ptr(ws), _ops.stream_ptr()))
    f8 = lambda: _ops.check(L.p2phd_conv_fwd_fp8(C.byref(d), _ops.ptr(x8), _ops.ptr(wq), None, 0, _ops.ptr(y8), _ops.ptr(stats), _ops.ptr(ws), _ops.stream_ptr()))
    flops = 2.0 * B * Ho * Wo * cin * cout * k * k
    for r in range(2):
        ta = bench.time_graphed(f16, 20); tb = bench.time_graphed(f8, 20)
        print(f"{tag}: bf16 {ta*1e6:7.1f} us {flops/ta/1e12:6.0f} TF | fp8 {tb*1e6:7.1f} us {flops/tb/1e12:6.0f} TF | rel diff of outputs {float((y8.float()-y.float()).norm()/y.float().norm()):.3f}", flush=True)

run("trunk 768->768 3x3 reflect 32x16", 768, 768, 3, 1, 1, 32, 16)
run("D 256->512 4x4 p2 65x33", 256, 512, 4, 2, 0, 65, 33)
run("cfg5 trunk 1024->1024 3x3 64x32 B=8", 1024, 1024, 3, 1, 1, 64, 32, B=8)
