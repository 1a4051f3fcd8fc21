#!/usr/bin/env python3
"""A/B timing of the gather-conv tile variants on the residual-trunk layer (interleaved rounds, one process)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pix2pixhdaudiosr_amd import _ops, _lib

def run(batch=32, cin=768, cout=768, H=32, W=16, k=3, pad=1, pad_mode=1, rounds=5, iters=10, stride=1, transposed=False, opad=0):
    spec = _ops.ConvSpec(cin, cout, k, stride, pad, pad_mode, transposed, opad, True, _ops.ACT_RELU)
    x = torch.randn(batch, H, W, _ops.cpitch(cin), device="cuda").to(torch.bfloat16)
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.02
    d = spec.desc(batch, H, W, torch.bfloat16)
    Ho, Wo = spec.out_size(d)
    wp = spec.packed(w, 0, d)
    y = torch.empty(batch, Ho, Wo, _ops.cpitch(cout), device="cuda", dtype=torch.bfloat16)
    yref = None
    stats = torch.zeros(batch, _ops.cpitch(cout), 2, device="cuda")
    L = _ops.lib()
    ws = torch.empty(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), 16), dtype=torch.uint8, device="cuda")
    call = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp), None, 0, _ops.ptr(y), _ops.ptr(stats), _ops.ptr(ws), _ops.stream_ptr()))
    flops = 2.0 * batch * Ho * Wo * cin * cout * k * k
    variants = tuple(int(v) for v in os.environ.get('VARIANTS', '0,128,256,512').split(','))
    res = {v: [] for v in variants}
    for r in range(rounds):
        for bm in variants:
            _lib.check(L.p2phd_set_option(b"gconv_bm", bm))
            call(); torch.cuda.synchronize()
            if r == 0:                                            # every variant must produce the same tensor
                if yref is None: yref = y.clone()
                else: assert torch.equal(yref, y) or float((yref.float() - y.float()).abs().max()) < 1e-2, (bm, float((yref.float() - y.float()).abs().max()))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                call()
            e1.record(); torch.cuda.synchronize()
            res[bm].append(e0.elapsed_time(e1) / iters * 1e3)
    _lib.check(L.p2phd_set_option(b"gconv_bm", 0))
    for bm, v in res.items():
        v = sorted(v)
        print(f"cin{cin} cout{cout} {H}x{W} k{k} B{batch}: BM={bm}: median {v[len(v)//2]:.1f} us  min {v[0]:.1f} us  -> {flops / v[len(v)//2] / 1e6:.0f} TFLOP/s")

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "thin":
        run(cin=2, cout=48, H=512, W=256, k=7, pad=3, pad_mode=1, rounds=3, iters=5)
        run(cin=48, cout=2, H=512, W=256, k=7, pad=3, pad_mode=1, rounds=3, iters=5)
        run(cin=4, cout=64, H=512, W=256, k=4, pad=2, pad_mode=0, stride=2, rounds=3, iters=5)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "outer":
        run(cin=48, cout=96, H=512, W=256, k=3, pad=1, pad_mode=0, stride=2, rounds=3, iters=5)
        run(cin=96, cout=192, H=256, W=128, k=3, pad=1, pad_mode=0, stride=2, rounds=3, iters=5)
        run(cin=192, cout=96, H=128, W=64, k=3, pad=1, pad_mode=0, stride=2, transposed=True, opad=1, rounds=3, iters=5)
        run(cin=96, cout=48, H=256, W=128, k=3, pad=1, pad_mode=0, stride=2, transposed=True, opad=1, rounds=3, iters=5)
        sys.exit(0)
    run()
    run(cin=384, cout=768, H=64, W=32, k=3, pad=1, pad_mode=0)
    run(cin=256, cout=512, H=65, W=33, k=4, pad=2, pad_mode=0)
