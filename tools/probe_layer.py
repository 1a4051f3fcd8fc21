#!/usr/bin/env python3
"""Cycle breakdown of the gather-conv kernel on one layer (needs a -DP2PHD_PROBE build selected with P2PHD_LIB).
usage: probe_layer.py cin cout H W k stride pad pad_mode transposed opad [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pix2pixhdaudiosr_amd import _ops
a = [int(v) for v in sys.argv[1:]]
cin, cout, H, W, k, stride, pad, pad_mode, transposed, opad = a[:10]
batch = a[10] if len(a) > 10 else 32
spec = _ops.ConvSpec(cin, cout, k, stride, pad, pad_mode, bool(transposed), opad, True, _ops.ACT_RELU)
x = torch.randn(batch, H, W, _ops.cpitch(cin), device="cuda").to(torch.bfloat16)
w = (torch.randn(cin, cout, k, k, device="cuda") if transposed else torch.randn(cout, cin, k, k, device="cuda")) * 0.02
d = spec.desc(batch, H, W, torch.bfloat16)
Ho, Wo = spec.out_size(d)
wp = spec.packed(w, 0, d)
y = torch.empty(batch, Ho, Wo, _ops.cpitch(cout), device="cuda", dtype=torch.bfloat16)
stats = torch.zeros(batch, _ops.cpitch(cout), 2, device="cuda")
L = _ops.lib()
ws = torch.empty(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), 16), dtype=torch.uint8, device="cuda")
call = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp), None, 0, _ops.ptr(y), _ops.ptr(stats), _ops.ptr(ws), _ops.stream_ptr()))
if os.environ.get("GCONV_BM"):
    _ops.check(L.p2phd_set_option(b"gconv_bm", int(os.environ["GCONV_BM"])))
for _ in range(3):
    call()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
L.p2phd_debug_probe(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call(); e1.record(); torch.cuda.synchronize()
L.p2phd_debug_probe(buf, 1)
if os.environ.get("FINE"):                                   # -DP2PHD_PROBE -DP2PHD_PROBE_FINE build
    tb, setup, first, loop, stage, bar2, nw, store = (float(v) for v in buf[:8])
    print(f"{' '.join(sys.argv[1:])}: {e0.elapsed_time(e1)*1e3:.0f} us; per workgroup (wave 0) cycles: table {tb/nw:.0f} | setup + DMA issue {setup/nw:.0f} | "
          f"first wait {first/nw:.0f} | K loop {loop/nw:.0f} | statistics + LDS staging {stage/nw:.0f} | barrier {bar2/nw:.0f} | stores {store/nw:.0f}; workgroups {nw:.0f}")
    sys.exit(0)
wait, bar, comp, nst, pro, epi, nw, tot = (float(v) for v in buf[:8])
print(f"{' '.join(sys.argv[1:])}: {e0.elapsed_time(e1)*1e3:.0f} us (fwd incl. stats pass); per workgroup (wave 0): total {tot/nw:.0f} cyc = "
      f"prologue {pro/nw:.0f} + loop {comp/nw:.0f} ({nst/nw:.0f} steps: wait {wait/nw:.0f}, barrier {bar/nw:.0f}) + epilogue {epi/nw:.0f}; workgroups {nw:.0f}")
