"""Marching kernels of the generator's outermost stride-2 layers (csrc/march.hip, round 4): the same launches through the
generic gather-GEMM (option march = 0) must agree to the rounding of a bf16 store; statistics to fp32 summation order.
(Against the oracle: the march_* cases of tests/test_gpu_conv.py::test_conv_block and the fused-sums test there.)"""
import ctypes as C

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _run(spec, x, w, b, cot, march):
    from pix2pixhdaudiosr_amd import _ops
    L = _ops.lib()
    _ops.check(L.p2phd_set_option(b"march", march))
    try:
        xd = x.clone().requires_grad_(True)
        wd, bd = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        xp = _ops.ToPhysical.apply(torch.bfloat16, xd)
        y = _ops.FromPhysical.apply(_ops.conv_block(xp, wd, bd, spec), spec.cout)
        g = torch.autograd.grad((y * cot).sum(), [xd, wd, bd])
        torch.cuda.synchronize()
        return y.detach(), g
    finally:
        _ops.check(L.p2phd_set_option(b"march", 1))


@pytest.mark.parametrize("geom", [(2, 16, 128), (1, 24, 256), (3, 64, 128), (2, 512, 256)], ids=lambda g: "x".join(map(str, g)))
def test_marching_forward_equals_the_gather_gemm(geom):
    """Conv2d(48, 96, 3, stride 2, padding 1) + InstanceNorm + ReLU (models/networks.py:194-195 at ngf 48)."""
    from pix2pixhdaudiosr_amd import _ops
    N, H, W = geom
    gen = torch.Generator().manual_seed(H * W + N)
    x = torch.randn(N, 48, H, W, generator=gen).cuda()
    w = (torch.randn(96, 48, 3, 3, generator=gen) * 0.05).cuda()
    b = (torch.randn(96, generator=gen) * 0.1).cuda()
    spec = _ops.ConvSpec(48, 96, 3, 2, 1, 0, False, 0, True, _ops.ACT_RELU)
    d = spec.desc(N, H, W, torch.bfloat16)
    assert _ops.lib().p2phd_conv_packed_bytes(C.byref(d), 0) > 128 * 448 * 2      # generic pack + the fragment-ordered copy
    cot = torch.randn(N, 96, H // 2, W // 2, generator=gen).cuda()
    y1, g1 = _run(spec, x, w, b, cot, 1)
    y0, g0 = _run(spec, x, w, b, cot, 0)
    assert rel_err(y1.cpu().numpy(), y0.cpu().numpy()) < 4e-3                     # two roundings of a bf16 store apart
    for a, c in zip(g1[:2], g0[:2]):                                               # (bias in front of InstanceNorm: noise on both sides)
        assert rel_err(a.cpu().numpy(), c.cpu().numpy()) < 1.2e-2                  # input gradient: the "U" kernel vs the merged sub-pixel launch


@pytest.mark.parametrize("geom", [(2, 8, 64), (1, 12, 128), (2, 256, 128)], ids=lambda g: "x".join(map(str, g)))
def test_marching_input_gradient_equals_the_gather_gemm(geom):
    """ConvTranspose2d(96, 48, 3, stride 2, padding 1, output_padding 1) (networks.py:205 at ngf 48): its input gradient is
    the 48 -> 96 stride-2 gather."""
    from pix2pixhdaudiosr_amd import _ops
    N, H, W = geom
    gen = torch.Generator().manual_seed(H * W + N + 1)
    x = torch.randn(N, 96, H, W, generator=gen).cuda()
    w = (torch.randn(96, 48, 3, 3, generator=gen) * 0.05).cuda()
    b = (torch.randn(48, generator=gen) * 0.1).cuda()
    spec = _ops.ConvSpec(96, 48, 3, 2, 1, 0, True, 1, True, _ops.ACT_RELU)
    cot = torch.randn(N, 48, 2 * H, 2 * W, generator=gen).cuda()
    y1, g1 = _run(spec, x, w, b, cot, 1)
    y0, g0 = _run(spec, x, w, b, cot, 0)
    assert rel_err(y1.cpu().numpy(), y0.cpu().numpy()) < 4e-3                     # forward: the "U" kernel vs the merged sub-pixel launch
    for a, c in zip(g1[:2], g0[:2]):                                               # input gradient: the "S" kernel vs the stride-2 gather-GEMM
        assert rel_err(a.cpu().numpy(), c.cpu().numpy()) < 1.2e-2


@pytest.mark.parametrize("kind", ["conv", "convT"])
def test_lazily_normalised_input_equals_the_materialised_form(kind):
    """Round 4: a block whose only consumer is a marching layer hands out its RAW conv output + statistics (`defer`), and that
    layer's forward and weight gradient apply InstanceNorm + ReLU while staging rows (p2phd_conv_fwd_lazy / _wgrad_lazy): the
    InstanceNorm forward pass over the plane is not run.  Outputs and every gradient must equal the materialised form BIT FOR
    BIT (same kernels, same operand values)."""
    from pix2pixhdaudiosr_amd import _ops
    gen = torch.Generator().manual_seed(41)
    if kind == "conv":            # c7-like producer 8 -> 48, then Conv2d(48, 96, 3, s2)
        N, c0, H, W = 2, 8, 16, 128
        specP = _ops.ConvSpec(c0, 48, 3, 1, 1, 1, False, 0, True, _ops.ACT_RELU)
        specL = _ops.ConvSpec(48, 96, 3, 2, 1, 0, False, 0, True, _ops.ACT_RELU)
        w1 = torch.randn(48, c0, 3, 3, generator=gen) * 0.1
        w2 = torch.randn(96, 48, 3, 3, generator=gen) * 0.05
        cout = 96
    else:                         # producer 16 -> 96, then ConvTranspose2d(96, 48, 3, s2)
        N, c0, H, W = 2, 16, 8, 64
        specP = _ops.ConvSpec(c0, 96, 3, 1, 1, 1, False, 0, True, _ops.ACT_RELU)
        specL = _ops.ConvSpec(96, 48, 3, 2, 1, 0, True, 1, True, _ops.ACT_RELU)
        w1 = torch.randn(96, c0, 3, 3, generator=gen) * 0.1
        w2 = torch.randn(96, 48, 3, 3, generator=gen) * 0.05
        cout = 48
    assert _ops.lazy_static_ok(specL)
    x = torch.randn(N, c0, H, W, generator=gen)
    b1 = torch.randn(specP.cout, generator=gen) * 0.1

    def run(defer):
        xd = x.cuda().requires_grad_(True)
        w1d, w2d, b1d = w1.cuda().requires_grad_(True), w2.cuda().requires_grad_(True), b1.cuda().requires_grad_(True)
        h = _ops.conv_block(_ops.ToPhysical.apply(torch.bfloat16, xd), w1d, b1d, specP, defer=defer)
        assert (getattr(h, "_p2phd_lazy", None) is not None) == defer
        o = _ops.conv_block(h, w2d, None, specL, exclusive=True)
        out = _ops.FromPhysical.apply(o, cout)
        cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(6)).cuda()
        g = torch.autograd.grad((out * cot).sum(), [xd, w1d, w2d])
        torch.cuda.synchronize()
        return out.detach(), g

    o1, g1 = run(True)
    o0, g0 = run(False)
    assert torch.equal(o1, o0)
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)
    # a consumer that cannot normalise on load gets the materialised tensor (fp32 here: no marching kernels)
    xd = x.cuda().requires_grad_(True)
    h = _ops.conv_block(_ops.ToPhysical.apply(torch.float32, xd), w1.cuda(), b1.cuda(), specP, defer=True)
    o = _ops.FromPhysical.apply(_ops.conv_block(h, w2.cuda(), None, specL), cout)
    h2 = _ops.conv_block(_ops.ToPhysical.apply(torch.float32, xd), w1.cuda(), b1.cuda(), specP)
    o2 = _ops.FromPhysical.apply(_ops.conv_block(h2, w2.cuda(), None, specL), cout)
    assert torch.equal(o, o2)
    with pytest.raises(Exception):
        _ops.FromPhysical.apply(h, specP.cout)                      # a raw tensor must not leave the chain


@pytest.mark.parametrize("kind", ["conv", "convT"])
def test_marching_kernels_reproduce_their_results_launch_after_launch(kind):
    """The marching kernels hand rows from wave to wave through LDS rings with one barrier per step and hand-placed waits: 100
    repeats of every launch form (forward, input gradient with and without the fused InstanceNorm-backward sums, weight
    gradient, and the lazily normalising forward / weight gradient) must reproduce the first launch bit for bit -- a race in
    the ring shows up as one differing launch in a few hundred (cf. test_halo_loop_equals_the_generic_loop)."""
    from pix2pixhdaudiosr_amd import _ops
    L = _ops.lib()
    dt = torch.bfloat16
    B = 8
    if kind == "conv":
        cin, cout, transposed, H, W = 48, 96, False, 128, 256
    else:
        cin, cout, transposed, H, W = 96, 48, True, 64, 128
    gen = torch.Generator().manual_seed(7)
    spec = _ops.ConvSpec(cin, cout, 3, 2, 1, 0, transposed, 1 if transposed else 0, True, 0)
    d = spec.desc(B, H, W, dt)
    Ho, Wo = spec.out_size(d)
    x = torch.randn(B, H, W, _ops.cpitch(cin), generator=gen).cuda().to(dt)
    w = (torch.randn((cin, cout, 3, 3) if transposed else (cout, cin, 3, 3), generator=gen) * 0.05).cuda()
    dy = torch.randn(B, Ho, Wo, _ops.cpitch(cout), generator=gen).cuda().to(dt)
    y = torch.empty_like(dy); gx = torch.empty_like(x); gw = torch.empty_like(w)
    stats = torch.zeros(B, _ops.cpitch(cout), 2, device="cuda")
    prev_stats = torch.zeros(B, _ops.cpitch(cin), 2, device="cuda"); prev_stats[..., 1] = H * W
    bst = torch.empty(B, _ops.cpitch(cin), 2, device="cuda")
    wp0, wp1 = spec.packed(w, 0, d), spec.packed(w, 1, d)
    ws = _ops.workspace(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), L.p2phd_conv_dgrad_workspace_bytes(C.byref(d)),
                            L.p2phd_conv_dgrad_bsum_workspace_bytes(C.byref(d)), L.p2phd_conv_wgrad_workspace_bytes(C.byref(d)), 1 << 20), "cuda")
    P, S = _ops.ptr, _ops.stream_ptr
    forms = {
        "fwd": (lambda: L.p2phd_conv_fwd(C.byref(d), P(x), P(wp0), None, 0, P(y), P(stats), P(ws), S()), lambda: (y, stats)),
        "dgrad": (lambda: L.p2phd_conv_dgrad(C.byref(d), P(dy), P(wp1), None, P(gx), P(ws), S()), lambda: (gx,)),
        "dgrad+sums": (lambda: L.p2phd_conv_dgrad_bsum(C.byref(d), P(dy), P(wp1), None, P(gx), P(x), P(prev_stats), _ops.ACT_RELU, 1e-5, P(bst), P(ws), S()),
                       lambda: (gx, bst)),
        "wgrad": (lambda: L.p2phd_conv_wgrad(C.byref(d), P(x), P(dy), P(gw), None, P(ws), S()), lambda: (gw,)),
        "fwd_lazy": (lambda: L.p2phd_conv_fwd_lazy(C.byref(d), P(x), P(prev_stats), _ops.ACT_RELU, 1e-5, P(wp0), None, P(y), P(stats), P(ws), S()),
                     lambda: (y, stats)),
        "wgrad_lazy": (lambda: L.p2phd_conv_wgrad_lazy(C.byref(d), P(x), P(prev_stats), _ops.ACT_RELU, 1e-5, P(dy), P(gw), None, 0, P(ws), S()),
                       lambda: (gw,)),
    }
    assert L.p2phd_conv_lazy_ok(C.byref(d))
    for name, (call, outs) in forms.items():
        _ops.check(call(), name)
        first = [t.clone() for t in outs()]
        assert all(torch.isfinite(t.float()).all() for t in first), name
        for rep in range(100):
            _ops.check(call(), name)
            for a, b in zip(outs(), first):
                assert torch.equal(a, b), (name, rep)
