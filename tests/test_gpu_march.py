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
