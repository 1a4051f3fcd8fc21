"""GPU parity of the HIP DCT-II/III operators and MDCT2 / IMDCT2 (csrc/dct.hip through the C ABI) against the
reference's KAT and golden vectors: values, shapes (len(signal) quirk), autograd gradients, round trip."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, mdct_cases, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def g2():
    return np.load(os.path.join(GOLDEN, "mdct2.npz"))


def test_dct_operators(g2):
    from pix2pixhdaudiosr_amd.dct.dct_native import DCT_2N_native, IDCT_2N_native
    a = torch.arange(1, 17, dtype=torch.float32).cuda()
    d = DCT_2N_native()(a)
    assert np.max(np.abs(d.cpu().numpy() - g2["kat_dct"])) < 1e-4
    assert torch.allclose(IDCT_2N_native()(d), 2 * a, atol=1e-3)          # idct(dct(a)) = 2a
    x = torch.from_numpy(g2["dct_x"]).cuda().requires_grad_(True)
    y = DCT_2N_native()(x)
    assert tuple(y.shape) == g2["dct_y"].shape and rel_err(y.detach().cpu().numpy(), g2["dct_y"]) < TOL
    (gx,) = torch.autograd.grad((y * torch.from_numpy(g2["dct_cot"]).cuda()).sum(), x)
    assert rel_err(gx.cpu().numpy(), g2["dct_gx"]) < TOL
    x2 = torch.from_numpy(g2["idct_x"]).cuda().requires_grad_(True)
    y2 = IDCT_2N_native()(x2)
    assert rel_err(y2.detach().cpu().numpy(), g2["idct_y"]) < TOL
    (gx2,) = torch.autograd.grad((y2 * torch.from_numpy(g2["idct_cot"]).cuda()).sum(), x2)
    assert rel_err(gx2.cpu().numpy(), g2["idct_gx"]) < TOL
    with pytest.raises(NotImplementedError):
        DCT_2N_native()(torch.zeros(4, 48).cuda())


def test_mdct2_imdct2_golden(g2):
    from pix2pixhdaudiosr_amd.models.mdct import MDCT2, IMDCT2
    from pix2pixhdaudiosr_amd.dct.dct_native import DCT_2N_native, IDCT_2N_native
    for name, n_fft, hop, win, center, shape in mdct_cases(g2):
        w = torch.from_numpy(g2[f"{name}_w"])
        mdct = MDCT2(n_fft=n_fft, hop_length=hop, win_length=win, window=w, center=center, device="cuda", dct_op=DCT_2N_native())
        imdct = IMDCT2(n_fft=n_fft, hop_length=hop, win_length=win, window=w, center=center, device="cuda", idct_op=IDCT_2N_native())
        x = torch.from_numpy(g2[f"{name}_x"]).cuda().requires_grad_(True)
        S = mdct(x)
        assert tuple(S.shape) == g2[f"{name}_S"].shape, name
        assert rel_err(S.detach().cpu().numpy(), g2[f"{name}_S"]) < TOL, name
        (gx,) = torch.autograd.grad((S * torch.from_numpy(g2[f"{name}_cot"]).cuda()).sum(), x)
        assert rel_err(gx.cpu().numpy(), g2[f"{name}_gx"]) < TOL, name
        Sin = torch.from_numpy(g2[f"{name}_S"]).cuda().requires_grad_(True)
        y = imdct(Sin)
        assert tuple(y.shape) == g2[f"{name}_y"].shape and rel_err(y.detach().cpu().numpy(), g2[f"{name}_y"]) < TOL, name
        (gS,) = torch.autograd.grad((y * torch.from_numpy(g2[f"{name}_ycot"]).cuda()).sum(), Sin)
        assert rel_err(gS.cpu().numpy(), g2[f"{name}_gS"]) < TOL, name
        ol = IMDCT2(n_fft=n_fft, hop_length=hop, win_length=win, window=w, center=center, device="cuda", out_length=shape[-1])
        assert tuple(ol(Sin.detach()).shape) == g2[f"{name}_y_outlen"].shape


def test_default_geometry_roundtrip():
    """Reference default: n_fft 512, hop 256, BINS 128 -> segment 32512 (options/audio_config.py); MSE pin 2.4e-14 fp32
    (test/metrics_test.ipynb cell 5) -- bound 1e-9 as for MDCT4."""
    from pix2pixhdaudiosr_amd.models.mdct import MDCT2, IMDCT2
    from pix2pixhdaudiosr_amd.util.util import kbdwin
    x = 0.1 * torch.randn(8, 32512, generator=torch.Generator().manual_seed(3))
    mdct = MDCT2(n_fft=512, hop_length=256, win_length=512, window=kbdwin, device="cuda")
    imdct = IMDCT2(n_fft=512, hop_length=256, win_length=512, window=kbdwin, device="cuda", out_length=32512)
    S = mdct(x.cuda())
    assert tuple(S.shape) == (8, 128, 512)
    y = imdct(S).squeeze().cpu()
    assert float(((y - x) ** 2).mean()) < 1e-9
    with pytest.raises(NotImplementedError):
        MDCT2(n_fft=4096, hop_length=2048, win_length=4096, window=kbdwin, device="cuda")
    # the reference's class default (n_fft 2048): round trip at that size too
    x2 = 0.1 * torch.randn(3, 16 * 1024, generator=torch.Generator().manual_seed(4))
    S2 = MDCT2(n_fft=2048, hop_length=1024, win_length=2048, window=kbdwin, device="cuda")(x2.cuda())
    assert tuple(S2.shape) == (3, 17, 2048)
    y2 = IMDCT2(n_fft=2048, hop_length=1024, win_length=2048, window=kbdwin, device="cuda", out_length=16 * 1024)(S2).squeeze().cpu()
    assert float(((y2 - x2) ** 2).mean()) < 1e-9


def test_eval_lines_of_the_reference_scripts_under_its_own_module_names():
    """train.py:56-60 / generate_audio.py:21-25, as the reference writes them (`from dct.dct import IDCT`, `IMDCT2(window=kbdwin,
    ..., device='cuda', idct_op=_idct)`), through the zero-edit launcher's import hook (pix2pixhdaudiosr_amd.dropin): the
    objects are this build's, `IDCT()` is accepted as the fused inverse DCT, and the round trip with the model-side MDCT2 holds
    to 1e-9 (test/metrics_test.ipynb cell 5).  Run in a subprocess: the hook changes how `models` / `util` / `dct` resolve."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import sys, types
        import torch
        from pix2pixhdaudiosr_amd import dropin
        dropin.install()
        sys.modules.setdefault("util", types.ModuleType("util")).__path__ = []      # (the reference's own `util` package: out of scope)
        from models.mdct import IMDCT2, MDCT2
        from util.util import kbdwin, imdct
        from dct.dct import IDCT, DCT
        import pix2pixhdaudiosr_amd.models.mdct as ours
        assert IMDCT2 is ours.IMDCT2
        n_fft, hop, seg = 512, 256, 32512
        _idct = IDCT()
        _imdct = IMDCT2(window=kbdwin, win_length=n_fft, hop_length=hop, n_fft=n_fft, center=True, out_length=seg, device='cuda', idct_op=_idct)
        _mdct = MDCT2(window=kbdwin, win_length=n_fft, hop_length=hop, n_fft=n_fft, center=True, device='cuda', dct_op=DCT())
        x = 0.1 * torch.randn(4, seg, generator=torch.Generator().manual_seed(1))
        S = _mdct(x.cuda())
        y = _imdct(S).squeeze().cpu()
        mse = float(((y - x) ** 2).mean())
        # the operator itself under the reference's name: idct(dct(a)) = 2 a (test/DCT_test.ipynb cell 34)
        a = torch.randn(3, 64).cuda()
        r = float(((_idct(DCT()(a)) - 2 * a).abs().max()))
        print("EVAL_OK", mse, r)
        assert mse < 1e-9 and r < 1e-4
    ''')
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "EVAL_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
