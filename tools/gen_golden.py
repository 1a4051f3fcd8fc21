#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Runs only in the build container (needs /root/reference); the reference never travels to
the GPU box, the small .npz fixtures do.  Absent third-party imports the reference pulls in
at module import time are stubbed with empty modules: torchvision (VGG only),
turtle (a stray import), pysepm, torchaudio (only the two dB one-liners, pinned by the reference's
own KAT from test/metrics_test.ipynb cell 11, and `spectrogram` = |torch.stft|^power are provided).

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

REF = "/root/reference"


def _stub_modules():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m
    tv = mod("torchvision"); tv.models = mod("torchvision.models"); tv.transforms = mod("torchvision.transforms")
    mod("turtle").forward = None
    mod("pysepm")
    ta = mod("torchaudio"); taf = mod("torchaudio.functional"); ta.functional = taf

    def amplitude_to_DB(x, multiplier, amin, db_multiplier, top_db=None):
        return multiplier * torch.log10(torch.clamp(x, min=amin)) - multiplier * db_multiplier

    def DB_to_amplitude(x, ref, power):
        return ref * torch.pow(torch.pow(10.0, 0.1 * x), power)
    def spectrogram(waveform, pad, window, n_fft, hop_length, win_length, power, normalized, center=True,
                    pad_mode="reflect", onesided=True):
        # torchaudio.functional.spectrogram by its documented definition (pad = 0, normalized = False on this path)
        assert pad == 0 and not normalized
        shape = waveform.shape
        S = torch.stft(waveform.reshape(-1, shape[-1]), n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window,
                       center=center, pad_mode=pad_mode, normalized=False, onesided=onesided, return_complex=True)
        return S.reshape(shape[:-1] + S.shape[-2:]).abs().pow(power)
    taf.amplitude_to_DB = amplitude_to_DB
    taf.DB_to_amplitude = DB_to_amplitude
    taf.spectrogram = spectrogram


def _np(t):
    return t.detach().cpu().numpy().copy()      # copy: state_dict tensors alias live parameters


def gen_mdct(out):
    from models.mdct import MDCT4, IMDCT4
    from util.util import kbdwin
    d = {}
    for N in (16, 48, 64, 512, 1024, 2048):
        d[f"kbdwin_{N}"] = _np(kbdwin(N))
    # (name, n_fft, hop, win, input shape, center)
    cases = [
        ("n16_1d", 16, 8, 16, (72,), True),
        ("n16_b3", 16, 8, 16, (3, 72), True),
        ("n16_b3_odd", 16, 8, 16, (3, 77), True),           # quirk: pad from B % hop, T not hop-aligned
        ("n16_b8_odd", 16, 8, 16, (8, 77), True),           # B % hop == 0 branch
        ("n16_nocenter", 16, 8, 16, (2, 80), False),
        ("n64_b2", 64, 32, 64, (2, 480), True),
        ("n64_hop16", 64, 16, 64, (2, 480), True),          # 75 % overlap
        ("n64_win48", 64, 24, 48, (2, 480), True),          # win < n_fft (zero-padded frames)
        ("n1024_b2", 1024, 512, 1024, (2, 7 * 512), True),
        ("n1024_b2_odd", 1024, 512, 1024, (2, 3700), True),
        ("n2048_b1", 2048, 1024, 2048, (1, 5 * 1024), True),
    ]
    g = torch.Generator().manual_seed(1234)
    meta = []
    for name, n_fft, hop, win, shape, center in cases:
        x = torch.randn(*shape, generator=g) * 0.1
        x.requires_grad_(True)
        mdct = MDCT4(n_fft=n_fft, hop_length=hop, win_length=win, window=kbdwin, center=center, device="cpu")
        S = mdct(x)
        cot = torch.randn(S.shape, generator=g, dtype=torch.float64)
        (gx,) = torch.autograd.grad((S * cot).sum(), x)
        d[f"{name}_x"] = _np(x)
        d[f"{name}_S"] = _np(S)
        d[f"{name}_cot"] = _np(cot)
        d[f"{name}_gx"] = _np(gx)
        if S.dim() == 3:
            imdct = IMDCT4(n_fft=n_fft, hop_length=hop, win_length=win, window=kbdwin, center=center, device="cpu")
            S2 = S.detach().clone().requires_grad_(True)
            y = imdct(S2)
            ycot = torch.randn(y.shape, generator=g, dtype=torch.float64)
            (gS,) = torch.autograd.grad((y * ycot).sum(), S2)
            d[f"{name}_y"] = _np(y)
            d[f"{name}_ycot"] = _np(ycot)
            d[f"{name}_gS"] = _np(gS)
            imdct_ol = IMDCT4(n_fft=n_fft, hop_length=hop, win_length=win, window=kbdwin, center=center,
                              out_length=shape[-1], device="cpu")
            d[f"{name}_y_outlen"] = _np(imdct_ol(S.detach()))
        meta.append(f"{name},{n_fft},{hop},{win},{int(center)}," + "x".join(map(str, shape)))
    d["cases"] = np.array(meta)
    # frame-count quirk table: frames for (B, T) at N=1024 (bit-exact integer target)
    rows = []
    mdct = MDCT4(n_fft=1024, hop_length=512, win_length=1024, window=kbdwin, center=True, device="cpu")
    for B in (1, 2, 3, 256, 512, 513):
        for T in (130560, 130300, 130816, 1024, 1500):
            rows.append((B, T, mdct(torch.zeros(B, T)).shape[1]))
    d["quirk_frames_n1024"] = np.array(rows, dtype=np.int64)
    # round-trip MSE pins (README.md:115 geometry: 130816 samples, fp32 + fp64 window)
    x = torch.randn(130816, generator=g)
    S = mdct(x)
    imdct = IMDCT4(n_fft=1024, hop_length=512, win_length=1024, window=kbdwin, center=True, device="cpu")
    y = imdct(S.unsqueeze(0)).squeeze()
    d["roundtrip_shape"] = np.array(S.shape)
    d["roundtrip_mse_fp32win"] = np.array(float(((y[: x.numel()] - x) ** 2).mean()))
    np.savez_compressed(os.path.join(out, "mdct4.npz"), **d)
    print("mdct4.npz", len(d), "arrays")


def gen_mdct2(out):
    """MDCT2 / IMDCT2 with the reference's DCT_2N_native / IDCT_2N_native (what the shipped model uses)."""
    from models.mdct import MDCT2, IMDCT2
    from dct.dct_native import DCT_2N_native, IDCT_2N_native
    from util.util import kbdwin
    d = {}
    a = torch.arange(1, 17, dtype=torch.float32)
    d["kat_in"] = _np(a)
    d["kat_dct"] = _np(DCT_2N_native()(a))
    d["kat_idct_dct"] = _np(IDCT_2N_native()(DCT_2N_native()(a)))
    g = torch.Generator().manual_seed(99)
    r = torch.randn(3, 5, 64, generator=g).requires_grad_(True)
    y = DCT_2N_native()(r)
    c = torch.randn(y.shape, generator=g)
    (gr,) = torch.autograd.grad((y * c).sum(), r)
    d["dct_x"] = _np(r); d["dct_y"] = _np(y); d["dct_cot"] = _np(c); d["dct_gx"] = _np(gr)
    r2 = torch.randn(4, 128, generator=g).requires_grad_(True)
    y2 = IDCT_2N_native()(r2)
    c2 = torch.randn(y2.shape, generator=g)
    (gr2,) = torch.autograd.grad((y2 * c2).sum(), r2)
    d["idct_x"] = _np(r2); d["idct_y"] = _np(y2); d["idct_cot"] = _np(c2); d["idct_gx"] = _np(gr2)
    cases = [("n16_b3", 16, 8, 16, (3, 72), True), ("n64_b2_odd", 64, 32, 64, (2, 470), True),
             ("n64_hop16", 64, 16, 64, (2, 480), True), ("n64_win48", 64, 24, 48, (2, 480), True),
             ("n64_nocenter", 64, 32, 64, (2, 480), False), ("n512_b2", 512, 256, 512, (2, 7 * 256), True),
             ("n1024_b1", 1024, 512, 1024, (1, 5 * 512), True), ("n2048_b1", 2048, 1024, 2048, (1, 4 * 1024), True)]
    meta = []
    for name, n_fft, hop, win, shape, center in cases:
        x = (torch.randn(*shape, generator=g) * 0.1).requires_grad_(True)
        w = kbdwin(win)
        d[f"{name}_w"] = _np(w)
        mdct = MDCT2(n_fft=n_fft, hop_length=hop, win_length=win, window=w, center=center, device="cpu", dct_op=DCT_2N_native())
        S = mdct(x)
        cot = torch.randn(S.shape, generator=g)
        (gx,) = torch.autograd.grad((S * cot).sum(), x)
        imdct = IMDCT2(n_fft=n_fft, hop_length=hop, win_length=win, window=w, center=center, device="cpu", idct_op=IDCT_2N_native())
        S2 = S.detach().clone().requires_grad_(True)
        y = imdct(S2)
        ycot = torch.randn(y.shape, generator=g)
        (gS,) = torch.autograd.grad((y * ycot).sum(), S2)
        imdct_ol = IMDCT2(n_fft=n_fft, hop_length=hop, win_length=win, window=w, center=center, device="cpu",
                          out_length=shape[-1], idct_op=IDCT_2N_native())
        for k, v in (("x", x), ("S", S), ("cot", cot), ("gx", gx), ("y", y), ("ycot", ycot), ("gS", gS), ("y_outlen", imdct_ol(S.detach()))):
            d[f"{name}_{k}"] = _np(v)
        meta.append(f"{name},{n_fft},{hop},{win},{int(center)}," + "x".join(map(str, shape)))
    d["cases"] = np.array(meta)
    np.savez_compressed(os.path.join(out, "mdct2.npz"), **d)
    print("mdct2.npz", len(d), "arrays; KAT", d["kat_dct"][:4])


def _sd(net):
    return OrderedDict((k, _np(v)) for k, v in net.state_dict().items())


def gen_networks(out):
    import models.networks as RN
    import contextlib, io
    d = {}
    torch.manual_seed(1234)
    quiet = contextlib.redirect_stdout(io.StringIO())

    def run_G(tag, net, x):
        x = x.clone().requires_grad_(True)
        y = net(x)
        cot = torch.randn_like(y)
        grads = torch.autograd.grad((y * cot).sum(), [x] + list(net.parameters()))
        d[f"{tag}_x"] = _np(x); d[f"{tag}_y"] = _np(y); d[f"{tag}_cot"] = _np(cot); d[f"{tag}_gx"] = _np(grads[0])
        for (k, v), gr in zip(net.named_parameters(), grads[1:]):
            d[f"{tag}_p_{k}"] = _np(v); d[f"{tag}_g_{k}"] = _np(gr)
        d[f"{tag}_keys"] = np.array(list(net.state_dict().keys()))

    with quiet:
        G = RN.define_G(2, 2, 8, "global", 2, 2, 0, 0, "instance", [])
    run_G("Gglobal", G, torch.rand(2, 2, 32, 16))
    with quiet:
        G4 = RN.define_G(2, 2, 2, "global", 4, 1, 0, 0, "instance", [])
    run_G("Gglobal_nd4", G4, torch.rand(1, 2, 64, 32))
    with quiet:
        L = RN.define_G(2, 2, 4, "local", 2, 2, 1, 1, "instance", [])
    run_G("Glocal", L, torch.rand(2, 2, 32, 32))
    with quiet:
        L2 = RN.define_G(2, 2, 4, "local", 1, 1, 2, 1, "instance", [])
    run_G("Glocal2", L2, torch.rand(1, 2, 32, 32))

    for tag, gi in (("D", True), ("Dnofeat", False)):
        with quiet:
            D = RN.define_D(4, 8, 3, "instance", False, 2, gi, [])
        x = torch.rand(2, 4, 32, 16).requires_grad_(True)
        res = D(x)
        flat = [f for s in res for f in s]
        cots = [torch.randn_like(f) for f in flat]
        tot = sum((f * c).sum() for f, c in zip(flat, cots))
        grads = torch.autograd.grad(tot, [x] + list(D.parameters()))
        d[f"{tag}_x"] = _np(x); d[f"{tag}_gx"] = _np(grads[0])
        for i, (f, c) in enumerate(zip(flat, cots)):
            d[f"{tag}_f{i}"] = _np(f); d[f"{tag}_c{i}"] = _np(c)
        d[f"{tag}_nfeat"] = np.array([len(s) for s in res])
        for (k, v), gr in zip(D.named_parameters(), grads[1:]):
            d[f"{tag}_p_{k}"] = _np(v); d[f"{tag}_g_{k}"] = _np(gr)
        d[f"{tag}_keys"] = np.array(list(D.state_dict().keys()))

    # GANLoss KAT
    crit = RN.GANLoss(use_lsgan=True, tensor=torch.FloatTensor)
    pred = [[torch.rand(2, 1, 5, 3)], [torch.rand(2, 1, 3, 2)]]
    d["ganloss_p0"] = _np(pred[0][0]); d["ganloss_p1"] = _np(pred[1][0])
    d["ganloss_real"] = np.array(float(crit(pred, True))); d["ganloss_fake"] = np.array(float(crit(pred, False)))

    # parameter-count + key-list KATs at the real configurations (train_script.sh:38,49-71)
    counts = {}
    with quiet:
        for name, args in {
            "G_local_ngf48_nd4_nbg3_nle1_nbl2": (2, 2, 48, "local", 4, 3, 1, 2),
            "G_global_ngf48_nd4_nb9": (2, 2, 48, "global", 4, 9, 0, 0),
            "G_global_ngf32_nd4_nb9": (2, 2, 32, "global", 4, 9, 0, 0),
            "G_local_ngf64_default": (2, 2, 64, "local", 4, 9, 1, 3),
            "G_local_ngf48_nd3_nb9_nle2_nbl3": (2, 2, 48, "local", 3, 9, 2, 3),
        }.items():
            net = RN.define_G(*args, "instance", [])
            counts[name] = sum(p.numel() for p in net.parameters())
            d[f"keys_{name}"] = np.array(list(net.state_dict().keys()))
            del net
        for name, args in {"D_ndf64_nl3_numD2": (4, 64, 3, "instance", False, 2, True),
                           "D_ndf64_nl3_numD3": (4, 64, 3, "instance", False, 3, True)}.items():
            net = RN.define_D(*args, [])
            counts[name] = sum(p.numel() for p in net.parameters())
            d[f"keys_{name}"] = np.array(list(net.state_dict().keys()))
    d["count_names"] = np.array(list(counts.keys()))
    d["count_values"] = np.array(list(counts.values()), dtype=np.int64)
    np.savez_compressed(os.path.join(out, "networks.npz"), **d)
    print("networks.npz", len(d), "arrays;", counts)


def gen_networks_d3(out):
    """Three-scale MultiscaleDiscriminator (BASELINE configs[4]: num_D 3; reference models/networks.py:292-331): features of
    every stage of every scale, input and parameter gradients.  Own file so that networks.npz keeps its random draws."""
    import models.networks as RN
    import contextlib, io
    d = {}
    torch.manual_seed(4321)
    with contextlib.redirect_stdout(io.StringIO()):
        D = RN.define_D(4, 8, 3, "instance", False, 3, True, [])
    x = torch.rand(2, 4, 64, 32).requires_grad_(True)
    res = D(x)
    flat = [f for s in res for f in s]
    cots = [torch.randn_like(f) for f in flat]
    tot = sum((f * c).sum() for f, c in zip(flat, cots))
    grads = torch.autograd.grad(tot, [x] + list(D.parameters()))
    tag = "D3"
    d[f"{tag}_x"] = _np(x); d[f"{tag}_gx"] = _np(grads[0])
    for i, (f, c) in enumerate(zip(flat, cots)):
        d[f"{tag}_f{i}"] = _np(f); d[f"{tag}_c{i}"] = _np(c)
    d[f"{tag}_nfeat"] = np.array([len(s) for s in res])
    for (k, v), gr in zip(D.named_parameters(), grads[1:]):
        d[f"{tag}_p_{k}"] = _np(v); d[f"{tag}_g_{k}"] = _np(gr)
    d[f"{tag}_keys"] = np.array(list(D.state_dict().keys()))
    d["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(out, "networks_d3.npz"), **d)
    print("networks_d3.npz", len(d), "arrays; feature shapes", [tuple(f.shape) for f in flat])


def gen_model(out):
    """Pix2PixHDModel on CPU with the reference's own MDCT4 swapped in for MDCT2 (README.md:133 invites it)."""
    import contextlib, io
    from types import SimpleNamespace
    from models.pix2pixHD_model import Pix2PixHDModel
    from models.mdct import MDCT4, IMDCT4
    d = {}
    opt = SimpleNamespace(
        gpu_ids=[], isTrain=True, checkpoints_dir="/tmp/p2phd_golden", name="g", resize_or_crop="none",
        instance_feat=False, label_feat=False, load_features=False, label_nc=0, input_nc=2, output_nc=2,
        hr_sampling_rate=48000, lr_sampling_rate=8000, n_fft=64, hop_length=32, win_length=64, center=True,
        no_instance=True, feat_num=3, ngf=8, netG="global", n_downsample_global=2, n_blocks_global=2,
        n_local_enhancers=1, n_blocks_local=1, norm="instance", no_lsgan=False, ndf=8, n_layers_D=3, num_D=2,
        no_ganFeat_loss=False, use_hifigan_D=False, use_time_D=False, verbose=False, continue_train=False,
        load_pretrain="", which_epoch="latest", pool_size=0, lr=0.0002, beta1=0.5, no_vgg_loss=True,
        use_match_loss=False, niter_fix_global=0, explicit_encoding=True, alpha=0.6, min_value=1e-7, mask=True,
        mask_mode="mode2", phase_encoding_mode=None, lambda_feat=10.0, lambda_mat=10.0, lambda_time=0.4,
        abs_spectro=True, fp16=False, nef=16, n_downsample_E=4)
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        model = Pix2PixHDModel()
        model.initialize(opt)
    # MDCT4 returns float64 (complex128 twiddles, mdct.py:483-484) which the fp32 conv stack rejects;
    # the cast to fp32 right after the transform is the one documented deviation.
    _m4 = MDCT4(n_fft=opt.n_fft, hop_length=opt.hop_length, win_length=opt.win_length, window=model.window, device="cpu")
    class _Cast32(torch.nn.Module):
        def forward(self, a):
            return _m4(a).float()
    model._mdct = _Cast32()
    model._imdct = IMDCT4(n_fft=opt.n_fft, hop_length=opt.hop_length, win_length=opt.win_length, window=model.window, device="cpu")
    frames = 16
    T = (frames - 1) * opt.hop_length
    B = 2
    g = torch.Generator().manual_seed(77)
    hr = 0.1 * torch.randn(B, T, generator=g)
    lr = 0.1 * torch.randn(B, T, generator=g)
    d["hr"] = _np(hr); d["lr"] = _np(lr)
    d["window"] = _np(model.window)
    # hr: no randomness
    hs, hpha, hnorm = model.to_spectro(hr, mask=False)
    d["hr_spectro"] = _np(hs); d["hr_pha"] = _np(hpha)
    d["hr_max"] = _np(hnorm["max"]); d["hr_min"] = _np(hnorm["min"])
    # lr: the reference draws torch.randn inside; replay the same draw to record the noise it used
    bins = opt.n_fft // 2
    mask_size = int(bins * (1 - 1 / (48000 / 8000)))
    torch.manual_seed(4321)
    ls, lpha, lnorm = model.to_spectro(lr, mask=True)
    torch.manual_seed(4321)
    noise = torch.randn(B, 2, mask_size, frames)
    d["mask_noise"] = _np(noise)
    d["lr_spectro"] = _np(ls); d["lr_pha"] = _np(lpha)
    d["lr_max"] = _np(lnorm["max"]); d["lr_min"] = _np(lnorm["min"])
    # to_audio on the hr spectrogram (fp32 dtype path: reference multiplies fp64 MDCT output; cast here)
    aud = model.to_audio(hs.float(), {k: v.float() for k, v in hnorm.items() if k in ("max", "min")})
    d["hr_audio_rt"] = _np(aud)
    # full forward + the two backward passes (train.py:155-184)
    for k, v in model.netG.state_dict().items():
        d[f"G_p_{k}"] = _np(v)
    for k, v in model.netD.state_dict().items():
        d[f"D_p_{k}"] = _np(v)
    # the mask noise is an input of the parity tests: hand the recorded tensor to the reference's
    # torch.randn call (pix2pixHD_model.py:202) instead of relying on generator state
    _randn = torch.randn
    torch.randn = lambda *a, **k: noise.clone()
    try:
        losses, sr = model.forward(lr, None, hr, None, infer=True)
    finally:
        torch.randn = _randn
    names = model.loss_names
    d["loss_names"] = np.array(names)
    d["loss_values"] = np.array([float(l) for l in losses], dtype=np.float64)
    d["sr"] = _np(sr)
    ld = dict(zip(names, losses))
    loss_D = (ld["D_fake"] + ld["D_real"]) * 0.5
    loss_G = ld["G_GAN"] + ld.get("G_GAN_Feat", 0)
    model.optimizer_G.zero_grad(); loss_G.backward(retain_graph=True)
    for k, v in model.netG.named_parameters():
        d[f"G_g_{k}"] = _np(v.grad)
    model.optimizer_G.step()
    model.optimizer_D.zero_grad(); loss_D.backward()
    for k, v in model.netD.named_parameters():
        d[f"D_g_{k}"] = _np(v.grad)
    model.optimizer_D.step()
    for k, v in model.netG.state_dict().items():
        d[f"G_p1_{k}"] = _np(v)
    for k, v in model.netD.state_dict().items():
        d[f"D_p1_{k}"] = _np(v)
    d["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(out, "model_step.npz"), **d)
    print("model_step.npz", len(d), "arrays; losses", dict(zip(names, d["loss_values"])))


def gen_spectro_modes(out):
    """to_spectro / to_audio of the reference in the configurations besides the published one: mask_mode mode0 / mode1,
    and the single-channel encoding (explicit_encoding off) with every phase_encoding_mode (pix2pixHD_model.py:142-249).
    The reference draws its noise with torch.rand / randn / randint inside; the draws are recorded in call order and are
    inputs of the parity tests."""
    import contextlib, io
    from types import SimpleNamespace
    from models.pix2pixHD_model import Pix2PixHDModel
    from models.mdct import MDCT4, IMDCT4
    d = {}
    base = dict(
        gpu_ids=[], isTrain=True, checkpoints_dir="/tmp/p2phd_golden", name="g", resize_or_crop="none",
        instance_feat=False, label_feat=False, load_features=False, label_nc=0, input_nc=2, output_nc=2,
        hr_sampling_rate=48000, lr_sampling_rate=8000, n_fft=64, hop_length=32, win_length=64, center=True,
        no_instance=True, feat_num=3, ngf=8, netG="global", n_downsample_global=2, n_blocks_global=2,
        n_local_enhancers=1, n_blocks_local=1, norm="instance", no_lsgan=False, ndf=8, n_layers_D=3, num_D=2,
        no_ganFeat_loss=False, use_hifigan_D=False, use_time_D=False, verbose=False, continue_train=False,
        load_pretrain="", which_epoch="latest", pool_size=0, lr=0.0002, beta1=0.5, no_vgg_loss=True,
        use_match_loss=False, niter_fix_global=0, explicit_encoding=True, alpha=0.6, min_value=1e-7, mask=True,
        mask_mode="mode2", phase_encoding_mode=None, lambda_feat=10.0, lambda_mat=10.0, lambda_time=0.4,
        abs_spectro=True, fp16=False, nef=16, n_downsample_E=4)
    frames, B = 16, 2
    T = (frames - 1) * 32
    g = torch.Generator().manual_seed(99)
    lr = 0.1 * torch.randn(B, T, generator=g)
    d["lr"] = _np(lr)
    cases = [("e_mode0", dict(mask_mode="mode0")), ("e_mode1", dict(mask_mode="mode1")),
             ("p_none_mode2", dict(explicit_encoding=False)),
             ("p_uni_nomask", dict(explicit_encoding=False, phase_encoding_mode="uni_dist", mask_mode=None)),
             ("p_norm_mode0", dict(explicit_encoding=False, phase_encoding_mode="norm_dist", mask_mode="mode0")),
             ("p_norm2_mode1", dict(explicit_encoding=False, phase_encoding_mode="norm_dist2", mask_mode="mode1")),
             ("p_scale_mode2", dict(explicit_encoding=False, phase_encoding_mode="scale"))]
    for name, kw in cases:
        opt = SimpleNamespace(**dict(base, **kw))
        torch.manual_seed(1234)
        with contextlib.redirect_stdout(io.StringIO()):
            model = Pix2PixHDModel()
            model.initialize(opt)
        _m4 = MDCT4(n_fft=opt.n_fft, hop_length=opt.hop_length, win_length=opt.win_length, window=model.window, device="cpu")

        class _Cast32(torch.nn.Module):
            def forward(self, a):
                return _m4(a).float()
        model._mdct = _Cast32()
        model._imdct = IMDCT4(n_fft=opt.n_fft, hop_length=opt.hop_length, win_length=opt.win_length, window=model.window, device="cpu")
        draws = []
        real = {k: getattr(torch, k) for k in ("rand", "randn", "randint")}

        def rec(kind):
            def f(*a, **k):
                t = real[kind](*a, **k)
                draws.append((kind, t.clone()))
                return t
            return f
        torch.manual_seed(555)
        for k in real:
            setattr(torch, k, rec(k))
        try:
            ls, pha, norm = model.to_spectro(lr, mask=True)
            n_enc = len(draws)
            aud = None
            if not opt.explicit_encoding:
                aud = model.to_audio(ls.float(), {k: v.float() for k, v in norm.items() if k in ("max", "min")}, pha.float())
        finally:
            for k, v in real.items():
                setattr(torch, k, v)
        d[f"{name}_spectro"] = _np(ls); d[f"{name}_pha"] = _np(pha)
        d[f"{name}_max"] = _np(norm["max"]); d[f"{name}_min"] = _np(norm["min"])
        d[f"{name}_draw_kinds"] = np.array([k for k, _ in draws])
        d[f"{name}_n_encode_draws"] = np.array(n_enc)
        for i, (_, t) in enumerate(draws):
            d[f"{name}_draw{i}"] = _np(t)
        if aud is not None:
            d[f"{name}_audio"] = _np(aud)
        print(name, tuple(ls.shape), [k for k, _ in draws])
    np.savez_compressed(os.path.join(out, "spectro_modes.npz"), **d)
    print("spectro_modes.npz", len(d), "arrays")


def gen_evaltail(out):
    """util.imdct + compute_matrics (util/util.py:104-184) as generate_audio.py:40-49 calls them."""
    import util.util as U
    from models.mdct import IMDCT4
    d = {}
    g = torch.Generator().manual_seed(4321)
    torch.Tensor.cuda = lambda self, *a, **k: self          # compute_matrics hard-codes kbdwin(...).cuda() (util.py:178)
    # --- imdct: explicit and plain encodings, up_ratio 1 / 3 / 6 (6 -> int(H/6) is not a divisor of H)
    n_fft, hop, H, W = 64, 32, 32, 9
    for tag, B, explicit, up in (("ex_b2_u6", 2, True, 6.0), ("ex_b1_u3", 1, True, 3.0), ("ex_b2_u1", 2, True, 1.0),
                                 ("pl_b2_u1", 2, False, 1.0), ("pl_b2_u6", 2, False, 6.0)):
        C = 2 if explicit else 1
        spectro = torch.rand(B, C, H, W, generator=g)
        if explicit:
            spectro[..., 3, 2] = spectro[..., :1, 3, 2]        # equal channels -> sign 0
        pha = torch.sign(torch.randn(B, H, W, generator=g))
        norm = {"min": torch.tensor(-140.0), "max": torch.tensor(-35.5)}
        _imdct = IMDCT4(n_fft=n_fft, hop_length=hop, win_length=n_fft, window=U.kbdwin, out_length=(W - 1) * hop, device="cpu")
        if not explicit and up > 1:
            draws = []
            real_randint = torch.randint

            def fake_randint(low=0, high=2, size=None, device=None, **kw):
                r = real_randint(low=low, high=high, size=tuple(size), generator=g)
                draws.append(2 * r - 1)
                return r
            torch.randint = fake_randint
            audio = U.imdct(spectro if explicit else spectro, pha.unsqueeze(1), norm, _imdct, up_ratio=up, explicit_encoding=explicit)
            torch.randint = real_randint
            d[f"imdct_{tag}_pseudo"] = _np(draws[0].to(torch.float32))
        else:
            audio = U.imdct(spectro, pha if explicit else pha.unsqueeze(1), norm, _imdct, up_ratio=up, explicit_encoding=explicit)
        d[f"imdct_{tag}_spectro"] = _np(spectro)
        d[f"imdct_{tag}_pha"] = _np(pha)
        d[f"imdct_{tag}_audio"] = _np(audio)
        d[f"imdct_{tag}_meta"] = np.array([n_fft, hop, H, W, int(explicit), up, -140.0, -35.5])
    # --- compute_matrics
    class O:
        pass
    for tag, B, T, N in (("n64_b3", 3, 1000, 64), ("n64_1d", 0, 777, 64), ("n1024_b2", 2, 9000, 1024), ("n64_nc", 2, 1000, 64)):
        o = O(); o.n_fft = N; o.hop_length = N // 2; o.win_length = N; o.center = tag != "n64_nc"; o.hr_sampling_rate = 48000
        shape = (T,) if B == 0 else (B, T)
        hr = 0.1 * torch.randn(*shape, generator=g)
        lr = hr + 0.03 * torch.randn(*shape, generator=g)
        sr = 0.7 * hr + 0.02 * torch.randn(*shape, generator=g) + 0.01
        mse, snr_sr, snr_lr, _, _, _, lsd = U.compute_matrics(hr, lr, sr, o)
        d[f"met_{tag}_hr"] = _np(hr); d[f"met_{tag}_lr"] = _np(lr); d[f"met_{tag}_sr"] = _np(sr)
        d[f"met_{tag}_out"] = np.array([mse, snr_sr, snr_lr, lsd])
        d[f"met_{tag}_meta"] = np.array([N, N // 2, N, int(o.center)])
        d[f"met_{tag}_win2"] = _np(U.kbdwin(2 * N))
    d["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(out, "evaltail.npz"), **d)
    print("evaltail.npz", len(d), "arrays")


def gen_feeder(out):
    """seg_pad_audio of both datasets (data/audio_dataset.py:81-88,124-135) called on a bare object, plus an excerpt of
    the reference's own test clip (test/test.wav, PCM16 mono 48 kHz) as the feeder's real-audio fixture."""
    import wave
    from data.audio_dataset import AudioDataset, AudioTestDataset
    d = {}
    g = torch.Generator().manual_seed(99)

    class Bare:
        pass
    for tag, shape, seg in (("train_long", (1, 50), 32), ("train_exact", (1, 32), 32), ("train_short", (1, 20), 32),
                            ("train_stereo_long", (2, 40), 32)):
        o = Bare(); o.segment_length = seg
        w = torch.randn(*shape, generator=g)
        d[f"seg_{tag}_in"] = _np(w); d[f"seg_{tag}_out"] = _np(AudioDataset.seg_pad_audio(o, w)); d[f"seg_{tag}_len"] = np.array(seg)
    for tag, shape, seg in (("test_multi", (1, 100), 32), ("test_exact", (1, 64), 32), ("test_short", (1, 20), 32), ("test_1d", (70,), 32)):
        o = Bare(); o.segment_length = seg
        w = torch.randn(*shape, generator=g)
        d[f"seg_{tag}_in"] = _np(w); d[f"seg_{tag}_out"] = _np(AudioTestDataset.seg_pad_audio(o, w)); d[f"seg_{tag}_len"] = np.array(seg)
    with wave.open(os.path.join(REF, "test", "test.wav")) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (1, 2, 48000)
        w.setpos(96000)
        d["test_wav_excerpt_i16"] = np.frombuffer(w.readframes(24000), dtype="<i2").copy()
    d["test_wav_rate"] = np.array(48000)
    np.savez_compressed(os.path.join(out, "feeder.npz"), **d)
    print("feeder.npz", len(d), "arrays")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    _stub_modules()
    sys.path.insert(0, REF)
    torch.set_num_threads(4)
    todo = a.only.split(",") if a.only else ["mdct", "mdct2", "networks", "networks_d3", "model", "spectro_modes", "evaltail", "feeder"]
    if "mdct2" in todo:
        gen_mdct2(a.out)
    if "mdct" in todo:
        gen_mdct(a.out)
    if "networks" in todo:
        gen_networks(a.out)
    if "networks_d3" in todo:
        gen_networks_d3(a.out)
    if "model" in todo:
        gen_model(a.out)
    if "spectro_modes" in todo:
        gen_spectro_modes(a.out)
    if "evaltail" in todo:
        gen_evaltail(a.out)
    if "feeder" in todo:
        gen_feeder(a.out)


if __name__ == "__main__":
    main()
