"""Oracle: spectrogram codec + the two-phase GAN training step, torch-CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates (paths relative to the reference checkout):

  amplitude_to_DB / DB_to_amplitude   torchaudio.functional (absent from the image; formulae
                                      pinned by test/metrics_test.ipynb cell 11)
  to_spectro       models/pix2pixHD_model.py:142-227  (explicit and single-channel encoding, every mask / phase mode)
  denormalize      models/pix2pixHD_model.py:229-232
  to_audio         models/pix2pixHD_model.py:234-249
  forward (losses) models/pix2pixHD_model.py:331-435
  step             train.py:148-184  (G step then D step, Adam(lr, (beta1, 0.999)))

The random noise of the mask (pix2pixHD_model.py:202) is an *input* here: parity tests inject
the tensor, not the seed.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch

from . import mdct4 as M
from . import networks as N


def default_opt(**kw):
    """Option names/defaults the hot path reads (options/base_options.py, train_options.py)."""
    o = dict(n_fft=1024, hop_length=512, win_length=1024, center=True,
             hr_sampling_rate=48000, lr_sampling_rate=8000,
             input_nc=2, output_nc=2, ngf=48, netG="global", n_downsample_global=4,
             n_blocks_global=9, n_local_enhancers=1, n_blocks_local=3,
             ndf=64, n_layers_D=3, num_D=2, no_ganFeat_loss=False, no_lsgan=False,
             lambda_feat=10.0, explicit_encoding=True, alpha=0.6, min_value=1e-7,
             mask=True, mask_mode="mode2", lr=0.0002, beta1=0.5)
    o.update(kw)
    return SimpleNamespace(**o)


def amplitude_to_DB(x, multiplier, amin, db_multiplier):
    """torchaudio.functional.amplitude_to_DB with top_db=None:
    multiplier*log10(clamp(x, amin)) - multiplier*db_multiplier.
    KAT: amplitude_to_DB([1..6], 20, 1e-7, 1) = [-20, -13.9794, -10.4576, -7.9588, -6.0206, -4.4370]."""
    return multiplier * torch.log10(torch.clamp(x, min=amin)) - multiplier * db_multiplier


def DB_to_amplitude(x, ref, power):
    """torchaudio.functional.DB_to_amplitude: ref * (10^(0.1 x))^power."""
    return ref * torch.pow(torch.pow(10.0, 0.1 * x), power)


def mdct4_torch(audio, opt, window):
    spec = M.mdct4_forward(audio.numpy(), opt.n_fft, opt.hop_length, opt.win_length, window, opt.center)
    return torch.from_numpy(spec)


def to_spectro(audio, opt, window, mask=False, noise=None, phase_noise=None, noise_sign=None):
    """pix2pixHD_model.py:142-227.  The MDCT output (fp64 in the reference) is cast to fp32
    right after the transform -- the documented choice of this build.  The tensors the reference draws
    inside (`noise`: randn of :202, `noise_sign`: 2 * randint - 1 of :215, `phase_noise`: rand / randn of :180-188)
    are arguments here."""
    up_ratio = opt.hr_sampling_rate / opt.lr_sampling_rate
    spectro = mdct4_torch(audio, opt, window).to(torch.float32)
    spectro = spectro.unsqueeze(1).permute(0, 1, 3, 2)                     # :147
    explicit = getattr(opt, "explicit_encoding", True)
    if explicit:
        neg = 0.5 * (torch.abs(spectro) - spectro)                          # :150
        pos = spectro + neg
        a = opt.alpha
        log_spectro = torch.cat((amplitude_to_DB(a * pos + (1 - a) * neg, 20, opt.min_value, 1),
                                 amplitude_to_DB((1 - a) * pos + a * neg, 20, opt.min_value, 1)), dim=1)
    else:
        log_spectro = amplitude_to_DB(torch.abs(spectro) + opt.min_value, 20, opt.min_value, 1)   # :160-162
    pha = torch.sign(spectro)                                               # :163
    mean = log_spectro.mean()
    std = log_spectro.var().sqrt()
    amax = log_spectro.max()
    amin = log_spectro.min()
    if not explicit:                                                        # :178-191
        pem = getattr(opt, "phase_encoding_mode", None)
        if pem == "uni_dist":
            pha = pha * phase_noise
        elif pem == "norm_dist":
            pha = pha * ((phase_noise - phase_noise.min()) / (phase_noise.max() - phase_noise.min()))
        elif pem == "norm_dist2":
            pha = pha * phase_noise.abs()
        elif pem == "scale":
            pha = pha * 0.5
    log_spectro = (log_spectro - amin) / (amax - amin)                      # :193
    if mask:
        size = log_spectro.size()
        mask_size = int(size[2] * (1 - 1 / up_ratio))                       # :199
        if opt.mask_mode in ("mode0", "mode1", "mode2"):
            assert noise is not None and tuple(noise.shape) == (size[0], size[1], mask_size, size[3])
            nmin, nmax = noise.min(), noise.max()
            if opt.mask_mode == "mode0":
                fill = noise / (nmax - nmin)                                # :209
            else:
                fill = (noise - nmin) / (nmax - nmin)                       # :213 / :219
                if opt.mask_mode == "mode1":
                    fill = fill * noise_sign                                # :215-216
        elif opt.mask_mode is None:
            fill = torch.zeros(size[0], size[1], mask_size, size[3])
        else:
            raise NotImplementedError(opt.mask_mode)
        log_spectro = torch.cat((log_spectro[:, :, :-mask_size, :], fill), dim=2)
    return log_spectro, pha, {"max": amax, "min": amin, "mean": mean, "std": std}


def denormalize(log_spectro, norm_param, opt):
    s = torch.abs(log_spectro) * (norm_param["max"] - norm_param["min"]) + norm_param["min"]
    return DB_to_amplitude(s, 10, 0.5) - opt.min_value                      # :232


def to_audio(log_spectro, norm_param, opt, window, pha=None, pseudo_pha=None):
    """pix2pixHD_model.py:234-249 with IMDCT4 as the inverse transform."""
    up_ratio = opt.hr_sampling_rate / opt.lr_sampling_rate
    s = denormalize(log_spectro, norm_param, opt)
    if getattr(opt, "explicit_encoding", True):
        s = (s[..., 0, :, :] - s[..., 1, :, :]) / (2 * opt.alpha - 1)       # :237
    else:
        if up_ratio > 1:                                                    # :239-243 (pseudo_pha: the randint of :240)
            size = pha.size(-2)
            keep = int(size * (1 / up_ratio))
            pha = torch.cat((pha[..., :keep, :], pseudo_pha[..., keep:, :].to(pha.dtype)), dim=-2)
            s = s * pha
        s = s.squeeze(1)                                                    # :248
    audio = M.imdct4_forward(s.permute(0, 2, 1).contiguous().numpy(), opt.n_fft, opt.hop_length,
                             opt.win_length, window, opt.center)
    return math.sqrt(up_ratio - 1) * torch.from_numpy(audio)


def netG_forward(pG, x, opt):
    if opt.netG == "global":
        return N.global_generator_forward(pG, x, opt.n_downsample_global, opt.n_blocks_global)
    return N.local_enhancer_forward(pG, x, opt.n_downsample_global, opt.n_blocks_global,
                                    opt.n_local_enhancers, opt.n_blocks_local)


def netG_spec(opt):
    if opt.netG == "global":
        return N.global_generator_spec(opt.input_nc, opt.output_nc, opt.ngf, opt.n_downsample_global,
                                       opt.n_blocks_global)
    return N.local_enhancer_spec(opt.input_nc, opt.output_nc, opt.ngf, opt.n_downsample_global,
                                 opt.n_blocks_global, opt.n_local_enhancers, opt.n_blocks_local)


def netD_spec(opt):
    return N.multiscale_discriminator_spec(opt.input_nc + opt.output_nc, opt.ndf, opt.n_layers_D,
                                           opt.num_D, not opt.no_ganFeat_loss)


def losses(pG, pD, lr_spectro, hr_spectro, opt):
    """pix2pixHD_model.py:348-398 -> dict of the five scalar losses + sr."""
    gi = not opt.no_ganFeat_loss
    D = lambda x: N.multiscale_discriminator_forward(pD, x, opt.ndf, opt.n_layers_D, opt.num_D, gi)
    sr = netG_forward(pG, lr_spectro, opt)
    pred_fake_pool = D(torch.cat((lr_spectro, sr.detach()), dim=1))         # :351
    loss_D_fake = N.gan_loss(pred_fake_pool, False)
    pred_real = D(torch.cat((lr_spectro, hr_spectro), dim=1))               # :355
    loss_D_real = N.gan_loss(pred_real, True)
    pred_fake = D(torch.cat((lr_spectro, sr), dim=1))                       # :360
    loss_G_GAN = N.gan_loss(pred_fake, True)
    loss_G_GAN_Feat = 0
    if gi:
        loss_G_GAN_Feat = N.feature_matching_loss(pred_fake, pred_real, opt.n_layers_D, opt.num_D, opt.lambda_feat)
    return dict(G_GAN=loss_G_GAN, G_GAN_Feat=loss_G_GAN_Feat, D_real=loss_D_real, D_fake=loss_D_fake, sr=sr)


def step_grads(pG, pD, lr_spectro, hr_spectro, opt):
    """train.py:155-184 without the optimiser: loss_G.backward() then loss_D.backward().

    Returns (losses, gradG, gradD).  gradD is what optimizer_D sees: train.py:176 zeroes the
    D grads accumulated by the G backward before loss_D.backward()."""
    pG = {k: v.detach().clone().requires_grad_(True) for k, v in pG.items()}
    pD = {k: v.detach().clone().requires_grad_(True) for k, v in pD.items()}
    L = losses(pG, pD, lr_spectro, hr_spectro, opt)
    loss_D = (L["D_fake"] + L["D_real"]) * 0.5                              # train.py:158
    loss_G = L["G_GAN"] + L["G_GAN_Feat"]                                   # train.py:159
    gG = torch.autograd.grad(loss_G, list(pG.values()), retain_graph=True)
    gD = torch.autograd.grad(loss_D, list(pD.values()))
    out = {k: (float(v) if k != "sr" else v.detach()) for k, v in L.items()}
    return out, dict(zip(pG.keys(), gG)), dict(zip(pD.keys(), gD))


def adam_step(params, grads, state, lr, beta1, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults as the reference constructs it (pix2pixHD_model.py:131,140)."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    for k in params:
        m = state.setdefault("m", {}).setdefault(k, torch.zeros_like(params[k]))
        v = state.setdefault("v", {}).setdefault(k, torch.zeros_like(params[k]))
        g = grads[k]
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        bc1 = 1 - beta1 ** t
        bc2 = 1 - beta2 ** t
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        params[k] = params[k] - (lr / bc1) * m / denom
    return params


def synthetic_batch(B, opt, seed=1234):
    """hr = 0.1*randn(B, T), lr = second draw; T = (frames-1)*hop with frames = n_fft/4."""
    frames = opt.n_fft // 4
    T = (frames - 1) * opt.hop_length
    g = torch.Generator().manual_seed(seed)
    hr = 0.1 * torch.randn(B, T, generator=g)
    lr = 0.1 * torch.randn(B, T, generator=g)
    bins = opt.n_fft // 2
    up = opt.hr_sampling_rate / opt.lr_sampling_rate
    mask_size = int(bins * (1 - 1 / up))
    noise = torch.randn(B, 2, mask_size, frames, generator=g)
    return hr, lr, noise


def full_step(hr, lr, noise, pG, pD, opt, window, stateG, stateD):
    """One reference-faithful optimisation step on the CPU (train.py:148-184): MDCT4 encode of hr and lr,
    G forward, three D forwards, LSGAN + feature-matching losses, both backward passes, both Adam updates.
    Used as the `cpu_baseline` leg of bench.py and by tests; returns (losses, pG, pD)."""
    with torch.no_grad():
        hr_s, _, _ = to_spectro(hr, opt, window, mask=False)
        lr_s, _, _ = to_spectro(lr, opt, window, mask=opt.mask, noise=noise)
    L, gG, gD = step_grads(pG, pD, lr_s, hr_s, opt)
    pG = adam_step(dict(pG), gG, stateG, opt.lr, opt.beta1)
    pD = adam_step(dict(pD), gD, stateD, opt.lr, opt.beta1)
    return L, pG, pD


def to_frames_mdct2(log_spectro, norm_param, opt):
    """pix2pixHD_model.py:251-258 with IDCT_2N_native as the frame operator (oracle/mdct2.py)."""
    from . import mdct2 as M2
    s = denormalize(log_spectro, norm_param, opt)
    s = (s[..., 0, :, :] - s[..., 1, :, :]) / (2 * opt.alpha - 1)
    return torch.from_numpy(M2.idct_2n(s.permute(0, 2, 1).contiguous().double().numpy()))


def match_loss(sr, norm_param, opt, window, lambda_mat=10.0):
    """pix2pixHD_model.py:408-415."""
    half = opt.win_length // 2
    fr = to_frames_mdct2(sr, norm_param, opt)
    w = torch.as_tensor(window, dtype=torch.float64)
    a = fr[..., :-1, half:] * w[:half]
    b = fr[..., 1:, :half] * w[half:]
    return float(((a - b) ** 2).mean()) * lambda_mat
