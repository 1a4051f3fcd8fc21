"""Pin oracle/model.py against the reference Pix2PixHDModel run on CPU (tools/gen_golden.py):
dB codec KAT, to_spectro, to_audio, the four losses, both gradient sets and one Adam step."""
import numpy as np
import torch

from conftest import rel_err, assert_grad_close, noise_bias_keys
from oracle import model as OM
from oracle import networks as N


def _opt():
    return OM.default_opt(n_fft=64, hop_length=32, win_length=64, ngf=8, netG="global", n_downsample_global=2,
                          n_blocks_global=2, ndf=8, n_layers_D=3, num_D=2)


def test_db_codec_kat():
    """test/metrics_test.ipynb cell 11."""
    x = torch.arange(1, 7, dtype=torch.float32)
    db = OM.amplitude_to_DB(x, 20, 1e-7, 1)
    ref = torch.tensor([-20.0, -13.9794, -10.4576, -7.9588, -6.0206, -4.4370])
    assert torch.allclose(db, ref, atol=1e-4)
    assert torch.allclose(OM.DB_to_amplitude(db, 10, 0.5), x, rtol=1e-5)


def test_to_spectro_and_audio(golden_model):
    g = golden_model
    opt = _opt()
    w = g["window"]
    hs, hpha, hn = OM.to_spectro(torch.from_numpy(g["hr"]), opt, w, mask=False)
    assert rel_err(hs.numpy(), g["hr_spectro"]) < 1e-5
    assert np.array_equal(hpha.numpy(), g["hr_pha"])
    assert abs(float(hn["max"]) - float(g["hr_max"])) < 1e-4 and abs(float(hn["min"]) - float(g["hr_min"])) < 1e-4
    ls, lpha, ln = OM.to_spectro(torch.from_numpy(g["lr"]), opt, w, mask=True, noise=torch.from_numpy(g["mask_noise"]))
    assert ls.shape == g["lr_spectro"].shape
    assert rel_err(ls.numpy(), g["lr_spectro"]) < 1e-5
    aud = OM.to_audio(torch.from_numpy(g["hr_spectro"]), {"max": torch.tensor(float(g["hr_max"])), "min": torch.tensor(float(g["hr_min"]))}, opt, w)
    assert aud.shape == g["hr_audio_rt"].shape
    assert rel_err(aud.numpy(), g["hr_audio_rt"]) < 1e-5


def test_losses_grads_and_adam(golden_model):
    g = golden_model
    opt = _opt()
    pG = {k: torch.from_numpy(g[f"G_p_{k}"]) for k in OM.netG_spec(opt)}
    pD = {k: torch.from_numpy(g[f"D_p_{k}"]) for k in OM.netD_spec(opt)}
    lr_s = torch.from_numpy(g["lr_spectro"])
    hr_s = torch.from_numpy(g["hr_spectro"])
    L, gG, gD = OM.step_grads(pG, pD, lr_s, hr_s, opt)
    ref = dict(zip([str(n) for n in g["loss_names"]], g["loss_values"]))
    for k in ("G_GAN", "G_GAN_Feat", "D_real", "D_fake"):
        assert abs(L[k] - ref[k]) < 2e-5 * max(1.0, abs(ref[k])), k
    assert rel_err(L["sr"].numpy(), g["sr"]) < 2e-5
    for k, v in gG.items():
        assert_grad_close(k, v.numpy(), g[f"G_g_{k}"], rtol=2e-4, noise_biases=noise_bias_keys(list(gG)))
    for k, v in gD.items():
        assert_grad_close(k, v.numpy(), g[f"D_g_{k}"], rtol=2e-4, noise_biases=noise_bias_keys(list(gD)))
    # one Adam step from the reference's own gradients reproduces its updated weights
    newG = OM.adam_step({k: v.clone() for k, v in pG.items()}, {k: torch.from_numpy(g[f"G_g_{k}"]) for k in pG},
                        {}, opt.lr, opt.beta1)
    for k in pG:
        assert np.max(np.abs(newG[k].numpy() - g[f"G_p1_{k}"])) < 2e-6, k


def test_to_spectro_other_encodings_and_mask_modes():
    """mask_mode mode0 / mode1 and the single-channel encoding with every phase_encoding_mode
    (pix2pixHD_model.py:159-162,178-191,207-221,238-249) against the reference's own outputs."""
    import _spectro_mode_cases as SC
    g = SC.load()
    w = OM.M.kbdwin(64)
    lr = torch.from_numpy(g["lr"])
    for name, kw in SC.CASES.items():
        opt = _opt()
        for k, v in {**dict(explicit_encoding=True, phase_encoding_mode=None, mask_mode="mode2"), **kw}.items():
            setattr(opt, k, v)
        pn, noise, sgn, pseudo = SC.draws(g, name)
        ls, pha, nrm = OM.to_spectro(lr, opt, w, mask=True, noise=noise, phase_noise=pn, noise_sign=sgn)
        assert ls.shape == g[f"{name}_spectro"].shape, name
        assert rel_err(ls.numpy(), g[f"{name}_spectro"]) < 1e-5, name
        assert rel_err(pha.numpy(), g[f"{name}_pha"]) < 1e-6, name
        # the single-channel minimum is the dB value of the smallest |bin| (~1e-7 here, not clamped): its relative rounding
        # error in the fp32 cast of the transform is amplified by the logarithm
        assert abs(float(nrm["max"]) - float(g[f"{name}_max"])) < 1e-4 and abs(float(nrm["min"]) - float(g[f"{name}_min"])) < 5e-3
        if not opt.explicit_encoding:
            aud = OM.to_audio(torch.from_numpy(g[f"{name}_spectro"]), {"max": torch.tensor(float(g[f"{name}_max"])),
                                                                     "min": torch.tensor(float(g[f"{name}_min"]))},
                              opt, w, pha=torch.from_numpy(g[f"{name}_pha"]), pseudo_pha=pseudo)
            assert aud.shape == g[f"{name}_audio"].shape and rel_err(aud.numpy(), g[f"{name}_audio"]) < 1e-5, name
