"""GPU parity of the whole hot path behind Pix2PixHDModel against the reference model run on CPU
(tests/golden/model_step.npz from tools/gen_golden.py): to_spectro, to_audio, the four losses, generator
output, both gradient sets (train.py:155-184 order) and the weights after one Adam step of each optimiser."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import rel_err, assert_grad_close, noise_bias_keys  # noqa: F401

pytestmark = pytest.mark.gpu


def make_opt(**kw):
    o = dict(gpu_ids=[0], isTrain=True, checkpoints_dir="/tmp/p2phd_test_ckpt", name="t", model="pix2pixHD",
             input_nc=2, output_nc=2, label_nc=0, hr_sampling_rate=48000, lr_sampling_rate=8000,
             n_fft=64, hop_length=32, win_length=64, center=True, no_instance=True, ngf=8, netG="global",
             n_downsample_global=2, n_blocks_global=2, n_local_enhancers=1, n_blocks_local=1, norm="instance",
             no_lsgan=False, ndf=8, n_layers_D=3, num_D=2, no_ganFeat_loss=False, use_hifigan_D=False, use_time_D=False,
             verbose=False, continue_train=False, load_pretrain="", which_epoch="latest", pool_size=0, lr=0.0002,
             beta1=0.5, no_vgg_loss=True, use_match_loss=False, niter_fix_global=0, explicit_encoding=True, alpha=0.6,
             min_value=1e-7, mask=True, mask_mode="mode2", phase_encoding_mode=None, lambda_feat=10.0, fp16=False, niter_decay=100,
             instance_feat=False, label_feat=False)
    o.update(kw)
    return SimpleNamespace(**o)


def _model(g, **kw):
    from pix2pixhdaudiosr_amd.models.models import create_model
    model = create_model(make_opt(**kw))
    for net, tag in ((model.netG, "G"), (model.netD, "D")):
        sd = {k: torch.from_numpy(g[f"{tag}_p_{k}"]) for k in net.state_dict().keys()}
        net.load_state_dict(sd)
    from pix2pixhdaudiosr_amd import _ops
    _ops.bump_weight_epoch()
    return model


def test_to_spectro_and_audio(golden_model):
    g = golden_model
    m = _model(g)
    hs, hpha, hn = m.to_spectro(torch.from_numpy(g["hr"]), mask=False)
    assert tuple(hs.shape) == g["hr_spectro"].shape
    assert rel_err(hs.cpu().numpy(), g["hr_spectro"]) < 1e-4
    assert np.mean(hpha.cpu().numpy() != g["hr_pha"]) < 1e-3          # sign of near-zero bins may flip in fp32
    assert abs(float(hn["max"]) - float(g["hr_max"])) < 1e-3 and abs(float(hn["min"]) - float(g["hr_min"])) < 1e-3
    ls, lpha, ln = m.to_spectro(torch.from_numpy(g["lr"]), mask=True, noise=torch.from_numpy(g["mask_noise"]))
    assert tuple(ls.shape) == g["lr_spectro"].shape
    assert rel_err(ls.cpu().numpy(), g["lr_spectro"]) < 1e-4
    aud = m.to_audio(torch.from_numpy(g["hr_spectro"]), {"max": torch.tensor(float(g["hr_max"])), "min": torch.tensor(float(g["hr_min"]))})
    assert tuple(aud.shape) == g["hr_audio_rt"].shape
    assert rel_err(aud.cpu().numpy(), g["hr_audio_rt"]) < 1e-4


@pytest.mark.parametrize("name", ["e_mode0", "e_mode1", "p_none_mode2", "p_uni_nomask", "p_norm_mode0", "p_norm2_mode1", "p_scale_mode2"])
def test_to_spectro_other_encodings_and_mask_modes(golden_model, name):
    """mask_mode mode0 / mode1 and the single-channel encoding with every phase_encoding_mode, to_audio included
    (pix2pixHD_model.py:159-162,178-191,207-221,238-249), against the reference's outputs on the random tensors it drew."""
    import _spectro_mode_cases as SC
    g = SC.load()
    kw = {**dict(explicit_encoding=True, phase_encoding_mode=None, mask_mode="mode2"), **SC.CASES[name]}
    m = _model(golden_model, **kw)
    pn, noise, sgn, pseudo = SC.draws(g, name)
    ls, pha, nrm = m.to_spectro(torch.from_numpy(g["lr"]), mask=True, noise=noise, phase_noise=pn, noise_sign=sgn)
    assert tuple(ls.shape) == g[f"{name}_spectro"].shape
    assert rel_err(ls.cpu().numpy(), g[f"{name}_spectro"]) < 1e-4
    # signs of near-zero bins may flip in fp32; elsewhere pha (sign x phase noise) agrees
    assert np.mean(np.abs(pha.cpu().numpy() - g[f"{name}_pha"]) > 1e-5) < 2e-3
    assert abs(float(nrm["max"]) - float(g[f"{name}_max"])) < 1e-3 and abs(float(nrm["min"]) - float(g[f"{name}_min"])) < 2e-2
    if not kw["explicit_encoding"]:
        aud = m.to_audio(torch.from_numpy(g[f"{name}_spectro"]), {"max": torch.tensor(float(g[f"{name}_max"])),
                                                                "min": torch.tensor(float(g[f"{name}_min"]))},
                         pha=torch.from_numpy(g[f"{name}_pha"]), pseudo_pha=pseudo)
        assert tuple(aud.shape) == g[f"{name}_audio"].shape and rel_err(aud.cpu().numpy(), g[f"{name}_audio"]) < 1e-4
    # without handed-in tensors the draws happen on the device: shapes and range only
    ls2, pha2, _ = m.to_spectro(torch.from_numpy(g["lr"]), mask=True)
    assert ls2.shape == ls.shape and torch.isfinite(ls2).all() and float(ls2.abs().max()) <= 1.0 + 1e-5


def test_forward_losses_grads_and_step(golden_model):
    g = golden_model
    m = _model(g)
    assert m.loss_names == [str(n) for n in g["loss_names"]]
    losses, sr = m.forward(torch.from_numpy(g["lr"]), None, torch.from_numpy(g["hr"]), None, infer=True,
                           noise=torch.from_numpy(g["mask_noise"]))
    ref = dict(zip(m.loss_names, g["loss_values"]))
    got = dict(zip(m.loss_names, losses))
    for k in m.loss_names:
        assert abs(float(got[k]) - ref[k]) < 2e-4 * max(1.0, abs(ref[k])), (k, float(got[k]), ref[k])
    assert rel_err(sr.detach().cpu().numpy(), g["sr"]) < 1e-4
    # train.py:155-184
    loss_D = (got["D_fake"] + got["D_real"]) * 0.5
    loss_G = got["G_GAN"] + got["G_GAN_Feat"]
    m.optimizer_G.zero_grad(); loss_G.backward(); 
    nbG = noise_bias_keys([k for k, _ in m.netG.named_parameters()])
    nbD = noise_bias_keys([k for k, _ in m.netD.named_parameters()])
    assert len(nbD) == 2 * 3                                      # 2 scales x (5 convs - first - last)
    for k, p in m.netG.named_parameters():
        assert_grad_close("G:" + k, p.grad.cpu().numpy(), g[f"G_g_{k}"], rtol=5e-4, noise_biases=nbG)
    m.optimizer_G.step()
    m.optimizer_D.zero_grad(); loss_D.backward()
    for k, p in m.netD.named_parameters():
        assert_grad_close("D:" + k, p.grad.cpu().numpy(), g[f"D_g_{k}"], rtol=5e-4, noise_biases=nbD)
    m.optimizer_D.step()
    # Adam moves every weight by ~lr on step 1 (sign-like update): compare the update direction where the
    # reference gradient is clearly non-zero, and the magnitude everywhere
    for tag, net in (("G", m.netG), ("D", m.netD)):
        for k, p in net.state_dict().items():
            new_ref, old = g[f"{tag}_p1_{k}"], g[f"{tag}_p_{k}"]
            d_ref, d_got = new_ref - old, p.cpu().numpy() - old
            assert np.max(np.abs(d_got)) <= 2.0001e-4 + 1e-7
            gr = g[f"{tag}_g_{k}"]
            strong = np.abs(gr) > 1e-3 * max(np.abs(gr).max(), 1e-12)
            if strong.any() and not (k.endswith(".bias")):
                assert np.mean(np.sign(d_got[strong]) == np.sign(d_ref[strong])) > 0.999, k


def test_train_step_matches_manual_order(golden_model):
    """model.train_step (G all-reduce overlapped with the D backward) == train.py's order on 1 GPU."""
    g = golden_model
    a, b = _model(g), _model(g)
    lr, hr, noise = (torch.from_numpy(g[k]) for k in ("lr", "hr", "mask_noise"))
    a.train_step(lr, hr, noise=noise)
    losses, _ = b.forward(lr, None, hr, None, noise=noise)
    ld = dict(zip(b.loss_names, losses))
    b.optimizer_G.zero_grad(); (ld["G_GAN"] + ld["G_GAN_Feat"]).backward(); b.optimizer_G.step()
    b.optimizer_D.zero_grad(); ((ld["D_fake"] + ld["D_real"]) * 0.5).backward(); b.optimizer_D.step()
    # Float atomics (InstanceNorm sums, split-K) make gradients differ in the last bits between two runs; Adam's first
    # step is sign-like, so an element whose gradient is pure rounding noise (every conv bias in front of an
    # InstanceNorm) may move by +lr in one run and -lr in the other.  Weights must agree except for such elements.
    for net_a, net_b in ((a.netG, b.netG), (a.netD, b.netD)):
        for (k, pa), (_, pb) in zip(net_a.state_dict().items(), net_b.state_dict().items()):
            if k.endswith(".weight"):
                assert float(((pa - pb).abs() > 1e-6).float().mean()) < 2e-3, k


def test_bf16_step_runs_and_tracks_fp32(golden_model):
    g = golden_model
    m = _model(g, fp16=True)
    lr, hr, noise = (torch.from_numpy(g[k]) for k in ("lr", "hr", "mask_noise"))
    losses, sr = m.forward(lr, None, hr, None, infer=True, noise=noise)
    ref = dict(zip(m.loss_names, g["loss_values"]))
    for k, v in zip(m.loss_names, losses):
        assert abs(float(v) - ref[k]) < 0.1 * max(1.0, abs(ref[k])), (k, float(v), ref[k])     # bf16 activations
    assert rel_err(sr.detach().cpu().numpy(), g["sr"]) < 0.1
    m.train_step(lr, hr, noise=noise)
    assert all(torch.isfinite(p).all() for p in m.netG.parameters())


def test_fp16_storage_step_with_the_device_loss_scaler(golden_model):
    """opt.fp16_storage: IEEE fp16 activations (the reference's autocast type, train.py:62-67) through the fp16 build of the
    library, with optim.DeviceGradScaler in the place of torch.cuda.amp.GradScaler.  (1) the forward tracks the fp32 golden
    values closer than bf16 does; (2) one train_step moves the weights as the unscaled bf16/fp32 rule would (the scale never
    reaches the update) and leaves the scale alone, growth tracker = 1; (3) a scale so large that the gradients overflow in fp16
    skips BOTH updates, does not advance the step counters and halves the scale; (4) the captured step does the same."""
    from pix2pixhdaudiosr_amd import _lib
    g = golden_model
    S0 = 1024.0        # these 8-channel nets on 2 clips have gradients ~1e2 x those of configs[1]: 65536 overflows (and backs off, as it should)
    m = _model(g, fp16=True, fp16_storage=True, mask=False, loss_scale=S0)
    assert m.compute_dtype == torch.float16 and m.scaler is not None and m.scaler.get_scale() == S0
    assert _lib.lib_for(torch.float16).p2phd_half_type() != _lib.lib_for(torch.bfloat16).p2phd_half_type()
    lr, hr = _fresh_audio(g)
    f32 = _model(g, mask=False)
    l32, sr32 = f32.forward(lr, None, hr, None, infer=True)
    losses, sr = m.forward(lr, None, hr, None, infer=True)
    for k, v, r in zip(m.loss_names, losses, l32):
        assert abs(float(v) - float(r)) < 1.5e-2 * max(1.0, abs(float(r))), (k, float(v), float(r))
    assert rel_err(sr.detach().float().cpu().numpy(), sr32.detach().cpu().numpy()) < 1.5e-2
    # (2) one scaled step against one fp32 step from the same weights: Adam's first step is sign-like, so compare directions
    w0 = m.optimizer_G.flat_p.clone()
    m.train_step(lr, hr)
    f32.train_step(lr, hr)
    st = m.scaler.state.cpu()
    assert float(st[0]) == S0 and float(st[2]) == 1.0 and float(st[3]) == 0.0 and float(st[4]) == 0.0, st
    assert m.optimizer_G.steps_taken() == 1 and m.optimizer_D.steps_taken() == 1
    d16, d32 = m.optimizer_G.flat_p - w0, f32.optimizer_G.flat_p - w0
    strong = _signal_mask(f32, f32.optimizer_G) & (f32.optimizer_G.flat_g.abs() > 1e-2 * f32.optimizer_G.flat_g.abs().max())
    assert float((torch.sign(d16[strong]) == torch.sign(d32[strong])).float().mean()) > 0.99
    assert float(d16.abs().max()) <= 2e-4 * 1.001                   # |update| <= lr: the scale did not leak into the step
    # (3) overflow: skip, back off
    m.scaler.state[0] = 2.0 ** 60; m.scaler.state[1] = 2.0 ** -60
    wG, wD = m.optimizer_G.flat_p.clone(), m.optimizer_D.flat_p.clone()
    m.train_step(lr, hr)
    assert torch.equal(wG, m.optimizer_G.flat_p) and torch.equal(wD, m.optimizer_D.flat_p)
    assert m.optimizer_G.steps_taken() == 1 and m.optimizer_D.steps_taken() == 1
    assert m.scaler.get_scale() == 2.0 ** 59 and float(m.scaler.state[2]) == 0.0
    # (4) the captured step: same scaler state machine, replayed
    m.scaler.state[0] = S0; m.scaler.state[1] = 1.0 / S0
    for _ in range(4):
        out = m.train_step_graphed(lr, hr)
    assert m._graph_state['graphs'] is not None
    assert all(np.isfinite(float(v)) for v in out.values())
    assert m.optimizer_G.steps_taken() == 5 and float(m.scaler.state[2]) == 4.0 and m.scaler.get_scale() == S0
    m.scaler.state[0] = 2.0 ** 60; m.scaler.state[1] = 2.0 ** -60
    wG = m.optimizer_G.flat_p.clone()
    m.train_step_graphed(lr, hr)
    assert torch.equal(wG, m.optimizer_G.flat_p) and m.optimizer_G.steps_taken() == 5 and m.scaler.get_scale() == 2.0 ** 59


def test_unsupported_configs_raise():
    from pix2pixhdaudiosr_amd.models.models import create_model
    with pytest.raises(NotImplementedError):
        create_model(make_opt(use_time_D=True))                    # hifigan discriminator: source absent in the reference
    with pytest.raises(NotImplementedError):
        create_model(make_opt(mask_mode="mode7"))
    with pytest.raises(NotImplementedError):
        create_model(make_opt(norm="batch"))
    with pytest.raises(RuntimeError):
        create_model(make_opt(gpu_ids=[]))


def test_mdct2_model_and_match_loss():
    """mdct_type='mdct2' (the transform the shipped reference hard-codes): n_fft bins, to_audio round trip, to_frames and
    the TDAC matching loss against the oracle, and a finite optimisation step with --use_match_loss."""
    from pix2pixhdaudiosr_amd.models.models import create_model
    from oracle import model as OM, mdct2 as M2
    opt = make_opt(mdct_type="mdct2", use_match_loss=True, lambda_mat=10.0)
    torch.manual_seed(5)
    m = create_model(opt)
    assert m.loss_names == ['G_GAN', 'G_GAN_Feat', 'G_mat', 'D_real', 'D_fake']
    T = 15 * opt.hop_length
    gen = torch.Generator().manual_seed(11)
    hr = 0.1 * torch.randn(2, T, generator=gen)
    lr = 0.1 * torch.randn(2, T, generator=gen)
    hs, _, hn = m.to_spectro(hr, mask=False)
    assert tuple(hs.shape) == (2, 2, opt.n_fft, 16)                          # n_fft bins x frames
    w = m.window.cpu().numpy()
    ref = M2.mdct2_forward(hr.numpy(), opt.n_fft, opt.hop_length, opt.win_length, w)
    assert ref.shape == (2, 16, opt.n_fft)
    aud = m.to_audio(hs, hn).squeeze().cpu()
    # to_audio = sqrt(up_ratio - 1) * IMDCT2(decode(...)); MDCT2->IMDCT2 reconstructs x up to the codec's 1e-7 floor
    assert float(((aud / np.sqrt(m.up_ratio - 1) - hr) ** 2).mean()) < 1e-6
    losses, sr = m.forward(lr, None, hr, None, infer=True)
    got = dict(zip(m.loss_names, losses))
    _, _, ln = m.to_spectro(lr, mask=False)       # same min/max as the masked call inside forward
    oo = OM.default_opt(n_fft=opt.n_fft, hop_length=opt.hop_length, win_length=opt.win_length)
    want = OM.match_loss(sr.detach().cpu(), {"max": ln["max"].cpu(), "min": ln["min"].cpu()}, oo, w, 10.0)
    assert abs(float(got["G_mat"]) - want) < 2e-3 * max(1.0, abs(want)), (float(got["G_mat"]), want)
    m.train_step(lr, hr)
    assert all(torch.isfinite(p).all() for p in m.parameters())


def test_checkpoint_roundtrip_and_tolerant_load(tmp_path, golden_model):
    """save()/load_network with the reference's file names and keys (base_model.py:43-89): exact reload, a checkpoint
    with extra layers (tier 2) and one with a mismatching layer (tier 3) -- plus the linear LR decay of
    update_learning_rate (pix2pixHD_model.py:530-539)."""
    import os
    from pix2pixhdaudiosr_amd.models.models import create_model
    g = golden_model
    opt = make_opt(checkpoints_dir=str(tmp_path), name="ck")
    m = _model(g, checkpoints_dir=str(tmp_path), name="ck")
    m.save("latest")
    assert sorted(os.listdir(tmp_path / "ck")) == ["latest_net_D.pth", "latest_net_G.pth"]
    sd = torch.load(tmp_path / "ck" / "latest_net_G.pth")
    assert list(sd.keys()) == [k[4:] for k in g.files if k.startswith("G_p_")]          # reference key order
    assert all(v.dtype == torch.float32 and v.device.type == "cpu" for v in sd.values())
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), g[f"G_p_{k}"])
    # exact reload into a freshly initialised model
    m2 = create_model(make_opt(checkpoints_dir=str(tmp_path), name="ck", continue_train=True))
    for (k, a), (_, b) in zip(m.netG.state_dict().items(), m2.netG.state_dict().items()):
        assert torch.equal(a, b), k
    # tier 2: excessive layers in the file; tier 3: one tensor with a different shape stays at its initial value
    extra = dict(sd); extra["model.99.weight"] = torch.zeros(3)
    torch.save(extra, tmp_path / "ck" / "x_net_G.pth")
    broken = dict(sd); broken["model.1.weight"] = torch.zeros(1, 1, 1, 1)
    torch.save(broken, tmp_path / "ck" / "y_net_G.pth")
    m3 = create_model(make_opt(checkpoints_dir=str(tmp_path), name="ck"))
    init = m3.netG.state_dict()["model.1.weight"].clone()
    m3.load_network(m3.netG, "G", "x", str(tmp_path / "ck"))
    assert torch.equal(m3.netG.state_dict()["model.4.weight"].cpu(), sd["model.4.weight"])
    m3.load_network(m3.netG, "G", "y", str(tmp_path / "ck"))
    assert torch.equal(m3.netG.state_dict()["model.1.weight"].cpu(), sd["model.1.weight"])   # kept (shape mismatch ignored)
    # loaded weights are what the kernels see (packed copies are refreshed)
    x = torch.from_numpy(g["lr_spectro"]).cuda()
    assert rel_err(m2.netG(x).detach().cpu().numpy(), g["sr"]) < 1e-4
    # linear decay
    lr0 = m.optimizer_G.param_groups[0]["lr"]
    m.update_learning_rate()
    assert abs(m.optimizer_G.param_groups[0]["lr"] - (lr0 - opt.lr / opt.niter_decay)) < 1e-12
    assert m.optimizer_D.param_groups[0]["lr"] == m.optimizer_G.param_groups[0]["lr"]


def _fresh_audio(g, seed=123):
    """Random clips of the golden length for run-vs-run comparisons.  The golden clip itself is unsuitable there: one
    element of a discriminator feature of its generated spectrogram coincides with the real one to ~1e-7, so the sign of
    (fake - real) in the L1 feature-matching gradient -- a legitimate discontinuity -- flips with the rounding noise of
    the atomically summed InstanceNorm statistics and moves the whole generator gradient by 1.6 % between two runs of
    the SAME code (tools/probe_knife_edge.py shows the element)."""
    gen = torch.Generator().manual_seed(seed)
    T = g["hr"].shape[1]
    return (0.1 * torch.randn(2, T, generator=gen)).cuda(), (0.1 * torch.randn(2, T, generator=gen)).cuda()


def _signal_mask(model, opt):
    """Elements of an optimiser's flat buffers that carry signal: everything but the conv biases in front of an
    InstanceNorm (true gradient 0: run-to-run rounding noise of ~1e-2 on these tiny nets)."""
    net = model.netG if opt is model.optimizer_G else model.netD
    names = [k for k, _ in net.named_parameters()]
    nb = noise_bias_keys(names)
    mask = torch.ones(opt._total, dtype=torch.bool, device=opt.flat_g.device)
    for k, p, o in zip(names, opt._params, opt._offs):
        if k in nb:
            mask[o:o + p.numel()] = False
    return mask


def _grad_diff(model_a, model_b, which):
    oa = getattr(model_a, which); ob = getattr(model_b, which)
    m = _signal_mask(model_a, oa)
    return float((oa.flat_g[m] - ob.flat_g[m]).norm() / oa.flat_g[m].norm())


def _reset(model, g):
    """Golden weights, zeroed Adam state and step counters: both paths restart from the identical point."""
    from pix2pixhdaudiosr_amd import _ops
    for net, tag, opt in ((model.netG, "G", model.optimizer_G), (model.netD, "D", model.optimizer_D)):
        with torch.no_grad():
            for k, p in net.state_dict().items():
                p.copy_(torch.from_numpy(g[f"{tag}_p_{k}"]))
        opt.exp_avg.zero_(); opt.exp_avg_sq.zero_(); opt.step_dev.zero_(); opt.flat_g.zero_()
        opt.step_count = 0
    _ops.bump_weight_epoch()


def test_two_eager_steps_from_the_same_state_are_bit_identical(golden_model):
    """Run-to-run reproducibility of the whole step (verdict r2 item 3 iv): no float atomics are left on any quantity with
    a real value -- the InstanceNorm statistics (per-wave partials + Chan merge), the InstanceNorm-backward sums, the bias
    column sums, the split-K slabs of the weight gradients and the loss accumulators all add in a fixed order.  What is
    still added atomically are the bias gradients of convs in FRONT of an InstanceNorm (true value exactly 0), masked."""
    g = golden_model
    lr, hr = _fresh_audio(g, seed=321)
    a = _model(g, mask=False)
    runs = []
    for _ in range(3):
        _reset(a, g)
        ld = a.train_step(lr, hr)
        torch.cuda.synchronize()
        runs.append(({k: float(v) for k, v in ld.items()}, a.optimizer_G.flat_g.clone(), a.optimizer_D.flat_g.clone()))
    mG, mD = _signal_mask(a, a.optimizer_G), _signal_mask(a, a.optimizer_D)
    for r in runs[1:]:
        assert r[0] == runs[0][0], (r[0], runs[0][0])              # loss values: same bits
        assert torch.equal(r[1][mG], runs[0][1][mG])
        assert torch.equal(r[2][mD], runs[0][2][mD])


def test_paired_discriminator_batch_equals_two_passes(golden_model, monkeypatch):
    """Round 3: the training step runs D(real) and D(fake) as ONE batch of 2B (real half first), the generator-loss
    backward on the fake half only (_ops.backward_on_samples), the discriminator-loss backward on the whole batch.  Every
    forward value is identical to the two-pass schedule (InstanceNorm is per sample); weight gradients differ only by the
    fp32 summation order of one 2B-pixel reduction against two B-pixel ones."""
    g = golden_model
    lr, hr = _fresh_audio(g, seed=99)
    a, b = _model(g, mask=False), _model(g, mask=False)
    monkeypatch.setenv("P2PHD_DPAIR", "1")
    la = a.train_step(lr, hr)
    assert a._pair_batch == 2 * lr.shape[0]
    ga = {"G": a.optimizer_G.flat_g.clone(), "D": a.optimizer_D.flat_g.clone()}
    monkeypatch.setenv("P2PHD_DPAIR", "0")
    lb = b.train_step(lr, hr)
    assert b._pair_batch is None
    for k in la:
        va, vb = float(la[k]), float(lb[k])
        assert abs(va - vb) <= 2e-6 * max(abs(va), 1.0), (k, va, vb)
    assert _grad_diff(a, b, "optimizer_G") < 2e-5 and _grad_diff(a, b, "optimizer_D") < 2e-5
    # ... and replayed from the captured graphs (paired mode)
    monkeypatch.setenv("P2PHD_DPAIR", "1")
    for _ in range(3):
        a.train_step_graphed(lr, hr)
    _reset(a, g); _reset(b, g)
    a.train_step_graphed(lr, hr)
    monkeypatch.setenv("P2PHD_DPAIR", "0")
    b.train_step(lr, hr)
    assert _grad_diff(a, b, "optimizer_G") < 2e-5 and _grad_diff(a, b, "optimizer_D") < 2e-5


def test_graphed_step_equals_eager_step(golden_model):
    """train_step_graphed (two eager steps, capture, replay) against train_step FROM IDENTICAL STATE: the gradients of
    one replay of graphs A + B equal one eager backward (every reduction on the path has a fixed order), and so do the
    weights after one Adam update; then five steps for the device-side step counter / learning rate."""
    g = golden_model
    lr, hr = _fresh_audio(g)
    a, b = _model(g, mask=False), _model(g, mask=False)            # no mask noise: both paths see identical inputs
    for _ in range(3):
        b.train_step_graphed(lr, hr)                               # 2 eager steps, then capture + first replay
    assert b._graph_state['graphs'] is not None
    _reset(a, g); _reset(b, g)
    la = a.train_step(lr, hr)
    ga = {"G": a.optimizer_G.flat_g.clone(), "D": a.optimizer_D.flat_g.clone()}
    lb = b.train_step_graphed(lr, hr)
    gb = {"G": b.optimizer_G.flat_g.clone(), "D": b.optimizer_D.flat_g.clone()}
    for k in la:
        va, vb = float(la[k]), float(lb[k])
        assert abs(va - vb) <= 1e-5 * max(abs(va), 1.0), (k, va, vb)
    for t, opt_a in (("G", a.optimizer_G), ("D", a.optimizer_D)):
        m = _signal_mask(a, opt_a)
        err = float((ga[t][m] - gb[t][m]).norm() / ga[t][m].norm())
        # same kernels, same inputs, same weights.  Until round 3 this was bounded at 2e-2: the first pass of the two-pass
        # InstanceNorm backward (LDS + global float atomics), the column sums of the real bias gradients and the loss
        # accumulators added their partials in whatever order the workgroups finished, and a (Leaky)ReLU input within that
        # noise of zero then took the other branch in one of the runs.  Those reductions now fold in a fixed order
        # (common.h: fold_arrive_last), so replay and eager agree to the last bit on every element that carries signal.
        assert err <= 1e-6, (t, err)
    # one Adam step from zeroed moments is sign-like (|update| = lr): elements whose gradient is rounding noise may move
    # the other way, everything else must agree
    for t, oa, ob in (("G", a.optimizer_G, b.optimizer_G), ("D", a.optimizer_D, b.optimizer_D)):
        d = (oa.flat_p - ob.flat_p).abs()
        strong = (ga[t].abs() > 1e-3 * ga[t].abs().max()) & _signal_mask(a, oa)
        assert float(d[strong].max()) <= 1e-6, (t, float(d[strong].max()))
        assert float(d.max()) <= 2 * 2e-4 + 1e-7
    # five more steps: counters, LR change reaching the replayed graph, finiteness
    for i in range(5):
        if i == 4:
            for m in (a, b):
                m.update_learning_rate()
        la = a.train_step(lr, hr)
        lb = b.train_step_graphed(lr, hr)
        assert all(np.isfinite(float(v)) for v in lb.values())
    assert a.optimizer_G.steps_taken() == b.optimizer_G.steps_taken() == 6
    assert b.optimizer_G.step_count == 6 and b.optimizer_D.step_count == 6
    assert float(b.optimizer_G.lr_dev.item()) == pytest.approx(a.optimizer_G.param_groups[0]["lr"])


def test_eager_call_after_replays_sees_current_weights(tmp_path, golden_model):
    """Graph C runs Adam on the device; an eager inference() between replays must re-pack the conv weights (ConvSpec's
    packed cache is stamped with the weight epoch, which the replay path bumps): the output must equal a fresh model
    loaded from save()."""
    from pix2pixhdaudiosr_amd.models.models import create_model
    g = golden_model
    lr, hr = torch.from_numpy(g["lr"]).cuda(), torch.from_numpy(g["hr"]).cuda()
    m = _model(g, mask=False, checkpoints_dir=str(tmp_path), name="ck")
    for _ in range(6):
        m.train_step_graphed(lr, hr)
    assert m._graph_state['graphs'] is not None
    sr = m.inference(lr, None)[0]
    m.save("latest")
    fresh = create_model(make_opt(mask=False, checkpoints_dir=str(tmp_path), name="ck", continue_train=True))
    sr2 = fresh.inference(lr, None)[0]
    assert rel_err(sr.cpu().numpy(), sr2.cpu().numpy()) < 5e-5   # two fp32 runs: atomics order of the InstanceNorm sums
    # ... and the stale copy WOULD have been visibly different: six Adam steps move the output
    old = _model(g, mask=False).inference(lr, None)[0]
    assert rel_err(old.cpu().numpy(), sr2.cpu().numpy()) > 1e-4


def test_stale_forward_is_refused(golden_model):
    """The per-step arena that holds the InstanceNorm statistics is recycled by the next step's forward: a backward that
    still needs the old statistics must fail loudly instead of using zeroed sums."""
    from pix2pixhdaudiosr_amd import _lib
    g = golden_model
    lr, hr = torch.from_numpy(g["lr"]).cuda(), torch.from_numpy(g["hr"]).cuda()
    m = _model(g, mask=False)
    losses1, _ = m.forward(lr, None, hr, None)
    m.forward(lr, None, hr, None)                                  # a second forward recycles the arena
    ld = dict(zip(m.loss_names, losses1))
    with pytest.raises(_lib.P2PHDError):
        (ld['G_GAN'] + ld['G_GAN_Feat']).backward()


def test_amp_call_sequence_of_train_py(golden_model):
    """train.py:148-181 with --fp16, verbatim call sequence: forward under autocast(), GradScaler.scale(loss).backward(),
    scaler.step(optimizer) for G then D, scaler.update() -- against the same model stepped without a scaler.  --fp16 here
    means bf16 MFMA compute with fp32 master weights (exponent range of fp32: the 65536x loss scale is harmless), and
    FlatAdam's gradients are views of one flat buffer that the scaler un-scales in place."""
    from torch.cuda.amp import autocast, GradScaler
    g = golden_model
    lr, hr, noise = (torch.from_numpy(g[k]) for k in ("lr", "hr", "mask_noise"))
    a, b = _model(g, fp16=True), _model(g, fp16=True)
    # --- a: the reference's AMP sequence
    scaler = GradScaler()
    with autocast():
        losses, _ = a.forward(lr, None, hr, None, noise=noise)
    ld = dict(zip(a.loss_names, losses))
    loss_D = (ld['D_fake'] + ld['D_real']) * 0.5
    loss_G = ld['G_GAN'] + ld.get('G_GAN_Feat', 0)
    a.optimizer_G.zero_grad()
    scaler.scale(loss_G).backward()
    gG_scaled = a.optimizer_G.flat_g.clone()
    scaler.step(a.optimizer_G)
    a.optimizer_D.zero_grad()
    scaler.scale(loss_D).backward()
    scaler.step(a.optimizer_D)
    scaler.update()
    assert scaler.get_scale() == 65536.0                           # no inf / nan was found: both steps were applied
    assert a.optimizer_G.steps_taken() == 1 and a.optimizer_D.steps_taken() == 1
    # --- b: plain sequence
    losses_b, _ = b.forward(lr, None, hr, None, noise=noise)
    lb = dict(zip(b.loss_names, losses_b))
    b.optimizer_G.zero_grad(); (lb['G_GAN'] + lb['G_GAN_Feat']).backward()
    gG = b.optimizer_G.flat_g.clone()
    b.optimizer_G.step()
    b.optimizer_D.zero_grad(); ((lb['D_fake'] + lb['D_real']) * 0.5).backward(); b.optimizer_D.step()
    for k in ld:
        # two bf16 runs of one model differ in the last fp32 bits of the atomically summed InstanceNorm statistics, which
        # moves a few bf16 roundings downstream: ~1e-3 on a loss
        assert abs(float(ld[k]) - float(lb[k])) <= 5e-3 * max(1.0, abs(float(lb[k]))), k
    # the scaled backward carried 65536 x the gradient (a power of two: mantissas unchanged, no overflow in bf16 / fp32)
    m = _signal_mask(a, a.optimizer_G)
    assert torch.isfinite(gG_scaled).all()
    assert float((gG_scaled[m] / 65536.0 - gG[m]).norm() / gG[m].norm()) < 0.2      # two bf16 runs of a tiny net: ~5e-2
    # after un-scaling, Adam saw the same gradients: weights agree wherever the gradient is not rounding noise
    for oa, ob, gr in ((a.optimizer_G, b.optimizer_G, gG),):
        strong = (gr.abs() > 5e-2 * gr.abs().max()) & m
        assert float(((oa.flat_p - ob.flat_p).abs()[strong] > 2e-6).float().mean()) < 1e-2
    assert all(torch.isfinite(p).all() for p in a.parameters())


def test_staged_backward_equals_single_backward(golden_model):
    """grad_buckets = 4 cuts the generator backward into four stages (what the data-parallel step overlaps with its
    all-reduces): same kernels in the same order as the single backward, so gradients agree to the order of the float
    fixed-order reductions, on one GPU with no communication at all; graphed replay of the stages included."""
    g = golden_model
    lr, hr = _fresh_audio(g)
    one, four = _model(g, mask=False), _model(g, mask=False, grad_buckets=4)
    n, cut_after, offs = four._bucket_plan()
    assert n == 4 and len(cut_after) == 3 and offs == sorted(offs) and 0 < offs[0] and offs[-1] < four.optimizer_G._total, (cut_after, offs)
    assert one._bucket_plan()[1] == []
    la = one.train_step(lr, hr)
    lb = four.train_step(lr, hr)
    bk = four.optimizer_G.bucket_log
    assert [b for _, b in bk][0] == four.optimizer_G._total and bk[-1][0] == 0 and len(bk) == 4
    assert all(bk[i][0] == bk[i + 1][1] for i in range(3))
    for k in la:
        assert abs(float(la[k]) - float(lb[k])) <= 1e-5 * max(1.0, abs(float(la[k]))), k
    # same kernels on the same data, every reduction in a fixed order (see the graph test): no run-to-run slack
    assert _grad_diff(one, four, "optimizer_G") < 1e-6 and _grad_diff(one, four, "optimizer_D") < 1e-6
    # the staged capture: A0 .. A3 | B | C
    for _ in range(4):
        four.train_step_graphed(lr, hr)
    gA, ranges, _, _, _ = four._graph_state['graphs']
    assert len(gA) == 4 and ranges == bk
    _reset(one, g); _reset(four, g)
    one.train_step(lr, hr)
    four.train_step_graphed(lr, hr)
    assert _grad_diff(one, four, "optimizer_G") < 1e-6
    assert four.optimizer_G.bucket_log == bk


def test_local_enhancer_staged_backward_equals_single_backward():
    """Round 3: the LocalEnhancer (configs[2] / [3]) has a staged backward too.  Its backward graph forks behind the sum
    `model1_1(x) + model(pool(x))`; the first cut is the global chain's output, the head stage also runs the enhancer's
    head (LocalEnhancer.staged_head_anchors), and the buckets are contiguous ranges of the flat buffer covering it exactly
    once.  Same kernels in the same order as the single backward: identical gradients; also replayed from the graphs."""
    from pix2pixhdaudiosr_amd.models.models import create_model
    from pix2pixhdaudiosr_amd import _ops
    kw = dict(netG="local", ngf=8, n_downsample_global=2, n_blocks_global=2, n_local_enhancers=1, n_blocks_local=1, mask=False)
    one, four = create_model(make_opt(**kw)), create_model(make_opt(grad_buckets=4, **kw))
    four.netG.load_state_dict(one.netG.state_dict()); four.netD.load_state_dict(one.netD.state_dict())
    _ops.bump_weight_epoch()
    n, cut_after, offs = four._bucket_plan()
    assert n == 4 and len(cut_after) == 3 and offs == sorted(offs) and 0 < offs[0] and offs[-1] < four.optimizer_G._total
    assert cut_after[-1] == len(four.netG._steps('model')) - 1       # the last cut = the global chain's output
    gen = torch.Generator().manual_seed(5)
    T = 15 * 32
    lr, hr = (0.1 * torch.randn(2, T, generator=gen)).cuda(), (0.1 * torch.randn(2, T, generator=gen)).cuda()
    la = one.train_step(lr, hr)
    lb = four.train_step(lr, hr)
    bk = four.optimizer_G.bucket_log
    assert len(bk) == 4 and bk[0][1] == four.optimizer_G._total and bk[-1][0] == 0 and all(bk[i][0] == bk[i + 1][1] for i in range(3)), bk
    for k in la:
        assert abs(float(la[k]) - float(lb[k])) <= 1e-6 * max(1.0, abs(float(la[k]))), k
    assert _grad_diff(one, four, "optimizer_G") < 1e-6 and _grad_diff(one, four, "optimizer_D") < 1e-6
    # every generator tensor received its gradient (the enhancer's head is the branch a naive staging would skip)
    for (k, p), off in zip(four.netG.named_parameters(), four.optimizer_G._offs):
        if k.endswith(".weight"):
            assert float(four.optimizer_G.flat_g[off:off + p.numel()].abs().max()) > 0, k
    for _ in range(4):
        four.train_step_graphed(lr, hr)
    assert len(four._graph_state['graphs'][0]) == 4


def test_niter_fix_global_and_update_fixed_params():
    """--niter_fix_global (pix2pixHD_model.py:110-131): only the outermost local enhancer's parameters are optimised; after
    update_fixed_params (:521-528) every generator parameter is, with a fresh Adam; the step captured before the switch is
    re-captured on the new optimiser's buffers."""
    torch.manual_seed(5)
    from pix2pixhdaudiosr_amd.models.models import create_model
    m = create_model(make_opt(netG="local", n_local_enhancers=1, n_blocks_local=1, niter_fix_global=1, mask=False))
    prefix = "model1"
    n_local = sum(p.numel() for k, p in m.netG.named_parameters() if k.startswith(prefix))
    n_all = sum(p.numel() for p in m.netG.parameters())
    owned = lambda: sum(p.numel() for p in m.optimizer_G._params)   # (_total also counts alignment padding)
    assert 0 < n_local < n_all and owned() == n_local
    g = torch.Generator().manual_seed(9)
    T = 15 * 32
    lr, hr = 0.1 * torch.randn(2, T, generator=g), 0.1 * torch.randn(2, T, generator=g)
    before = {k: v.detach().clone() for k, v in m.netG.state_dict().items()}
    for _ in range(4):                                              # two eager steps, capture, one replay
        m.train_step_graphed(lr.cuda(), hr.cuda())
    torch.cuda.synchronize()
    after = m.netG.state_dict()
    moved = {k for k in before if not torch.equal(before[k], after[k])}
    assert moved and all(k.startswith(prefix) for k in moved), sorted(moved)[:5]
    m.update_fixed_params()
    assert owned() == n_all and m._graph_state is None
    before = {k: v.detach().clone() for k, v in m.netG.state_dict().items()}
    for _ in range(4):
        m.train_step_graphed(lr.cuda(), hr.cuda())
    torch.cuda.synchronize()
    after = m.netG.state_dict()
    weights = [k for k in before if k.endswith(".weight")]
    assert all(not torch.equal(before[k], after[k]) for k in weights)
    assert all(torch.isfinite(v).all() for v in after.values())
