"""Whole networks at BASELINE geometry on the HIP path (fp32 parity mode, through the C ABI) against the CPU oracle
(oracle/model.py, pinned to the reference by tests/test_oracle_*.py):
  * configs[1]: GlobalGenerator ngf 48 / 4 down-samplings / 9 blocks + 2-scale discriminator on a 512x256 spectrogram,
    one sample: the four losses, the generated spectrogram, the gradients of BOTH backward passes (train.py:155-184);
  * configs[2] (opt.txt reading of G3L2_48ngf): LocalEnhancer forward at 512x256;
  * configs[4]: the 3-scale discriminator against the reference's own outputs (tests/golden/networks_d3.npz);
  * round 3: configs[0] at its real geometry (ngf 32 global, 512x256, B = 2: one whole step incl. codec and Adam against
    oracle.model.full_step), configs[2]'s LocalEnhancer BACKWARD at 512x256, configs[4]'s full model (n_fft 2048, ngf 64
    local generator = 730 713 346 parameters, 3-scale D, 1024x512, bf16 + fp8) through one graphed step."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err, assert_grad_close, noise_bias_keys, GOLDEN

pytestmark = pytest.mark.gpu


def _opt(**kw):
    from test_gpu_model import make_opt
    o = dict(n_fft=1024, hop_length=512, win_length=1024, ngf=48, netG="global", n_downsample_global=4, n_blocks_global=9,
             n_local_enhancers=0, n_blocks_local=3, ndf=64, n_layers_D=3, num_D=2, mask=False)
    o.update(kw)
    return make_opt(**o)


def _load_from(model_net, params):
    sd = {k: params[k] for k in model_net.state_dict().keys()}
    model_net.load_state_dict(sd)


def test_configs1_network_losses_and_both_backward_passes():
    from oracle import model as OM
    from oracle import mdct4 as M4
    from pix2pixhdaudiosr_amd.models.models import create_model
    from pix2pixhdaudiosr_amd import _ops
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    oo = OM.default_opt(ngf=48, netG="global", n_downsample_global=4, n_blocks_global=9, mask=False)
    pG = OM.N.init_params(OM.netG_spec(oo), seed=1)
    pD = OM.N.init_params(OM.netD_spec(oo), seed=2)
    assert sum(v.numel() for v in pG.values()) == 102_627_170              # configs[1]'s generator (bench.py prints the same count)
    assert sum(v.numel() for v in pD.values()) == 5_531_522                # train_script.sh KAT
    hr, lr, _ = OM.synthetic_batch(1, oo, seed=5)
    w = M4.kbdwin(oo.win_length)
    hr_s, _, _ = OM.to_spectro(hr, oo, w, mask=False)
    lr_s, _, _ = OM.to_spectro(lr, oo, w, mask=False)
    assert tuple(lr_s.shape) == (1, 2, 512, 256)
    L, gG, gD = OM.step_grads(pG, pD, lr_s, hr_s, oo)
    with torch.no_grad():                                          # the exact result (fp64) of the same generator
        sr64 = OM.netG_forward({k: v.double() for k, v in pG.items()}, lr_s.double(), oo)

    m = create_model(_opt())
    _load_from(m.netG, pG); _load_from(m.netD, pD)
    _ops.bump_weight_epoch()
    # both sides get the SAME encoded spectrograms: the codec (MDCT + dB + global min/max, checked on its own in
    # test_gpu_model.py at 1e-4) would otherwise put its own 1e-4 in front of the networks under test here
    enc = (lr_s.cuda(), None, hr_s.cuda(), None, None, None, None, None)
    m.encode_input = lambda *a, **k: enc
    losses, sr = m.forward(lr, None, hr, None, infer=True)
    got = dict(zip(m.loss_names, losses))
    for k in ("G_GAN", "G_GAN_Feat", "D_real", "D_fake"):
        assert abs(float(got[k]) - L[k]) <= 5e-4 * max(1.0, abs(L[k])), (k, float(got[k]), L[k])
    # north_star: activations within 1e-4 rel of the reference CPU path, no alternative (measured 1.1e-5 since the
    # InstanceNorm statistics are per-wave (sum, M2) partials merged with Chan's update; the fp64 distances are printed
    # for context only).
    e_sr = rel_err(sr.detach().cpu().numpy(), L["sr"].numpy())
    e_hip64 = rel_err(sr.detach().cpu().numpy(), sr64.numpy())
    e_cpu64 = rel_err(L["sr"].numpy(), sr64.numpy())
    print(f"full-size generator output: HIP vs CPU fp32 {e_sr:.2e}; vs fp64: HIP {e_hip64:.2e}, CPU fp32 {e_cpu64:.2e}")
    assert e_sr < 1e-4, (e_sr, e_hip64, e_cpu64)
    m.optimizer_G.zero_grad(); (got["G_GAN"] + got["G_GAN_Feat"]).backward(retain_graph=True)
    gG_hip = {k: p.grad.detach().cpu().clone() for k, p in m.netG.named_parameters()}
    m.optimizer_D.zero_grad(); ((got["D_fake"] + got["D_real"]) * 0.5).backward()
    gD_hip = {k: p.grad.detach().cpu().clone() for k, p in m.netD.named_parameters()}
    # Gradients.  Round 2 bounded these at 5e-2 and quoted "measured worst 6e-4"; the round-2 review asked for 2e-3.  Neither
    # figure survives measurement: on this input the REFERENCE'S OWN fp32 CPU path is 7e-3 .. 1e-2 away from the exact
    # (fp64) gradient on every generator layer (profiles/r03_reference_fp32_vs_fp64_gradients.txt: the L1 feature-matching
    # term differentiates sign(fake - real) and every InstanceNorm + ReLU has elements within fp32 rounding of its kink;
    # one flipped branch moves the gradient by a full term), so two correct fp32 implementations cannot agree better than
    # that with each other.  The anchor is therefore the exact gradient: per tensor, the HIP path must be no farther from
    # the fp64 result than 2x the reference's fp32 path is (+1e-4), and within 3e-2 of that path in any case.
    L64, gG64, gD64 = OM.step_grads({k: v.double() for k, v in pG.items()}, {k: v.double() for k, v in pD.items()},
                                    lr_s.double(), hr_s.double(), oo)
    nbG, nbD = noise_bias_keys(list(gG)), noise_bias_keys(list(gD))
    report = []
    for tag, ref, ref64, hip, nb in (("G", gG, gG64, gG_hip, nbG), ("D", gD, gD64, gD_hip, nbD)):
        for k, v in ref.items():
            if k in nb:
                continue
            e_hip = rel_err(hip[k].numpy(), ref64[k].numpy())
            e_cpu = rel_err(v.numpy(), ref64[k].numpy())
            e_pair = rel_err(hip[k].numpy(), v.numpy())
            report.append((f"{tag}:{k}", e_hip, e_cpu, e_pair))
    for name, e_hip, e_cpu, e_pair in report:
        print(f"  grad {name:40s} vs fp64: HIP {e_hip:.2e}  CPU fp32 {e_cpu:.2e} | HIP vs CPU fp32 {e_pair:.2e}")
    globals()["_LAST_REPORT"] = report
    for name, e_hip, e_cpu, e_pair in report:
        assert e_hip <= 2.0 * e_cpu + 1e-4, (name, e_hip, e_cpu)
        assert e_pair <= 3e-2, (name, e_pair)
    # biases in front of an InstanceNorm: exact gradient 0, both sides hold rounding noise -- absolute bound only
    for tag, ref, hip, nb in (("G", gG, gG_hip, nbG), ("D", gD, gD_hip, nbD)):
        for k in nb:
            assert_grad_close(f"{tag}:{k}", hip[k].numpy(), ref[k].numpy(), rtol=0.0, bias_floor=2e-2, noise_biases=nb)


def test_configs2_local_enhancer_forward_full_size():
    """GEN_VCTK_G3L2_48ngf as the reference's opt.txt reads it: LocalEnhancer ngf 48, 4 global down-samplings, 3 global
    blocks, 1 local enhancer, 2 local blocks (156 050 690 parameters), 512x256 input."""
    from oracle import networks as N
    from pix2pixhdaudiosr_amd.models import networks as PN
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    spec = N.local_enhancer_spec(2, 2, 48, 4, 3, 1, 2)
    assert N.param_count(spec) == 156_050_690
    p = N.init_params(spec, seed=3)
    x = torch.rand(1, 2, 512, 256, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref = N.local_enhancer_forward(p, x, 4, 3, 1, 2)
    net = PN.define_G(2, 2, 48, "local", 4, 3, 1, 2, "instance", [], dtype=torch.float32, verbose=False)
    net.load_state_dict({k: p[k] for k in net.state_dict().keys()})
    net = net.cuda()
    from pix2pixhdaudiosr_amd import _ops
    _ops.bump_weight_epoch()
    with torch.no_grad():
        y = net(x.cuda())
    assert tuple(y.shape) == tuple(ref.shape) == (1, 2, 512, 256)
    assert rel_err(y.cpu().numpy(), ref.numpy()) < 1e-4


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 1e-4, 3e-4), (torch.bfloat16, 5e-2, None)])
def test_three_scale_discriminator_against_reference(dtype, tol, gtol):
    """num_D = 3 (BASELINE configs[4]; reference models/networks.py:292-331): feature list structure, odd k4 p2 sizes,
    every feature, the input gradient and every parameter gradient against the reference's own run."""
    from pix2pixhdaudiosr_amd.models import networks as PN
    g = np.load(os.path.join(GOLDEN, "networks_d3.npz"))
    tag = "D3"
    net = PN.define_D(4, 8, 3, "instance", False, 3, True, [], dtype=dtype, verbose=False)
    keys = [str(k) for k in g[f"{tag}_keys"]]
    assert list(net.state_dict().keys()) == keys
    net.load_state_dict({k: torch.from_numpy(g[f"{tag}_p_{k}"]) for k in keys})
    net = net.cuda()
    x = torch.from_numpy(g[f"{tag}_x"]).cuda().requires_grad_(True)
    res = net(x)
    assert [len(s) for s in res] == list(g[f"{tag}_nfeat"]) == [5, 5, 5]
    flat = [f for s in res for f in s]
    tot = 0
    for i, f in enumerate(flat):
        assert tuple(f.shape) == g[f"{tag}_f{i}"].shape, i
        assert rel_err(f.detach().float().cpu().numpy(), g[f"{tag}_f{i}"]) < tol, i
        tot = tot + (f.float() * torch.from_numpy(g[f"{tag}_c{i}"]).cuda()).sum()
    params = dict(net.named_parameters())
    grads = torch.autograd.grad(tot, [x] + list(params.values()))
    assert all(torch.isfinite(gr).all() for gr in grads)
    if gtol is not None:
        assert rel_err(grads[0].cpu().numpy(), g[f"{tag}_gx"]) < gtol
        nb = noise_bias_keys(keys)
        assert len(nb) == 3 * 3
        for k, gr in zip(params.keys(), grads[1:]):
            assert_grad_close(f"{tag}:{k}", gr.cpu().numpy(), g[f"{tag}_g_{k}"], rtol=gtol, noise_biases=nb)


@pytest.mark.parametrize("half", ["bf16", "fp16"])
def test_configs1_bf16_step_tracks_the_fp32_step_at_full_size(half):
    """(`fp16`, round 4: the reference's actual AMP storage type -- opt.fp16_storage, the fp16 build of the library, loss scaled
    by 2^16 as torch.cuda.amp.GradScaler starts -- must track the fp32 step far closer: 11 significand bits against 8.)
    The benchmarked mode at the benchmarked geometry: configs[1]'s networks (real widths, 512x256, two samples) in bf16
    against the SAME step in fp32 on the HIP path (itself checked against the oracle above): losses, generated spectrogram
    and every weight gradient of both backward passes (what the direction-only checks on the 2-8-channel golden nets,
    test_gpu_networks.py, cannot say about the real widths)."""
    from oracle import model as OM
    from pix2pixhdaudiosr_amd.models.models import create_model
    from pix2pixhdaudiosr_amd import _ops
    oo = OM.default_opt(ngf=48, netG="global", n_downsample_global=4, n_blocks_global=9, mask=False)
    pG = OM.N.init_params(OM.netG_spec(oo), seed=1)
    pD = OM.N.init_params(OM.netD_spec(oo), seed=2)
    hr, lr, _ = OM.synthetic_batch(2, oo, seed=6)
    res = {}
    for name, kw in (("f32", dict(fp16=False)), (half, dict(fp16=True, fp16_storage=half == "fp16"))):
        m = create_model(_opt(**kw))
        assert m.compute_dtype == {"f32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[name]
        _load_from(m.netG, pG); _load_from(m.netD, pD)
        _ops.bump_weight_epoch()
        losses, sr = m.forward(lr, None, hr, None, infer=True)
        got = dict(zip(m.loss_names, losses))
        S = 65536.0 if name == "fp16" else 1.0                      # GradScaler's initial scale (train.py:62-67)
        m.optimizer_G.zero_grad(); ((got["G_GAN"] + got["G_GAN_Feat"]) * S).backward(retain_graph=True)
        gG = {k: p.grad.detach().float().cpu().clone() / S for k, p in m.netG.named_parameters()}
        m.optimizer_D.zero_grad(); ((got["D_fake"] + got["D_real"]) * (0.5 * S)).backward()
        gD = {k: p.grad.detach().float().cpu().clone() / S for k, p in m.netD.named_parameters()}
        assert all(torch.isfinite(v).all() for v in list(gG.values()) + list(gD.values())), name
        res[name] = ({k: float(v) for k, v in got.items()}, sr.detach().float().cpu(), gG, gD)
        del m
        torch.cuda.empty_cache()
    (l32, sr32, gG32, gD32), (l16, sr16, gG16, gD16) = res["f32"], res[half]
    for k in l32:
        assert abs(l16[k] - l32[k]) <= 2e-2 * max(1.0, abs(l32[k])), (k, l16[k], l32[k])
    e_sr = rel_err(sr16.numpy(), sr32.numpy())
    worst = []
    for tag, a, b in (("G", gG16, gG32), ("D", gD16, gD32)):
        nb = noise_bias_keys(list(b))
        for k, v in b.items():
            if k.endswith(".weight"):
                e = rel_err(a[k].numpy(), v.numpy())
                c = float((a[k].double().flatten() @ v.double().flatten()) / max(float(a[k].double().norm() * v.double().norm()), 1e-300))
                worst.append((e, c, f"{tag}:{k}"))
    if os.environ.get("P2PHD_VERBOSE_TESTS"):
        for e, c, name in worst:
            print(f"   {name:36s} rel err {e:.3f} cosine {c:.4f}")
    worst.sort(reverse=True)
    print(f"{half} vs fp32 at full size: sr rel err {e_sr:.2e}; worst weight-gradient rel err {worst[0][0]:.2e} ({worst[0][2]}), "
          f"median {worst[len(worst) // 2][0]:.2e}, min cosine {min(c for _, c, _ in worst):.4f}")
    if half == "fp16":
        # the round-3 review's bar for the reference's AMP type: cosine >= 0.97 at every generator layer, spectrogram <= 1.5e-2
        assert e_sr < 1.5e-2, e_sr
        assert min(c for _, c, _ in worst) >= 0.97, min(c for _, c, _ in worst)
        assert max(e for e, _, _ in worst) < 2.5e-1
        return
    # Measured: spectrogram 4.9e-2 (1.2e-1 before the generator input was centred, networks._centered_input);
    # discriminator gradients 1-8 %; generator gradients 9 % at the output layer growing to 50 % (cosine 0.87) at the
    # input layer -- not rounding of the gradient itself but ReLU masks: 3-5 % activation noise flips the branch of the
    # ~2 % of elements that sit that close to zero, in every one of 28 layers, and each flip moves a gradient element by a
    # full term.  bf16 has 8 mantissa bits; the reference's fp16 autocast has 11.
    assert e_sr < 8e-2
    d_err = [e for e, _, n in worst if n.startswith("D:")]
    assert max(d_err) < 1.5e-1, max(d_err)
    assert min(c for _, c, _ in worst) > 0.8
    assert [e for e, _, n in worst if n == "G:model.38.weight"][0] < 1.5e-1


def test_configs0_one_step_at_real_geometry_against_the_oracle():
    """BASELINE configs[0] ("CPU reference: n_fft 1024 (512x256), ngf 32, 1 global G, 2-scale D, batch 2"): the WHOLE step
    -- MDCT4 + dB codec + mode2 mask with injected noise, G, D x3, the four losses, both backward passes, both Adam
    updates (train.py:148-184) -- against oracle.model.full_step on the same audio and weights, fp32."""
    from oracle import model as OM
    from oracle import mdct4 as M4
    from pix2pixhdaudiosr_amd.models.models import create_model
    from pix2pixhdaudiosr_amd import _ops
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    oo = OM.default_opt(ngf=32, netG="global", n_downsample_global=4, n_blocks_global=9, mask=True)
    pG = OM.N.init_params(OM.netG_spec(oo), seed=11)
    pD = OM.N.init_params(OM.netD_spec(oo), seed=12)
    assert sum(v.numel() for v in pG.values()) == 45_617_730
    hr, lr, noise = OM.synthetic_batch(2, oo, seed=7)
    w = M4.kbdwin(oo.win_length)
    L, pG1, pD1 = OM.full_step(hr, lr, noise, dict(pG), dict(pD), oo, w, {}, {})

    m = create_model(_opt(ngf=32, mask=True))
    _load_from(m.netG, pG); _load_from(m.netD, pD)
    _ops.bump_weight_epoch()
    ld = m.train_step(lr, hr, noise=noise)
    for k in ("G_GAN", "G_GAN_Feat", "D_real", "D_fake"):
        assert abs(float(ld[k]) - L[k]) <= 5e-4 * max(1.0, abs(L[k])), (k, float(ld[k]), L[k])
    # Gradients of both backward passes, BEFORE the update (they stay in the optimisers' flat buffers until the next step's
    # zero_grad): magnitudes, not only signs -- a split-K or paired-batch scaling bug passes the sign test below.  This test
    # runs the HIP path's OWN codec in front of the networks (1e-4 on the spectrogram, tests/test_gpu_model.py), and an input
    # that differs by 1e-4 moves the (Leaky)ReLU branch of ~1e-4 of the elements of every layer; a share f of flipped
    # branches moves a gradient by ~sqrt(f) of its norm (profiles/r04_fp32_vs_fp64_gradient_parts.txt shows the same
    # mechanism at fp32 rounding: f ~ 1e-5 .. 1e-4 -> 5e-3 .. 1e-2).  So the HIP gradients cannot sit closer than ~1e-2 to
    # gradients computed from the oracle's spectrogram; the configs[1] test, which injects the oracle's spectrograms,
    # holds the tight anchor.  Here: within 3e-2 of the reference-style fp32 result per tensor (a scaling error is 0.5 .. 1).
    with torch.no_grad():
        hr_s, _, _ = OM.to_spectro(hr, oo, w, mask=False)
        lr_s, _, _ = OM.to_spectro(lr, oo, w, mask=oo.mask, noise=noise)
    _, gG, gD = OM.step_grads(pG, pD, lr_s, hr_s, oo)
    worst = (0.0, "")
    for tag, net, ref in (("G", m.netG, gG), ("D", m.netD, gD)):
        nb = noise_bias_keys(list(ref))
        for k, p in net.named_parameters():
            if k in nb:
                continue
            e_pair = rel_err(p.grad.detach().cpu().numpy(), ref[k].numpy())
            worst = max(worst, (e_pair, f"{tag}:{k}"))
            if os.environ.get("P2PHD_VERBOSE_TESTS"):
                print(f"  cfg0 grad {tag}:{k:36s} HIP vs CPU fp32 {e_pair:.2e}")
            assert e_pair <= 3e-2, (tag, k, e_pair)
    print(f"configs[0] gradients before the update: worst HIP vs CPU fp32 {worst[0]:.2e} ({worst[1]})")
    # weights after Adam: the first step moves every element by +-lr (sign of its gradient), so an element agrees with the
    # oracle exactly or is off by 2 lr; elements whose gradient is rounding noise (biases in front of an InstanceNorm, and
    # the near-zero tail of every tensor) may take either sign.  Bound the share of disagreeing weight elements.
    for net, ref in ((m.netG, pG1), (m.netD, pD1)):
        nb = noise_bias_keys(list(ref))
        for k, p in net.state_dict().items():
            if k in nb:
                continue
            bad = float(((p.detach().cpu() - ref[k]).abs() > 1e-6).float().mean())
            assert bad < 2e-2, (k, bad)


def test_configs2_local_enhancer_backward_full_size():
    """configs[2] (GEN_VCTK_G3L2_48ngf as opt.txt reads it): forward AND backward of the LocalEnhancer at 512x256, one
    sample, fp32, against the oracle: input gradient and every parameter gradient for a fixed linear functional of the
    output (models/networks.py:129-183)."""
    from oracle import networks as N
    from pix2pixhdaudiosr_amd.models import networks as PN
    from pix2pixhdaudiosr_amd import _ops
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    spec = N.local_enhancer_spec(2, 2, 48, 4, 3, 1, 2)
    p = {k: v.clone().requires_grad_(True) for k, v in N.init_params(spec, seed=3).items()}
    gen = torch.Generator().manual_seed(10)
    x = torch.rand(1, 2, 512, 256, generator=gen).requires_grad_(True)
    c = torch.randn(1, 2, 512, 256, generator=gen)
    ref = N.local_enhancer_forward(p, x, 4, 3, 1, 2)
    gref = torch.autograd.grad((ref * c).sum(), [x] + list(p.values()))
    net = PN.define_G(2, 2, 48, "local", 4, 3, 1, 2, "instance", [], dtype=torch.float32, verbose=False)
    net.load_state_dict({k: p[k].detach() for k in net.state_dict().keys()})
    net = net.cuda()
    _ops.bump_weight_epoch()
    xd = x.detach().cuda().requires_grad_(True)
    y = net(xd)
    assert rel_err(y.detach().cpu().numpy(), ref.detach().numpy()) < 1e-4
    params = dict(net.named_parameters())
    ghip = torch.autograd.grad((y * c.cuda()).sum(), [xd] + list(params.values()))
    # anchor = the exact (fp64) gradient of the same functional, as in the configs[1] test above: a ReLU input within fp32
    # rounding of 0 takes either branch, so the reference's fp32 path is itself ~5e-3 from the exact input gradient
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in p.items()}
    x64 = x.detach().double().requires_grad_(True)
    g64 = torch.autograd.grad((N.local_enhancer_forward(p64, x64, 4, 3, 1, 2) * c.double()).sum(), [x64] + list(p64.values()))
    nb = noise_bias_keys(list(params))
    names = ["input"] + list(params.keys())
    for name, gh, gr, ge in zip(names, ghip, gref, g64):
        if name in nb:
            # bias in front of an InstanceNorm: exact gradient 0; both sides hold the rounding noise of a sum over the plane
            # (here up to 131 072 pixels x 1 .. 768 channels: ~0.2 in norm), which says nothing about either side
            assert np.isfinite(gh.cpu().numpy()).all() and float(gh.abs().max()) < 1.0, name
            continue
        e_hip, e_cpu = rel_err(gh.cpu().numpy(), ge.numpy()), rel_err(gr.numpy(), ge.numpy())
        print(f"  grad L:{name:36s} vs fp64: HIP {e_hip:.2e}  CPU fp32 {e_cpu:.2e}")
        assert e_hip <= 2.0 * e_cpu + 1e-4, (name, e_hip, e_cpu)
        assert rel_err(gh.cpu().numpy(), gr.numpy()) <= 3e-2, name


def test_configs4_full_model_one_graphed_step_bf16_fp8():
    """configs[4] per GPU: n_fft 2048 (1024x512 spectrograms), ngf 64 LocalEnhancer (the reference's default generator,
    730 713 346 parameters -- train_script.sh KAT), 3-scale discriminator, bf16 + fp8 forward of the wide convs, batch 2,
    through the captured-graph step.  No CPU oracle at this size (minutes per step): size-independent properties."""
    from pix2pixhdaudiosr_amd.models.models import create_model
    o = _opt(n_fft=2048, hop_length=1024, win_length=2048, ngf=64, netG="local", n_downsample_global=4, n_blocks_global=9,
             n_local_enhancers=1, n_blocks_local=3, num_D=3, fp16=True, fp8=True, mask=True)
    m = create_model(o)
    assert sum(p.numel() for p in m.netG.parameters()) == 730_713_346
    assert m.fp8_layers > 0
    T = 511 * 1024
    gen = torch.Generator().manual_seed(3)
    hr = (0.1 * torch.randn(2, T, generator=gen)).cuda()
    lr = (0.1 * torch.randn(2, T, generator=gen)).cuda()
    # feature structure of the 3-scale D at this geometry (15 features: 3 scales x 5 stages)
    with torch.no_grad():
        ls, _, hs, _, _, _, _, _ = m.encode_input(lr, None, hr, None)
        assert tuple(ls.shape) == (2, 2, 1024, 512)
        feats = m.netD(torch.cat((ls, hs), dim=1))
    assert [len(s) for s in feats] == [5, 5, 5]
    shapes = [tuple(f.shape) for s in feats for f in s]
    assert shapes[0] == (2, 64, 513, 257) and shapes[4] == (2, 1, 131, 67) and shapes[5] == (2, 64, 257, 129) and shapes[14] == (2, 1, 35, 19), shapes
    w0 = m.optimizer_G.flat_p.clone()
    for _ in range(4):                                             # 2 eager steps, capture, one more replay
        ld = m.train_step_graphed(lr, hr)
    torch.cuda.synchronize()
    assert m._graph_state["graphs"] is not None
    for k in ("G_GAN", "G_GAN_Feat", "D_real", "D_fake"):
        assert np.isfinite(float(ld[k])) and float(ld[k]) > 0, (k, float(ld[k]))
    assert torch.isfinite(m.optimizer_G.flat_p).all() and torch.isfinite(m.optimizer_D.flat_p).all()
    # every parameter tensor received a gradient in the last replay (nonzero in every layer, G and D)
    for opt_ in (m.optimizer_G, m.optimizer_D):
        for p_, off in zip(opt_._params, opt_._offs):
            seg = opt_.flat_g[off:off + p_.numel()]
            assert torch.isfinite(seg).all() and float(seg.abs().max()) > 0, (tuple(p_.shape), off)
    assert float((m.optimizer_G.flat_p - w0).abs().max()) > 0


def test_bf16_training_tracks_fp32_over_200_steps():
    """Round-2 review item 4: is the benchmarked bf16 step a usable training step?  The same reduced-width configs[1] model
    (ngf = ndf = 16, 512x256, 4 down-samplings, 9 blocks, 2-scale D), identical initial weights, four fixed batches cycled in
    the same order, 200 graph-replayed optimisation steps in fp32 and in bf16: the four losses must stay inside a band
    around the fp32 run at every checkpoint (the reference's AMP recipe, train.py:62-67,148-181, is the analogue).
    Measured (tools/soak_pair.py, profiles/r03_soak_pair.log): largest deviation 14 % of the loss value (D_fake at step
    125; GAN training is chaotic, the trajectories do not stay bit-close), mean 2.5 %."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_opt
    from pix2pixhdaudiosr_amd.models.models import create_model

    def run(fp16):
        opt = make_opt(2, dtype_bf16=fp16)
        opt.ngf, opt.ndf, opt.mask = 16, 16, False
        torch.manual_seed(1234)
        m = create_model(opt)
        T = 255 * opt.hop_length
        g = torch.Generator(device="cuda").manual_seed(7)
        data = [(0.1 * torch.randn(2, T, device="cuda", generator=g), 0.1 * torch.randn(2, T, device="cuda", generator=g)) for _ in range(4)]
        traj = []
        for i in range(200):
            ld = m.train_step_graphed(*data[i % 4])
            if (i + 1) % 25 == 0:
                traj.append({k: float(v) for k, v in ld.items()})
        torch.cuda.synchronize()
        assert all(torch.isfinite(p).all() for p in m.parameters())
        del m
        torch.cuda.empty_cache()
        return traj

    t32, t16 = run(False), run(True)
    devs = []
    for i, (a, b) in enumerate(zip(t32, t16)):
        for k in a:
            dev = abs(b[k] - a[k]) / max(abs(a[k]), 0.25)
            devs.append(dev)
            assert dev <= 0.3, (25 * (i + 1), k, a[k], b[k])
    print(f"bf16 vs fp32 over 200 steps: max loss deviation {max(devs):.3f}, mean {sum(devs) / len(devs):.3f}")
    assert sum(devs) / len(devs) <= 0.08
    # both runs learn: the feature-matching loss falls by the same factor
    assert t32[-1]["G_GAN_Feat"] < 0.9 * t32[0]["G_GAN_Feat"] and t16[-1]["G_GAN_Feat"] < 0.9 * t16[0]["G_GAN_Feat"]


def test_configs3_per_rank_step_local_enhancer_bf16_batch32():
    """BASELINE configs[3] is configs[2]'s generator (GEN_VCTK_G3L2_48ngf as opt.txt reads it: LocalEnhancer ngf 48, 4 global
    down-samplings, 3 global blocks, 1 local enhancer, 2 local blocks) in bf16 at per-GPU batch 32 on 8 GPUs.  What one rank
    runs -- the graphed bf16 step of that model at B = 32, 512x256 -- runs here on one GPU (the exchange itself:
    tests/test_gpu_dp.py, test_dp_gloo.py; the LocalEnhancer's 4-stage backward: test_gpu_model.py), with the staged
    generator backward the data-parallel step uses (grad_buckets = 4)."""
    from pix2pixhdaudiosr_amd.models.models import create_model
    o = _opt(netG="local", n_downsample_global=4, n_blocks_global=3, n_local_enhancers=1, n_blocks_local=2, fp16=True, mask=True, grad_buckets=4)
    m = create_model(o)
    assert sum(p.numel() for p in m.netG.parameters()) == 156_050_690
    assert m._bucket_plan()[0] == 4 and len(m._bucket_plan()[1]) == 3
    T = 255 * 512
    gen = torch.Generator().manual_seed(4)
    hr = (0.1 * torch.randn(32, T, generator=gen)).cuda()
    lr = (0.1 * torch.randn(32, T, generator=gen)).cuda()
    w0 = m.optimizer_G.flat_p.clone()
    for _ in range(4):                                             # 2 eager steps, capture, one more replay
        ld = m.train_step_graphed(lr, hr)
    torch.cuda.synchronize()
    assert m._graph_state["graphs"] is not None
    for k in ("G_GAN", "G_GAN_Feat", "D_real", "D_fake"):
        assert np.isfinite(float(ld[k])) and float(ld[k]) > 0, (k, float(ld[k]))
    for opt_ in (m.optimizer_G, m.optimizer_D):
        assert torch.isfinite(opt_.flat_p).all()
        for p_, off in zip(opt_._params, opt_._offs):
            seg = opt_.flat_g[off:off + p_.numel()]
            assert torch.isfinite(seg).all() and float(seg.abs().max()) > 0, (tuple(p_.shape), off)
    assert float((m.optimizer_G.flat_p - w0).abs().max()) > 0


def test_baseline_wording_g3l2_forward_full_size():
    """BASELINE.json words configs[2] / [3] as "3 global downsamplings, 2 local enhancers, ngf=48" -- not what the reference's
    own opt.txt says (4 / 1, tested above), so both readings are covered.  This one: define_G(2, 2, 48, 'local',
    n_downsample_global=3, n_blocks_global=9, n_local_enhancers=2, n_blocks_local=3) = 413 049 986 parameters
    (models/networks.py:129-181: two enhancer stages around a global generator of ngf 192 at a quarter of the resolution);
    forward at 512x256, one sample, fp32, against the oracle at north_star's 1e-4."""
    from oracle import networks as N
    from pix2pixhdaudiosr_amd.models import networks as PN
    from pix2pixhdaudiosr_amd import _ops
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    spec = N.local_enhancer_spec(2, 2, 48, 3, 9, 2, 3)
    assert N.param_count(spec) == 413_049_986
    p = N.init_params(spec, seed=21)
    x = torch.rand(1, 2, 512, 256, generator=torch.Generator().manual_seed(22))
    with torch.no_grad():
        ref = N.local_enhancer_forward(p, x, 3, 9, 2, 3)
    net = PN.define_G(2, 2, 48, "local", 3, 9, 2, 3, "instance", [], dtype=torch.float32, verbose=False)
    assert sum(q.numel() for q in net.parameters()) == 413_049_986
    net.load_state_dict({k: p[k] for k in net.state_dict().keys()})
    net = net.cuda()
    _ops.bump_weight_epoch()
    with torch.no_grad():
        y = net(x.cuda())
    assert tuple(y.shape) == tuple(ref.shape) == (1, 2, 512, 256)
    e = rel_err(y.cpu().numpy(), ref.numpy())
    print(f"G3L2 (BASELINE wording, 413 M parameters) forward at 512x256: HIP vs CPU fp32 {e:.2e}")
    assert e < 1e-4


def test_baseline_wording_g3l2_graphed_bf16_step_batch8():
    """The same generator (nd3, nle2, nb9, nbl3) through one captured bf16 training step at batch 8: the size-independent
    properties the configs[3] test checks.  (Two enhancer stages put two forks into the backward graph; the staged backward
    of the data-parallel exchange covers one -- LocalEnhancer.bucket_plan -- so this generator's gradients travel as one
    bucket, exchanged beside the discriminator backward.)"""
    from pix2pixhdaudiosr_amd.models.models import create_model
    o = _opt(netG="local", n_downsample_global=3, n_blocks_global=9, n_local_enhancers=2, n_blocks_local=3, fp16=True, mask=True,
             grad_buckets=4)
    m = create_model(o)
    assert sum(p.numel() for p in m.netG.parameters()) == 413_049_986
    assert m._bucket_plan()[1] == []                               # one stage
    T = 255 * 512
    gen = torch.Generator().manual_seed(5)
    hr = (0.1 * torch.randn(8, T, generator=gen)).cuda()
    lr = (0.1 * torch.randn(8, T, generator=gen)).cuda()
    w0 = m.optimizer_G.flat_p.clone()
    for _ in range(4):                                             # 2 eager steps, capture, one more replay
        ld = m.train_step_graphed(lr, hr)
    torch.cuda.synchronize()
    assert m._graph_state["graphs"] is not None
    for k in ("G_GAN", "G_GAN_Feat", "D_real", "D_fake"):
        assert np.isfinite(float(ld[k])) and float(ld[k]) > 0, (k, float(ld[k]))
    for opt_ in (m.optimizer_G, m.optimizer_D):
        assert torch.isfinite(opt_.flat_p).all()
        for p_, off in zip(opt_._params, opt_._offs):
            seg = opt_.flat_g[off:off + p_.numel()]
            assert torch.isfinite(seg).all() and float(seg.abs().max()) > 0, (tuple(p_.shape), off)
    assert float((m.optimizer_G.flat_p - w0).abs().max()) > 0


def test_configs1_step_at_the_benchmarked_batch():
    """Round-4 review, weak 1: the kernels of rounds 4-5 (patch-staged HALO loop, tap-skipping merged launches, marching
    kernels, 256-wide tiles, split-K tails) engage only on big grids, and every whole-model comparison above runs at B <= 2.
    This is the driver's exact configuration -- `bench.make_opt(32)`, bench.py's synthetic audio, the paired-discriminator
    `train_step` (train.py:148-184) -- value-checked three ways:
      (i)   the fp32 HIP step's generated spectrogram of sample 0 equals the ORACLE's generator on that sample's encoded
            input at 1e-4 (InstanceNorm is per sample, so one sample ties the B = 32 launch set to the oracle);
      (ii)  the bf16 step tracks the fp32 step -- losses, spectrogram, every weight gradient of both networks -- within the
            bounds of the B = 2 test above;
      (iii) the library reports HALO, tap-skipping and marching launches during the bf16 step (p2phd_launch_count)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle import model as OM
    from pix2pixhdaudiosr_amd.models.models import create_model
    from pix2pixhdaudiosr_amd import _ops, _lib
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    B = 32
    T = (bench.FRAMES - 1) * 512
    g = torch.Generator(device="cuda").manual_seed(1234)
    hr = 0.1 * torch.randn(B, T, device="cuda", generator=g)
    lr = 0.1 * torch.randn(B, T, device="cuda", generator=g)
    mask_rows = int(512 * (1 - 1 / (48000 / 8000)))
    noise = torch.randn(B, 2, mask_rows, 256, generator=torch.Generator().manual_seed(77)).cuda()   # the mask's noise, same for both runs

    torch.manual_seed(1234)
    m32 = create_model(bench.make_opt(B, dtype_bf16=False))
    sdG = {k: v.detach().float().cpu().clone().contiguous() for k, v in m32.netG.state_dict().items()}
    sdD = {k: v.detach().float().cpu().clone().contiguous() for k, v in m32.netD.state_dict().items()}
    assert sum(v.numel() for v in sdG.values()) == 102_627_170

    def run(m, count):
        L = _lib.lib_for(m.compute_dtype)
        L.p2phd_launch_count(None, 1)
        ld = m.train_step(lr, hr, noise=noise)
        torch.cuda.synchronize()
        counts = {k: int(L.p2phd_launch_count(k.encode(), 0)) for k in ("gconv", "halo", "cls_skip", "march", "march_w", "wgrad", "splitk", "tile256")}
        lr_s, sr = m._visual[0].detach().float().cpu(), m._visual[1].detach().float().cpu()
        gG = {k: p.grad.detach().float().cpu().clone() for k, p in m.netG.named_parameters()}
        gD = {k: p.grad.detach().float().cpu().clone() for k, p in m.netD.named_parameters()}
        return {k: float(v) for k, v in ld.items()}, lr_s, sr, gG, gD, counts

    l32, lr_s32, sr32, gG32, gD32, c32 = run(m32, False)
    del m32
    torch.cuda.empty_cache()

    # (i) sample 0 of the fp32 B = 32 step against the oracle's generator fed that sample's encoded input
    oo = OM.default_opt(ngf=48, netG="global", n_downsample_global=4, n_blocks_global=9, mask=True)
    with torch.no_grad():
        ref0 = OM.netG_forward(sdG, lr_s32[0:1], oo)
    e0 = rel_err(sr32[0:1].numpy(), ref0.numpy())
    print(f"B = 32 fp32 step, sample 0 vs the oracle's generator: {e0:.2e}")
    assert e0 < 1e-4, e0

    m16 = create_model(bench.make_opt(B))
    assert m16.compute_dtype == torch.bfloat16
    _load_from(m16.netG, sdG); _load_from(m16.netD, sdD)
    _ops.bump_weight_epoch()
    l16, lr_s16, sr16, gG16, gD16, c16 = run(m16, True)
    del m16
    torch.cuda.empty_cache()

    # (iii) the step under test really ran on the kernels the bench runs on: 18 trunk convs forward + 18 input gradients on the
    # HALO loop, the 3 x 3 stride-2 layers inside the 48 <-> 96 pair on tap-skipping launches, the outermost pair marching
    print(f"launch counts of the bf16 B = 32 step: {c16}; fp32: {c32}")
    assert c16["halo"] >= 36, c16
    assert c16["cls_skip"] >= 6, c16
    assert c16["march"] >= 4 and c16["march_w"] >= 2, c16
    assert c16["wgrad"] >= 18 and c16["tile256"] >= 1, c16
    assert c32["halo"] == 0 and c32["march"] == 0, c32             # fp32 is the generic loop: the two runs are different kernels

    # (ii) bf16 against fp32, the bounds of test_configs1_bf16_step_tracks_the_fp32_step_at_full_size
    assert rel_err(lr_s16.numpy(), lr_s32.numpy()) < 1e-5           # the codec is fp32 in both modes
    for k in l32:
        assert abs(l16[k] - l32[k]) <= 2e-2 * max(1.0, abs(l32[k])), (k, l16[k], l32[k])
    e_sr = rel_err(sr16.numpy(), sr32.numpy())
    worst = []
    for tag, a, b in (("G", gG16, gG32), ("D", gD16, gD32)):
        for k, v in b.items():
            if k.endswith(".weight"):
                e = rel_err(a[k].numpy(), v.numpy())
                c = float((a[k].double().flatten() @ v.double().flatten()) / max(float(a[k].double().norm() * v.double().norm()), 1e-300))
                worst.append((e, c, f"{tag}:{k}"))
    worst.sort(reverse=True)
    print(f"bf16 vs fp32 at B = 32: sr rel err {e_sr:.2e}; worst weight-gradient rel err {worst[0][0]:.2e} ({worst[0][2]}), "
          f"median {worst[len(worst) // 2][0]:.2e}, min cosine {min(c for _, c, _ in worst):.4f}")
    assert e_sr < 8e-2, e_sr
    d_err = [e for e, _, n in worst if n.startswith("D:")]
    assert max(d_err) < 1.5e-1, max(d_err)
    assert min(c for _, c, _ in worst) > 0.8
    assert [e for e, _, n in worst if n == "G:model.38.weight"][0] < 1.5e-1
