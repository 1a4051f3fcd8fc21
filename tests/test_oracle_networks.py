"""Pin oracle/networks.py (torch-CPU functional restatement) against the reference's own modules:
outputs, input/parameter gradients, state_dict key order and parameter-count KATs."""
import numpy as np
import torch

from conftest import rel_err, assert_grad_close, noise_bias_keys
from oracle import networks as N

TOL = 2e-5   # fp32 CPU vs fp32 CPU, different op grouping only


def _params(g, tag):
    keys = [str(k) for k in g[f"{tag}_keys"]]
    return {k: torch.from_numpy(g[f"{tag}_p_{k}"]).requires_grad_(True) for k in keys}, keys


def _check_G(g, tag, spec, fwd):
    p, keys = _params(g, tag)
    assert list(spec.keys()) == keys
    for k in keys:
        assert tuple(spec[k]) == tuple(p[k].shape), k
    x = torch.from_numpy(g[f"{tag}_x"]).requires_grad_(True)
    y = fwd(p, x)
    assert rel_err(y.detach().numpy(), g[f"{tag}_y"]) < TOL
    grads = torch.autograd.grad((y * torch.from_numpy(g[f"{tag}_cot"])).sum(), [x] + list(p.values()))
    assert rel_err(grads[0].numpy(), g[f"{tag}_gx"]) < 1e-4
    nb = noise_bias_keys(keys)
    for k, gr in zip(keys, grads[1:]):
        assert_grad_close(f"{tag}:{k}", gr.numpy(), g[f"{tag}_g_{k}"], noise_biases=nb)


def test_global_generator(golden_networks):
    g = golden_networks
    _check_G(g, "Gglobal", N.global_generator_spec(2, 2, 8, 2, 2),
             lambda p, x: N.global_generator_forward(p, x, 2, 2))
    _check_G(g, "Gglobal_nd4", N.global_generator_spec(2, 2, 2, 4, 1),
             lambda p, x: N.global_generator_forward(p, x, 4, 1))


def test_local_enhancer(golden_networks):
    g = golden_networks
    _check_G(g, "Glocal", N.local_enhancer_spec(2, 2, 4, 2, 2, 1, 1),
             lambda p, x: N.local_enhancer_forward(p, x, 2, 2, 1, 1))
    _check_G(g, "Glocal2", N.local_enhancer_spec(2, 2, 4, 1, 1, 2, 1),
             lambda p, x: N.local_enhancer_forward(p, x, 1, 1, 2, 1))


def test_discriminator(golden_networks):
    g = golden_networks
    for tag, gi in (("D", True), ("Dnofeat", False)):
        p, keys = _params(g, tag)
        spec = N.multiscale_discriminator_spec(4, 8, 3, 2, gi)
        assert list(spec.keys()) == keys
        x = torch.from_numpy(g[f"{tag}_x"]).requires_grad_(True)
        res = N.multiscale_discriminator_forward(p, x, 8, 3, 2, gi)
        assert [len(s) for s in res] == list(g[f"{tag}_nfeat"])
        flat = [f for s in res for f in s]
        tot = 0
        for i, f in enumerate(flat):
            assert f.shape == g[f"{tag}_f{i}"].shape
            assert rel_err(f.detach().numpy(), g[f"{tag}_f{i}"]) < TOL
            tot = tot + (f * torch.from_numpy(g[f"{tag}_c{i}"])).sum()
        grads = torch.autograd.grad(tot, [x] + list(p.values()))
        assert rel_err(grads[0].numpy(), g[f"{tag}_gx"]) < 1e-4
        nb = noise_bias_keys(keys)
        assert len(nb) == 2 * 3
        for k, gr in zip(keys, grads[1:]):
            assert_grad_close(f"{tag}:{k}", gr.numpy(), g[f"{tag}_g_{k}"], noise_biases=nb)


def test_gan_loss_kat(golden_networks):
    g = golden_networks
    pred = [[torch.from_numpy(g["ganloss_p0"])], [torch.from_numpy(g["ganloss_p1"])]]
    assert abs(float(N.gan_loss(pred, True)) - float(g["ganloss_real"])) < 1e-6
    assert abs(float(N.gan_loss(pred, False)) - float(g["ganloss_fake"])) < 1e-6


def test_param_counts_and_keys(golden_networks):
    """train_script.sh:38,49-71 KATs (156 050 690 / 730 713 346 / 5 531 522) + exact key order."""
    g = golden_networks
    counts = dict(zip([str(n) for n in g["count_names"]], [int(v) for v in g["count_values"]]))
    assert counts["G_local_ngf48_nd4_nbg3_nle1_nbl2"] == 156050690
    assert counts["G_local_ngf64_default"] == 730713346
    assert counts["D_ndf64_nl3_numD2"] == 5531522
    specs = {
        "G_local_ngf48_nd4_nbg3_nle1_nbl2": N.local_enhancer_spec(2, 2, 48, 4, 3, 1, 2),
        "G_global_ngf48_nd4_nb9": N.global_generator_spec(2, 2, 48, 4, 9),
        "G_global_ngf32_nd4_nb9": N.global_generator_spec(2, 2, 32, 4, 9),
        "G_local_ngf64_default": N.local_enhancer_spec(2, 2, 64, 4, 9, 1, 3),
        "G_local_ngf48_nd3_nb9_nle2_nbl3": N.local_enhancer_spec(2, 2, 48, 3, 9, 2, 3),
        "D_ndf64_nl3_numD2": N.multiscale_discriminator_spec(4, 64, 3, 2, True),
        "D_ndf64_nl3_numD3": N.multiscale_discriminator_spec(4, 64, 3, 3, True),
    }
    for name, spec in specs.items():
        assert N.param_count(spec) == counts[name], name
        assert list(spec.keys()) == [str(k) for k in g[f"keys_{name}"]], name


def test_three_scale_discriminator():
    """oracle vs the reference's own num_D = 3 run (tests/golden/networks_d3.npz, tools/gen_golden.py --only networks_d3)."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "networks_d3.npz"))
    tag = "D3"
    p, keys = _params(g, tag)
    spec = N.multiscale_discriminator_spec(4, 8, 3, 3, True)
    assert list(spec.keys()) == keys
    x = torch.from_numpy(g[f"{tag}_x"]).requires_grad_(True)
    res = N.multiscale_discriminator_forward(p, x, 8, 3, 3, True)
    assert [len(s) for s in res] == list(g[f"{tag}_nfeat"])
    flat = [f for s in res for f in s]
    tot = 0
    for i, f in enumerate(flat):
        assert f.shape == g[f"{tag}_f{i}"].shape
        assert rel_err(f.detach().numpy(), g[f"{tag}_f{i}"]) < TOL
        tot = tot + (f * torch.from_numpy(g[f"{tag}_c{i}"])).sum()
    grads = torch.autograd.grad(tot, [x] + list(p.values()))
    assert rel_err(grads[0].numpy(), g[f"{tag}_gx"]) < 1e-4
    nb = noise_bias_keys(keys)
    for k, gr in zip(keys, grads[1:]):
        assert_grad_close(f"{tag}:{k}", gr.numpy(), g[f"{tag}_g_{k}"], noise_biases=nb)
