"""Oracle: pix2pixHD generator / discriminator forward as plain torch-CPU fp32 functional ops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the layer sequences of the reference (paths relative to the reference
checkout) over an explicit ``{state_dict key: tensor}`` parameter dict so that
the same weights can be pushed through the oracle, the reference and the HIP path:

  GlobalGenerator          models/networks.py:183-211
  LocalEnhancer            models/networks.py:129-181
  ResnetBlock              models/networks.py:214-253
  MultiscaleDiscriminator  models/networks.py:292-331
  NLayerDiscriminator      models/networks.py:334-383
  InstanceNorm2d(affine=False), eps 1e-5   models/networks.py:22
  AvgPool2d(3, 2, [1,1], count_include_pad=False)   models/networks.py:165,308
  GANLoss (LSGAN)          models/networks.py:68-110
  weights_init             models/networks.py:10-16

Gradients come from torch autograd over these functional ops.
"""
from collections import OrderedDict
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# parameter specs: ordered {key: shape}, reference state_dict names
# ----------------------------------------------------------------------------

def _conv(spec, key, cout, cin, k):
    spec[key + ".weight"] = (cout, cin, k, k)
    spec[key + ".bias"] = (cout,)


def _convT(spec, key, cin, cout, k):
    spec[key + ".weight"] = (cin, cout, k, k)
    spec[key + ".bias"] = (cout,)


def global_generator_spec(input_nc, output_nc, ngf, n_down, n_blocks, prefix="model", strip_last=False):
    """networks.py:190-208; ``strip_last`` drops ReflPad/Conv7/Tanh as LocalEnhancer does (:138)."""
    spec = OrderedDict()
    _conv(spec, f"{prefix}.1", ngf, input_nc, 7)
    idx = 4
    for i in range(n_down):
        m = 2 ** i
        _conv(spec, f"{prefix}.{idx}", ngf * m * 2, ngf * m, 3)
        idx += 3
    dim = ngf * 2 ** n_down
    for _ in range(n_blocks):
        _conv(spec, f"{prefix}.{idx}.conv_block.1", dim, dim, 3)
        _conv(spec, f"{prefix}.{idx}.conv_block.5", dim, dim, 3)
        idx += 1
    for i in range(n_down):
        m = 2 ** (n_down - i)
        _convT(spec, f"{prefix}.{idx}", ngf * m, ngf * m // 2, 3)
        idx += 3
    if not strip_last:
        _conv(spec, f"{prefix}.{idx + 1}", output_nc, ngf, 7)
    return spec


def local_enhancer_spec(input_nc, output_nc, ngf, n_down_global, n_blocks_global, n_local, n_blocks_local):
    """networks.py:135-163."""
    spec = global_generator_spec(input_nc, output_nc, ngf * 2 ** n_local, n_down_global,
                                 n_blocks_global, "model", strip_last=True)
    for n in range(1, n_local + 1):
        g = ngf * 2 ** (n_local - n)
        _conv(spec, f"model{n}_1.1", g, input_nc, 7)
        _conv(spec, f"model{n}_1.4", g * 2, g, 3)
        for i in range(n_blocks_local):
            _conv(spec, f"model{n}_2.{i}.conv_block.1", g * 2, g * 2, 3)
            _conv(spec, f"model{n}_2.{i}.conv_block.5", g * 2, g * 2, 3)
        _convT(spec, f"model{n}_2.{n_blocks_local}", g * 2, g, 3)
        if n == n_local:
            _conv(spec, f"model{n}_2.{n_blocks_local + 4}", output_nc, ngf, 7)
    return spec


def _nlayer_channels(input_nc, ndf, n_layers):
    """(cin, cout, stride, has_norm, has_act) per stage, networks.py:340-361."""
    st = [(input_nc, ndf, 2, False, True)]
    nf = ndf
    for _ in range(1, n_layers):
        prev, nf = nf, min(nf * 2, 512)
        st.append((prev, nf, 2, True, True))
    prev, nf = nf, min(nf * 2, 512)
    st.append((prev, nf, 1, True, True))
    st.append((nf, 1, 1, False, False))
    return st


def multiscale_discriminator_spec(input_nc, ndf, n_layers, num_D, get_interm_feat=True):
    """networks.py:300-306,366-373."""
    spec = OrderedDict()
    stages = _nlayer_channels(input_nc, ndf, n_layers)
    for i in range(num_D):
        if get_interm_feat:
            for j, (cin, cout, _, _, _) in enumerate(stages):
                _conv(spec, f"scale{i}_layer{j}.0", cout, cin, 4)
        else:
            idx = 0
            for (cin, cout, _, has_norm, has_act) in stages:
                _conv(spec, f"layer{i}.{idx}", cout, cin, 4)
                idx += 1 + int(has_norm) + int(has_act)
    return spec


def init_params(spec, seed=0, std=0.02):
    """weights_init (networks.py:10-16): conv weights ~ N(0, 0.02); biases keep the
    PyTorch default U(-1/sqrt(fan_in), 1/sqrt(fan_in)).  Fixtures ship the values, not the seed."""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for k, shp in spec.items():
        if k.endswith(".weight"):
            out[k] = torch.randn(shp, generator=g) * std
        else:
            wshape = spec[k[:-5] + ".weight"]
            is_T = False
            fan_in = wshape[1] * wshape[2] * wshape[3]
            b = 1.0 / math.sqrt(fan_in)
            out[k] = (torch.rand(shp, generator=g) * 2 - 1) * b
    return out


def param_count(spec):
    return sum(int(torch.Size(s).numel()) for s in spec.values())


# ----------------------------------------------------------------------------
# functional forward
# ----------------------------------------------------------------------------

def _in(x):
    return F.instance_norm(x, eps=1e-5)


def _c7(p, key, x, norm_relu=True):
    x = F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), p[key + ".weight"], p[key + ".bias"])
    return F.relu(_in(x)) if norm_relu else x


def _down(p, key, x):
    return F.relu(_in(F.conv2d(x, p[key + ".weight"], p[key + ".bias"], stride=2, padding=1)))


def _up(p, key, x):
    return F.relu(_in(F.conv_transpose2d(x, p[key + ".weight"], p[key + ".bias"], stride=2,
                                         padding=1, output_padding=1)))


def _resblock(p, key, x):
    y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), p[key + ".conv_block.1.weight"], p[key + ".conv_block.1.bias"])
    y = F.relu(_in(y))
    y = F.conv2d(F.pad(y, (1, 1, 1, 1), mode="reflect"), p[key + ".conv_block.5.weight"], p[key + ".conv_block.5.bias"])
    return x + _in(y)


def global_generator_forward(p, x, n_down, n_blocks, prefix="model", strip_last=False):
    x = _c7(p, f"{prefix}.1", x)
    idx = 4
    for _ in range(n_down):
        x = _down(p, f"{prefix}.{idx}", x)
        idx += 3
    for _ in range(n_blocks):
        x = _resblock(p, f"{prefix}.{idx}", x)
        idx += 1
    for _ in range(n_down):
        x = _up(p, f"{prefix}.{idx}", x)
        idx += 3
    if strip_last:
        return x
    return torch.tanh(_c7(p, f"{prefix}.{idx + 1}", x, norm_relu=False))


def avgpool(x):
    return F.avg_pool2d(x, 3, stride=2, padding=[1, 1], count_include_pad=False)


def local_enhancer_forward(p, x, n_down_global, n_blocks_global, n_local, n_blocks_local):
    pyr = [x]
    for _ in range(n_local):
        pyr.append(avgpool(pyr[-1]))                              # networks.py:169-171
    out = global_generator_forward(p, pyr[-1], n_down_global, n_blocks_global, "model", strip_last=True)
    for n in range(1, n_local + 1):
        xi = pyr[n_local - n]
        h = _c7(p, f"model{n}_1.1", xi)
        h = _down(p, f"model{n}_1.4", h)
        h = h + out                                               # networks.py:180
        for i in range(n_blocks_local):
            h = _resblock(p, f"model{n}_2.{i}", h)
        h = _up(p, f"model{n}_2.{n_blocks_local}", h)
        if n == n_local:
            h = torch.tanh(_c7(p, f"model{n}_2.{n_blocks_local + 4}", h, norm_relu=False))
        out = h
    return out


def multiscale_discriminator_forward(p, x, ndf, n_layers, num_D, get_interm_feat=True):
    """Returns list[num_D] of list of feature maps (all stages if get_interm_feat else the last only)."""
    stages = _nlayer_channels(x.shape[1], ndf, n_layers)
    result = []
    cur = x
    for i in range(num_D):
        d = num_D - 1 - i                                         # networks.py:325
        feats = []
        h = cur
        idx = 0
        for j, (_, _, stride, has_norm, has_act) in enumerate(stages):
            key = f"scale{d}_layer{j}.0" if get_interm_feat else f"layer{d}.{idx}"
            h = F.conv2d(h, p[key + ".weight"], p[key + ".bias"], stride=stride, padding=2)
            if has_norm:
                h = _in(h)
            if has_act:
                h = F.leaky_relu(h, 0.2)
            idx += 1 + int(has_norm) + int(has_act)
            feats.append(h)
        result.append(feats if get_interm_feat else [feats[-1]])
        if i != num_D - 1:
            cur = avgpool(cur)
    return result


def gan_loss(pred, target_is_real):
    """LSGAN: sum over scales of MSE(last feature, 1.0 / 0.0)   (networks.py:100-110)."""
    tgt = 1.0 if target_is_real else 0.0
    loss = 0
    for scale in pred:
        last = scale[-1]
        loss = loss + F.mse_loss(last, torch.full_like(last, tgt))
    return loss


def feature_matching_loss(pred_fake, pred_real, n_layers_D, num_D, lambda_feat):
    """models/pix2pixHD_model.py:391-398."""
    feat_w = 4.0 / (n_layers_D + 1)
    d_w = 1.0 / num_D
    loss = 0
    for i in range(num_D):
        for j in range(len(pred_fake[i]) - 1):
            loss = loss + d_w * feat_w * F.l1_loss(pred_fake[i][j], pred_real[i][j].detach()) * lambda_feat
    return loss
