"""pix2pixHD generator / discriminator with the reference's class names, constructor arguments,
module tree and state_dict keys (models/networks.py of the reference), executed by the HIP conv stack.

The module tree is built from ordinary ``torch.nn`` layer objects, used purely as parameter containers
and structure markers, so ``state_dict()``, ``weights_init``, ``print(net)``, checkpoints written by the
reference (``model.1.weight``, ``scale0_layer0.0.weight`` ...) and torch optimisers all see exactly what
they see on the reference.  ``forward`` never calls those layers: at first use each ``nn.Sequential`` is
compiled into fused steps  [ReflectionPad2d] -> Conv2d/ConvTranspose2d -> [InstanceNorm2d] -> [ReLU |
LeakyReLU | Tanh] (-> [+ residual]), each of which is one implicit-GEMM MFMA launch plus at most one
HBM-bound normalisation launch of libp2phd_hip.so (see _ops.py, csrc/conv.hip, csrc/norm.hip).

Not implemented on this path (raise instead of falling back): BatchNorm ('--norm batch'), Dropout in
ResnetBlock, the deprecated Encoder / Vgg19 / VGGLoss leftovers of upstream pix2pixHD.
"""
import functools

import numpy as np
import os

import torch
import torch.nn as nn

from .. import _ops
from .._ops import ConvSpec, conv_block

# compute dtype of networks built without an explicit dtype: float32 = exact-f32 MFMA (parity runs),
# bfloat16 = bf16 MFMA with fp32 accumulation (what `--fp16` selects in Pix2PixHDModel).
DEFAULT_COMPUTE_DTYPE = torch.float32


###############################################################################
# Functions
###############################################################################
def weights_init(m):
    """N(0, 0.02) on every layer whose class name contains 'Conv' (reference networks.py:10-16)."""
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        m.weight.data.normal_(0.0, 0.02)
    elif classname.find('BatchNorm2d') != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def get_norm_layer(norm_type='instance'):
    if norm_type == 'instance':
        return functools.partial(nn.InstanceNorm2d, affine=False)
    if norm_type == 'batch':
        raise NotImplementedError("BatchNorm2d is outside the HIP hot path; use norm='instance'")
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)


def define_G(input_nc, output_nc, ngf, netG, n_downsample_global=3, n_blocks_global=9, n_local_enhancers=1,
             n_blocks_local=3, norm='instance', gpu_ids=[], dtype=None, verbose=True):
    norm_layer = get_norm_layer(norm_type=norm)
    if netG == 'global':
        net = GlobalGenerator(input_nc, output_nc, ngf, n_downsample_global, n_blocks_global, norm_layer)
    elif netG == 'local':
        net = LocalEnhancer(input_nc, output_nc, ngf, n_downsample_global, n_blocks_global,
                            n_local_enhancers, n_blocks_local, norm_layer)
    else:
        raise NotImplementedError('generator [%s] is not implemented on the HIP path' % netG)
    if verbose:
        print(net)
    if len(gpu_ids) > 0:
        assert (torch.cuda.is_available())
        net.cuda(gpu_ids[0])
    net.apply(weights_init)
    if dtype is not None:
        net.compute_dtype = dtype
    net.compile_all()
    return net


def define_D(input_nc, ndf, n_layers_D, norm='instance', use_sigmoid=False, num_D=1, getIntermFeat=False, gpu_ids=[],
             dtype=None, verbose=True):
    norm_layer = get_norm_layer(norm_type=norm)
    net = MultiscaleDiscriminator(input_nc, ndf, n_layers_D, norm_layer, use_sigmoid, num_D, getIntermFeat)
    if verbose:
        print(net)
    if len(gpu_ids) > 0:
        assert (torch.cuda.is_available())
        net.cuda(gpu_ids[0])
    net.apply(weights_init)
    if dtype is not None:
        net.compute_dtype = dtype
    net.compile_all()
    return net


def print_network(net):
    if isinstance(net, list):
        net = net[0]
    num_params = sum(param.numel() for param in net.parameters())
    print(net)
    print('Total number of parameters: %d' % num_params)


###############################################################################
# Sequential -> fused HIP steps
###############################################################################
_ACTS = {nn.ReLU: _ops.ACT_RELU, nn.LeakyReLU: _ops.ACT_LRELU, nn.Tanh: _ops.ACT_TANH}


def _one(v):
    return v[0] if isinstance(v, (tuple, list)) else v


class _ConvStep:
    def __init__(self, conv, spec):
        self.conv, self.spec = conv, spec
        # big stride-1 layers (the residual trunk, D 256 -> 512): optim.FlatAdam keeps their master weights K-major
        # ([K][R][S][C]: the packed forward row IS the master row, see include/p2phd.h) and hands torch a permuted view
        if _ops.kmajor_eligible(spec.cin, spec.cout, spec.k, spec.stride, spec.transposed):
            conv.weight._p2phd_kmajor = True

    def run(self, x, residual=None, link=None, exclusive=False, defer=False):
        return conv_block(x, self.conv.weight, self.conv.bias, self.spec, residual, link, exclusive, defer)


class _ResStep:
    def __init__(self, block):
        self.a, self.b = _compile(block.conv_block)
        assert isinstance(self.a, _ConvStep) and isinstance(self.b, _ConvStep)

    def run(self, x, residual=None, exclusive=False):
        assert residual is None
        link = _ops.SkipLink()                              # skip-path gradient is added inside conv a's dgrad
        # x is read twice (conv a and the skip): never exclusive; a's output goes to b alone
        return self.b.run(self.a.run(x, link=link), residual=x, link=link, exclusive=True)   # x + conv_block(x), reference networks.py:252


def _compile(seq):
    """Group the layers of an nn.Sequential into fused steps."""
    # index, do not iterate children(): the generators reuse ONE ReLU object at many positions and
    # children() would de-duplicate it
    mods = [seq[i] for i in range(len(seq))]
    steps, i = [], 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, ResnetBlock):
            steps.append(_ResStep(m)); i += 1
            continue
        pad, pad_mode = 0, 0
        if isinstance(m, nn.ReflectionPad2d):
            pad, pad_mode = _one(m.padding), 1
            i += 1
            m = mods[i]
            if not isinstance(m, nn.Conv2d) or _one(m.padding) != 0:
                raise NotImplementedError("ReflectionPad2d must be followed by an unpadded Conv2d")
        if isinstance(m, nn.ConvTranspose2d):
            cin, cout, transposed, opad = m.in_channels, m.out_channels, True, _one(m.output_padding)
            pad = _one(m.padding)
        elif isinstance(m, nn.Conv2d):
            cin, cout, transposed, opad = m.in_channels, m.out_channels, False, 0
            if not pad_mode:
                pad = _one(m.padding)
        elif isinstance(m, nn.Sequential):
            steps.extend(_compile(m)); i += 1
            continue
        else:
            raise NotImplementedError("layer %s has no HIP implementation in this position" % type(m).__name__)
        k, stride = _one(m.kernel_size), _one(m.stride)
        if m.kernel_size[0] != m.kernel_size[1] or m.stride[0] != m.stride[1] or _one(m.dilation) != 1 or m.groups != 1:
            raise NotImplementedError("only square, undilated, ungrouped convolutions are on the hot path")
        i += 1
        norm, act = False, _ops.ACT_NONE
        if i < len(mods) and isinstance(mods[i], nn.InstanceNorm2d):
            if mods[i].affine or mods[i].track_running_stats:
                raise NotImplementedError("InstanceNorm2d(affine/running stats) is not on the hot path")
            norm = True; i += 1
        elif i < len(mods) and isinstance(mods[i], nn.BatchNorm2d):
            raise NotImplementedError("BatchNorm2d is outside the HIP hot path")
        if i < len(mods) and type(mods[i]) in _ACTS:
            if isinstance(mods[i], nn.LeakyReLU) and abs(mods[i].negative_slope - 0.2) > 1e-12:
                raise NotImplementedError("LeakyReLU slope must be 0.2")
            act = _ACTS[type(mods[i])]; i += 1
        if i < len(mods) and isinstance(mods[i], nn.Dropout):
            raise NotImplementedError("Dropout is not on the hot path (the reference never enables it)")
        steps.append(_ConvStep(m, ConvSpec(cin, cout, k, stride, pad, pad_mode, transposed, opad, norm, act)))
    return steps


def _flat_conv_steps(steps):
    out = []
    for s in steps:
        if isinstance(s, _ResStep):
            out += [s.a, s.b]
        else:
            out.append(s)
    return out


def enable_fp8(net, min_channels=256):
    """BASELINE configs[4] ("bf16 + fp8 MFMA conv weights"): run the forward of the wide stride-1 convs (residual trunk,
    discriminator 256 -> 512) on OCP e4m3 operands.  A layer qualifies when its input comes straight out of an InstanceNorm
    pass of this network (which then also writes the e4m3 twin, scale 1), it has >= `min_channels` input channels and the
    kernel's shape constraints hold.  fp32 master weights, bf16 activations and the whole backward pass are unchanged.
    Returns the number of layers switched."""
    n = 0
    if hasattr(net, '_scale_steps'):     # discriminator: one chain per scale (each stage is its own Sequential with getIntermFeat)
        chains = [[s for stage in net._scale_steps(d) for s in stage] for d in range(net.num_D)]
    else:
        chains = [net._steps(k) for k in net._modules if isinstance(getattr(net, k), nn.Sequential)]
    for chain in chains:
        convs = _flat_conv_steps(chain)
        for prev, cur in zip(convs[:-1], convs[1:]):
            sp = cur.spec
            ok = (prev.spec.norm and not sp.transposed and sp.stride == 1 and sp.cin >= min_channels and sp.cin % 16 == 0
                  and (sp.k * sp.k * sp.cin) % 128 == 0 and sp.cin > 4 and sp.cout > 4)
            if ok:
                sp.fp8, prev.spec.emit_q8 = True, True
                n += 1
    return n


def _run(steps, x, residual_last=None, cuts=None, cut_after=(), first_exclusive=False):
    """Run fused steps in order.  `cuts` (a list) collects the output of every step whose index is in `cut_after`: the
    points where the staged backward of the data-parallel step hands over (Pix2PixHDModel._phase_a)."""
    for j, s in enumerate(steps):
        # from the second step on, the input is the previous step's output and is consumed here only (a cut tensor is an
        # endpoint of torch.autograd.grad, not a consumer): the step may fuse the producer's backward sums (_ops)
        if isinstance(s, _ConvStep):
            # the next step normalises on load (csrc/march.hip): this one hands out its raw output + statistics (_ops.LazyNorm)
            nxt = steps[j + 1] if j + 1 < len(steps) else None
            defer = (isinstance(nxt, _ConvStep) and s.spec.norm and x.dtype in _ops.HALF_DTYPES and _ops.lazy_static_ok(nxt.spec)
                     and os.environ.get("P2PHD_LAZY", "1") != "0")
            x = s.run(x, residual_last if j == len(steps) - 1 else None, exclusive=j > 0 or first_exclusive, defer=defer)
        else:
            x = s.run(x, residual_last if j == len(steps) - 1 else None, exclusive=j > 0 or first_exclusive)
        if cuts is not None and j in cut_after:
            cuts.append(x)
    return x


def _balanced_cuts(counts, n_buckets):
    """Indices j such that a cut AFTER step j splits a chain with `counts` parameters per step into `n_buckets` runs of
    roughly equal size, assigned from the end (the backward's order)."""
    cut_after, acc, remaining, left = [], 0, sum(counts), n_buckets
    for j in range(len(counts) - 1, 0, -1):
        acc += counts[j]
        if left > 1 and acc >= remaining / left:              # this bucket holds its share of what is still unassigned
            cut_after.append(j - 1)
            remaining -= acc
            left -= 1
            acc = 0
    return sorted(set(cut_after))


def _step_params(step):
    if isinstance(step, _ResStep):
        return _step_params(step.a) + _step_params(step.b)
    return [p for p in (step.conv.weight, step.conv.bias) if p is not None]


class _HipNet(nn.Module):
    """Common plumbing: compute dtype, lazy compilation, NCHW <-> physical conversion."""
    compute_dtype = None

    def _dtype(self):
        return self.compute_dtype or DEFAULT_COMPUTE_DTYPE

    def _steps(self, name):
        cache = self.__dict__.setdefault('_compiled', {})
        if name not in cache:
            cache[name] = _compile(getattr(self, name))
        return cache[name]

    def compile_all(self):
        """Compile every Sequential now (specs only, nothing touches the GPU): marks the parameters whose master copy the
        optimiser should keep K-major BEFORE the optimiser lays out its flat buffers."""
        for name, mod in self.named_children():
            if isinstance(mod, nn.Sequential):
                self._steps(name)
        return self

    def _to_phys(self, x):
        if not x.is_cuda:
            raise RuntimeError("this network runs on the HIP kernels only: move the input to the GPU")
        return _ops.ToPhysical.apply(self._dtype(), x)


def _centered_input(net, x):
    """bf16 only: the generator's input minus its per-(sample, channel) mean.  The first layer of both generators is
    ReflectionPad2d + Conv2d + InstanceNorm2d (networks.py:190, :268): a constant added to an input channel adds a constant
    to every output channel (reflection padding keeps it constant up to the border), which the normalisation removes --
    the result is unchanged in exact arithmetic.  In bf16 it is not: a [0, 1]-scaled dB spectrogram has |mean| / sigma of
    10-25 per channel, so rounding x itself to 8 mantissa bits is noise of 5-10 % of the channel's SIGNAL, and the raw conv
    output (stored in bf16 before its normalisation) carries the same offset.  Centred, both are rounded relative to the
    signal.  Measured at configs[1] size: generated spectrogram bf16-vs-fp32 error 1.2e-1 -> see tests/test_gpu_fullsize.py."""
    if net._dtype() not in _ops.HALF_DTYPES:
        return x
    first = _flat_conv_steps(net._steps(net._first_seq))[0].spec
    if not (first.norm and first.pad_mode == 1):
        return x
    return x - x.mean(dim=(2, 3), keepdim=True)


def _view(t_phys, channels):
    """Reference-shaped [N,C,H,W] view of a physical tensor; remembers the physical tensor for the HIP losses."""
    v = t_phys.permute(0, 3, 1, 2)[:, :channels]
    v._p2phd_phys = (t_phys, channels)
    return v


def as_physical(t, dtype=None):
    """(physical tensor, channels) of a tensor produced by these networks, or of any NCHW tensor."""
    got = getattr(t, '_p2phd_phys', None)
    if got is not None:
        return got
    return _ops.ToPhysical.apply(dtype or DEFAULT_COMPUTE_DTYPE, t.float()), int(t.shape[1])


##############################################################################
# Losses
##############################################################################
class GANLoss(nn.Module):
    """LSGAN criterion of the reference (networks.py:68-110): MSE of the last feature of every scale
    against a constant 1.0 / 0.0 target, summed over scales.  use_lsgan=False (BCE) is not on the path."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0, tensor=torch.FloatTensor):
        super(GANLoss, self).__init__()
        if not use_lsgan:
            raise NotImplementedError("only the LSGAN criterion is implemented on the HIP path")
        self.real_label = target_real_label
        self.fake_label = target_fake_label
        self.Tensor = tensor

    def _one(self, pred, target_is_real):
        t, c = as_physical(pred)
        return _ops.mse_const_loss(t, c, self.real_label if target_is_real else self.fake_label)

    def __call__(self, input, target_is_real):
        if isinstance(input[0], list):
            loss = 0
            for input_i in input:
                loss = loss + self._one(input_i[-1], target_is_real)
            return loss
        return self._one(input[-1], target_is_real)


class FeatLoss(nn.Module):
    """criterionFeat = L1Loss between two feature maps (pix2pixHD_model.py:99); target is not differentiated."""

    def forward(self, a, b):
        ta, c = as_physical(a)
        tb, _ = as_physical(b.detach() if isinstance(b, torch.Tensor) else b)
        return _ops.l1_loss(ta, tb, c)


##############################################################################
# Generator
##############################################################################
def _c7(cin, cout, norm_layer=None, act=None):
    layers = [nn.ReflectionPad2d(3), nn.Conv2d(cin, cout, kernel_size=7, padding=0)]
    if norm_layer is not None:
        layers.append(norm_layer(cout))
    if act is not None:
        layers.append(act)
    return layers


class GlobalGenerator(_HipNet):
    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, norm_layer=nn.BatchNorm2d,
                 padding_type='reflect'):
        assert (n_blocks >= 0)
        super(GlobalGenerator, self).__init__()
        self.input_nc, self.output_nc = input_nc, output_nc
        relu = nn.ReLU(True)
        layers = _c7(input_nc, ngf, norm_layer, relu)
        ch = ngf
        for _ in range(n_downsampling):                       # stride-2 encoder
            layers += [nn.Conv2d(ch, ch * 2, kernel_size=3, stride=2, padding=1), norm_layer(ch * 2), relu]
            ch *= 2
        layers += [ResnetBlock(ch, padding_type=padding_type, activation=relu, norm_layer=norm_layer) for _ in range(n_blocks)]
        for _ in range(n_downsampling):                       # stride-2 decoder
            layers += [nn.ConvTranspose2d(ch, ch // 2, kernel_size=3, stride=2, padding=1, output_padding=1),
                       norm_layer(ch // 2), relu]
            ch //= 2
        layers += _c7(ngf, output_nc, None, nn.Tanh())
        self.model = nn.Sequential(*layers)

    def forward_physical(self, x, cuts=None, cut_after=()):
        return _run(self._steps('model'), x, cuts=cuts, cut_after=cut_after)

    def bucket_plan(self, n_buckets):
        """Cut points for a backward in `n_buckets` stages of roughly equal parameter bytes (the gradient
        all-reduce of a bucket starts when its last weight gradient is done).  Parameters are registered in execution
        order, so the layers after a cut own a contiguous TAIL of the flat gradient buffer.
        Returns (cut_after, first_param_index): step indices in execution order, and for every cut the index (into
        list(self.parameters())) of the first parameter of the step that follows it."""
        steps = self._steps('model')
        if n_buckets <= 1 or len(steps) < 2:
            return [], []
        cut_after = _balanced_cuts([sum(p.numel() for p in _step_params(s)) for s in steps], n_buckets)
        ids = [id(p) for p in self.parameters()]
        first = [ids.index(id(_step_params(steps[j + 1])[0])) for j in cut_after]
        return cut_after, first

    _first_seq = 'model'

    def input_physical(self, x):
        """NCHW f32 -> physical input of forward_physical (centred in bf16 mode, see _centered_input)."""
        return self._to_phys(_centered_input(self, x))

    def forward(self, input):
        return _ops.FromPhysical.apply(self.forward_physical(self.input_physical(input)), self.output_nc)


class LocalEnhancer(_HipNet):
    def __init__(self, input_nc, output_nc, ngf=32, n_downsample_global=3, n_blocks_global=9,
                 n_local_enhancers=1, n_blocks_local=3, norm_layer=nn.BatchNorm2d, padding_type='reflect'):
        super(LocalEnhancer, self).__init__()
        self.n_local_enhancers = n_local_enhancers
        self.input_nc, self.output_nc = input_nc, output_nc

        # coarsest level: a GlobalGenerator without its output head (ReflectionPad2d, Conv2d 7x7, Tanh)
        trunk = GlobalGenerator(input_nc, output_nc, ngf * (2 ** n_local_enhancers), n_downsample_global,
                                n_blocks_global, norm_layer).model
        self.model = nn.Sequential(*[trunk[i] for i in range(len(trunk) - 3)])

        for n in range(1, n_local_enhancers + 1):
            g = ngf * (2 ** (n_local_enhancers - n))
            head = _c7(input_nc, g, norm_layer, nn.ReLU(True))
            head += [nn.Conv2d(g, g * 2, kernel_size=3, stride=2, padding=1), norm_layer(g * 2), nn.ReLU(True)]
            tail = [ResnetBlock(g * 2, padding_type=padding_type, norm_layer=norm_layer) for _ in range(n_blocks_local)]
            tail += [nn.ConvTranspose2d(g * 2, g, kernel_size=3, stride=2, padding=1, output_padding=1),
                     norm_layer(g), nn.ReLU(True)]
            if n == n_local_enhancers:
                tail += _c7(ngf, output_nc, None, nn.Tanh())
            setattr(self, 'model' + str(n) + '_1', nn.Sequential(*head))
            setattr(self, 'model' + str(n) + '_2', nn.Sequential(*tail))

        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)

    def bucket_plan(self, n_buckets):
        """Staged backward for the data-parallel exchange (round 3; one local enhancer).  The backward graph is a tail
        (model1_2), then TWO parallel branches behind the sum `model1_1(x) + model(pool(x))`: the enhancer's head and the
        global chain.  Parameters are registered global chain first, then model1_1, then model1_2, so everything outside
        the global chain is one contiguous tail of the flat gradient buffer: the first cut is always the global chain's
        OUTPUT (bucket 0 = the enhancer's own layers, `staged_head_anchors` makes the head stage run the parallel branch
        too), the remaining n_buckets - 1 buckets balance the global chain like GlobalGenerator.bucket_plan."""
        steps = self._steps('model')
        if n_buckets <= 1 or self.n_local_enhancers != 1 or len(steps) < 2:
            return [], []
        counts = [sum(p.numel() for p in _step_params(s)) for s in steps]
        inner = _balanced_cuts(counts, n_buckets - 1) if n_buckets > 2 else []
        cut_after = sorted(set(inner + [len(steps) - 1]))
        ids = [id(p) for p in self.parameters()]
        first = []
        for j in cut_after:
            nxt = _step_params(steps[j + 1])[0] if j + 1 < len(steps) else _step_params(_flat_conv_steps(self._steps('model1_1'))[0])[0]
            first.append(ids.index(id(nxt)))
        return cut_after, first

    def staged_head_anchors(self):
        """Tensors the head stage of a staged backward must ALSO differentiate, so that autograd runs the branch that is
        parallel to the last cut (the enhancer's head, whose weight gradients land in bucket 0): its first conv weight."""
        return [_flat_conv_steps(self._steps('model1_1'))[0].conv.weight]

    def forward_physical(self, x, cuts=None, cut_after=()):
        pyramid = [x]
        for _ in range(self.n_local_enhancers):
            pyramid.append(_ops.avgpool(pyramid[-1], self.input_nc))
        out = _run(self._steps('model'), pyramid[-1], cuts=cuts, cut_after=cut_after)
        for n in range(1, self.n_local_enhancers + 1):
            xi = pyramid[self.n_local_enhancers - n]
            # model{n}_1(x_i) + out : the sum rides on the last InstanceNorm+ReLU launch of the head
            h = _run(self._steps('model%d_1' % n), xi, residual_last=out)
            out = _run(self._steps('model%d_2' % n), h)
        return out

    _first_seq = 'model1_1'          # (the coarsest level's first layer has the same form: 7x7 behind ReflectionPad2d + InstanceNorm)

    def input_physical(self, x):
        """NCHW f32 -> physical input of forward_physical (centred in bf16 mode: the average pooling of the pyramid maps
        a constant to the same constant, so every level sees its input shifted by it; see _centered_input)."""
        return self._to_phys(_centered_input(self, x))

    def forward(self, input):
        return _ops.FromPhysical.apply(self.forward_physical(self.input_physical(input)), self.output_nc)


class ResnetBlock(nn.Module):
    """x + [ReflectionPad2d(1), Conv3x3, IN, ReLU, ReflectionPad2d(1), Conv3x3, IN](x)."""

    def __init__(self, dim, padding_type, norm_layer, activation=nn.ReLU(True), use_dropout=False):
        super(ResnetBlock, self).__init__()
        if padding_type != 'reflect':
            raise NotImplementedError('padding [%s] is not implemented on the HIP path' % padding_type)
        if use_dropout:
            raise NotImplementedError('Dropout is not on the hot path')
        self.conv_block = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, kernel_size=3, padding=0),
                                        norm_layer(dim), activation,
                                        nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, kernel_size=3, padding=0),
                                        norm_layer(dim))

    def forward(self, x):
        raise RuntimeError("ResnetBlock runs as part of its generator (fused HIP steps), not stand-alone")


##############################################################################
# Discriminator
##############################################################################
class NLayerDiscriminator(_HipNet):
    """PatchGAN: Conv4x4 s2 + LReLU | (Conv4x4 s2 + IN + LReLU) x (n_layers-1) | Conv4x4 s1 + IN + LReLU | Conv4x4 s1 -> 1."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_sigmoid=False, getIntermFeat=False):
        super(NLayerDiscriminator, self).__init__()
        if use_sigmoid:
            raise NotImplementedError("Sigmoid head (no_lsgan) is not on the HIP path")
        self.getIntermFeat = getIntermFeat
        self.n_layers = n_layers
        kw, padw = 4, int(np.ceil((4 - 1.0) / 2))
        stages = [[nn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]]
        nf = ndf
        for n in range(1, n_layers + 1):
            nf_prev, nf = nf, min(nf * 2, 512)
            stages.append([nn.Conv2d(nf_prev, nf, kernel_size=kw, stride=2 if n < n_layers else 1, padding=padw),
                           norm_layer(nf), nn.LeakyReLU(0.2, True)])
        stages.append([nn.Conv2d(nf, 1, kernel_size=kw, stride=1, padding=padw)])
        self.channels = [s[0].out_channels for s in stages]
        if getIntermFeat:
            for n, s in enumerate(stages):
                setattr(self, 'model' + str(n), nn.Sequential(*s))
        else:
            self.model = nn.Sequential(*[l for s in stages for l in s])


class MultiscaleDiscriminator(_HipNet):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_sigmoid=False, num_D=3, getIntermFeat=False):
        super(MultiscaleDiscriminator, self).__init__()
        self.num_D = num_D
        self.n_layers = n_layers
        self.getIntermFeat = getIntermFeat
        self.input_nc = input_nc
        for i in range(num_D):
            netD = NLayerDiscriminator(input_nc, ndf, n_layers, norm_layer, use_sigmoid, getIntermFeat)
            self.channels = netD.channels
            if getIntermFeat:
                for j in range(n_layers + 2):
                    setattr(self, 'scale' + str(i) + '_layer' + str(j), getattr(netD, 'model' + str(j)))
            else:
                setattr(self, 'layer' + str(i), netD.model)
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)

    def _scale_steps(self, d):
        if self.getIntermFeat:
            return [self._steps('scale%d_layer%d' % (d, j)) for j in range(self.n_layers + 2)]
        flat = self._steps('layer%d' % d)
        return [[s] for s in flat]

    def forward_physical(self, x, exclusive=False):
        """list[num_D] of list of (physical tensor, channels); all stages if getIntermFeat else the last only.
        `exclusive`: the caller promises that the only gradient an intermediate feature receives besides its next stage's
        comes from `_ops.l1_loss(..., park=True)` (Pix2PixHDModel._losses): the stages then form an exclusive chain."""
        result, cur = [], x
        for i in range(self.num_D):
            if i != self.num_D - 1:
                cur._p2phd_pool_link = _ops.PoolLink()              # `cur` feeds this scale's first conv AND the pooling below
            stages = self._scale_steps(self.num_D - 1 - i)          # reference networks.py:325
            feats, h = [], cur
            for j, st in enumerate(stages):
                h = _run(st, h, first_exclusive=exclusive and j > 0)
                feats.append((h, self.channels[j]))
            result.append(feats if self.getIntermFeat else [feats[-1]])
            if i != self.num_D - 1:
                cur = _ops.avgpool(cur, self.input_nc)
        return result

    def forward(self, input):
        res = self.forward_physical(self._to_phys(input))
        return [[_view(t, c) for (t, c) in scale] for scale in res]
