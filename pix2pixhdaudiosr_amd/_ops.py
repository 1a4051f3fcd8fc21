"""Host-side glue between torch tensors and the conv-stack entry points of libp2phd_hip.so.

Internal activation format ("physical" tensors): contiguous NHWC ``[N, H, W, Cp]`` in the compute dtype
(torch.bfloat16 / torch.float16 or torch.float32), Cp = channels rounded up to 8 with zero pad channels.  Everything here
is plumbing: allocation through torch's caching allocator, pointers and the current stream handed to the
C ABI, and ``torch.autograd.Function`` wrappers whose backward calls the HIP backward kernels.
"""
import ctypes as C
import contextlib

import os
import torch

from . import _lib
from ._lib import ConvDesc, check, lib, lib_for, ptr, stream_ptr

ACT_NONE, ACT_LRELU, ACT_TANH, ACT_RELU = 0, 1, 2, 3
HALF_DTYPES = (torch.bfloat16, torch.float16)      # the 16-bit storage types (one library each)
IN_EPS = 1e-5                      # nn.InstanceNorm2d default, models/networks.py:22

# (float16: the fp16 build of the library, where the ABI's 16-bit dtype code names IEEE half: _lib.lib_for)
_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.BF16}

# bumped whenever master weights change behind torch's back (fused Adam, load_state_dict...)
_WEIGHT_EPOCH = [0]
# when set, conv blocks built in this scope skip their weight gradient (result-identical elision of the D
# weight gradients that train.py:176 zeroes before they are ever used)
_SKIP_WGRAD = [False]


def bump_weight_epoch():
    _WEIGHT_EPOCH[0] += 1


# Small zero-initialised buffers (InstanceNorm statistics, scalar loss accumulators): one arena per device that the
# model zeroes ONCE at the start of a step (`begin_step`) and hands out in slices, instead of one fill launch per
# buffer (~60 per step).  Slices stay valid until the next begin_step(), i.e. through that step's backward.
_ARENA = {}


def _dev_key(device):
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return str(device)


def begin_step(device, nbytes=32 << 20):
    device = _dev_key(device)
    a = _ARENA.get(device)
    if a is None or a["buf"].numel() < nbytes:
        a = {"buf": torch.empty(nbytes, dtype=torch.uint8, device=device), "off": 0, "live": False}
        _ARENA[device] = a
    a["buf"].zero_()
    lib()
    for l_ in _lib.loaded():                                       # tickets of the fixed-order reductions: re-armed per step
        check(l_.p2phd_reduction_reset(stream_ptr()), "reduction_reset")
    a["off"], a["live"] = 0, True
    a["gen"] = a.get("gen", 0) + 1


def end_arena(device):
    """Stop handing out arena slices (buffers created afterwards are zeroed individually again)."""
    a = _ARENA.get(_dev_key(device))
    if a is not None:
        a["live"] = False


def zeros(shape, device):
    """float32 zeros: a slice of the step arena when one is live and has room, else torch.zeros."""
    n = 1
    for d in shape:
        n *= int(d)
    a = _ARENA.get(str(device)) if _ARENA else None
    nbytes = (n * 4 + 15) & ~15
    if a is None or not a["live"] or a["off"] + nbytes > a["buf"].numel():
        return torch.zeros(shape, dtype=torch.float32, device=device)
    t = a["buf"][a["off"]:a["off"] + n * 4].view(torch.float32).view(shape)
    a["off"] += nbytes
    return t


def arena_generation(device):
    """Generation of the device's step arena (0 when there is none): a slice taken in generation g is recycled by the
    next begin_step(), so a backward that still needs it must run before that."""
    a = _ARENA.get(str(device))
    return a["gen"] if a is not None and a["live"] else 0


def _check_arena(ctx, device):
    g = getattr(ctx, "arena_gen", 0)
    if g:
        a = _ARENA.get(str(device))
        if a is None or a.get("gen", 0) != g:
            raise _lib.P2PHDError("the InstanceNorm statistics of this forward pass were recycled: a new training-step forward "
                                  "started before this backward ran (run backward before the next forward)")


# debugging hook: when a list, every conv block's backward appends (spec, g, dy, gx) clones (tools only)
_BWD_TRACE = [None]
# debugging hook: when a list, every normalised conv block's forward appends (spec, statistics clone, pixels per plane)
_STATS_TRACE = [None]
# number of forward convs that ran on the fp8 entry point (bench.py reports it with --fp8)
_FP8_CALLS = [0]
# number of InstanceNorm backward passes that took the sums from the consumer's input-gradient kernel
_BSUM_CALLS = [0]


# parameters whose weight gradient is not wanted by the backward pass that is running right now (a retained graph
# walked once per loss: Pix2PixHDModel.train_step keeps D's weights out of the generator-loss pass)
_BWD_SKIP_WGRAD_IDS = set()


@contextlib.contextmanager
def backward_without_weight_grads(params):
    ids = {id(p) for p in params}
    _BWD_SKIP_WGRAD_IDS.update(ids)
    try:
        yield
    finally:
        _BWD_SKIP_WGRAD_IDS.difference_update(ids)


# conv layers whose INPUT gradient is not wanted by the backward pass running right now (the first layer of each
# discriminator scale during the D-loss pass: its input is the generator output, which that pass must not touch)
_BWD_SKIP_DGRAD_SPECS = set()


@contextlib.contextmanager
def backward_without_input_grads(specs):
    ids = {id(sp) for sp in specs}
    _BWD_SKIP_DGRAD_SPECS.update(ids)
    try:
        yield
    finally:
        _BWD_SKIP_DGRAD_SPECS.difference_update(ids)


# Backward over a SAMPLE RANGE of a batch (Pix2PixHDModel: D(real) and D(fake) run as ONE batch of 2B samples -- real
# half first -- and the generator-loss pass only has a gradient on the fake half).  While set, every backward Function of
# this module whose batch size is `n_full` processes samples [lo, hi) only: it launches its kernels on the contiguous
# sub-batch views (NHWC is sample-major) and leaves the rest of the gradient tensor it returns UNWRITTEN -- the next
# Function up the chain is restricted the same way, and the chain ends in ToPhysicalPair.backward, which reads the fake half.
_BWD_RANGE = {"n_full": None, "lo": 0, "hi": 0}


@contextlib.contextmanager
def backward_on_samples(n_full, lo, hi):
    prev = dict(_BWD_RANGE)
    _BWD_RANGE.update(n_full=int(n_full), lo=int(lo), hi=int(hi))
    try:
        yield
    finally:
        _BWD_RANGE.update(prev)


def _bwd_range(n, paired=True):
    """The sample range a backward Function of batch size n is restricted to, or None.  `paired`: the Function was tagged
    in its forward as part of the stacked (real | fake) discriminator pass (its input descends from ToPhysicalPair); an
    untagged Function that merely has the same batch size is never truncated."""
    if paired and _BWD_RANGE["n_full"] is not None and _BWD_RANGE["n_full"] == int(n):
        return _BWD_RANGE["lo"], _BWD_RANGE["hi"]
    return None


def _zero_unused(t, rng):
    """The samples outside `rng` of a sample-range backward's result are never read by the Functions of the paired pass (they
    are restricted to the same range; ToPhysicalPair.backward reads the fake half only) and are left unwritten: 0.5 GB of
    memsets per step would buy nothing.  Anything that LOOKS at whole gradients -- anomaly mode, or a hook / retain_grad a
    user announces with P2PHD_ZERO_UNUSED=1 -- gets zeros there instead of stale memory."""
    if torch.is_anomaly_enabled() or os.environ.get("P2PHD_ZERO_UNUSED", "0") == "1":
        t[:rng[0]].zero_()
        t[rng[1]:].zero_()


def _is_pair(t):
    return bool(getattr(t, "_p2phd_pair", False))


def _tag_pair(out, src):
    if _is_pair(src):
        out._p2phd_pair = True
    return out


def _direct_grad(p):
    """True when `p.grad` is FlatAdam's view of its flat gradient buffer (optim.py marks the parameter)."""
    g = p.grad
    return (getattr(p, "_p2phd_direct_grad", False) and g is not None and g.dtype == torch.float32
            and (g.is_contiguous() or w_layout(g) == 1) and g.shape == p.shape)


@contextlib.contextmanager
def no_weight_grad():
    prev = _SKIP_WGRAD[0]
    _SKIP_WGRAD[0] = True
    try:
        yield
    finally:
        _SKIP_WGRAD[0] = prev


def cpitch(c):
    return (c + 7) & ~7


def w_layout(t):
    """p2phd_conv_desc::w_layout of a conv weight (or weight-gradient) tensor of logical shape [K, C, R, S]: 0 = PyTorch
    (contiguous), 1 = K-major -- storage order [K][R][S][C], what optim.FlatAdam gives the big stride-1 layers
    (p2phd_conv_kmajor_ok) so that the packed forward row IS the master row."""
    if t is None or t.dim() != 4:
        return 0
    K, Cc, R, S = t.shape
    if t.stride() == (R * S * Cc, 1, S * Cc, Cc) and not (R * S == 1 and t.is_contiguous()) and Cc > 1:
        return 1
    return 0


def kmajor_eligible(cin, cout, k, stride, transposed):
    """Shape rule of p2phd_conv_kmajor_ok, geometry-free (the library checks it again on every call)."""
    return (not transposed and stride == 1 and cin % 8 == 0 and cin >= 64 and cout >= 64 and k * k <= 16
            and (k * k * cin) % 64 == 0 and os.environ.get("P2PHD_KMAJOR", "1") != "0")


# Guard-band mode (tests / tools only; GPU AddressSanitizer is not available on this pool): every buffer this module
# hands to a kernel as an OUTPUT is carved out of a larger allocation whose 4 KiB margins are filled with a sentinel;
# `check_guards()` verifies that no kernel wrote outside its tensor.
_GUARD = {"on": False, "live": []}
_GUARD_BYTES = 4096
_SENTINEL = 0xA5


def empty(shape, dtype, device):
    if not _GUARD["on"]:
        return torch.empty(shape, dtype=dtype, device=device)
    n = 1
    for d in shape:
        n *= int(d)
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    pad = (nbytes + 255) & ~255
    raw = torch.full((pad + 2 * _GUARD_BYTES,), _SENTINEL, dtype=torch.uint8, device=device)
    t = raw[_GUARD_BYTES:_GUARD_BYTES + nbytes].view(dtype).view(shape)
    _GUARD["live"].append((raw, nbytes, tuple(int(d) for d in shape), str(dtype)))
    return t


def empty_like(t):
    return empty(tuple(t.shape), t.dtype, t.device)


def check_guards(clear=True):
    """Raise if any guard band was written; returns the number of buffers checked."""
    torch.cuda.synchronize()
    bad = []
    for raw, nbytes, shape, dt in _GUARD["live"]:
        lo = raw[:_GUARD_BYTES]
        hi = raw[_GUARD_BYTES + nbytes:]
        for name, band in (("below", lo), ("above", hi)):
            hit = (band != _SENTINEL).nonzero()
            if hit.numel():
                bad.append((shape, dt, name, int(hit.numel()), int(hit[0]), int(hit[-1])))
    n = len(_GUARD["live"])
    if clear:
        _GUARD["live"] = []
    if bad:
        raise _lib.P2PHDError(f"out-of-bounds device writes next to {len(bad)} buffer(s): {bad[:6]}")
    return n


def dt_code(dtype):
    try:
        return _DT[dtype]
    except KeyError:
        raise _lib.P2PHDError(f"compute dtype must be torch.float32, torch.bfloat16 or torch.float16, got {dtype}")


_WS = {}


def workspace(nbytes, device, tag=""):
    """Grow-only scratch buffer per device (and per `tag`: the side stream of the weight gradients has its own); kernels
    on one stream run in order, so consecutive ops share it."""
    if _GUARD["on"]:
        return empty((max(int(nbytes), 256),), torch.uint8, device)
    key = str(device) + tag
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = t
    return t


_BSUM_ALWAYS = os.environ.get("P2PHD_BSUM_ALWAYS", "0") == "1"     # A/B: fuse wherever possible, also where it measured slower


def phys(t, what="activation"):
    """Validate a physical NHWC tensor."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 4 and t.is_contiguous() and t.shape[-1] % 8 == 0
            and t.dtype in _DT):
        raise _lib.P2PHDError(f"{what}: expected a contiguous GPU NHWC tensor with channel pitch % 8 == 0 in f32/bf16, "
                              f"got {type(t).__name__} {getattr(t, 'shape', None)} {getattr(t, 'dtype', None)}")
    return t


# ------------------------------------------------------------------------------------------
# layout
# ------------------------------------------------------------------------------------------

def cat_to_physical(xs, dtype, out=None):
    """torch.cat(xs, dim=1) of f32 NCHW tensors -> NHWC physical, pad channels written as zeros by the same launch
    (`out`: a [N,H,W,Cp] tensor -- or batch slice of one -- to fill)."""
    xs = [x.contiguous().float() for x in xs]
    N, _, H, W = xs[0].shape
    chans = [int(x.shape[1]) for x in xs]
    if out is None:
        out = empty((N, H, W, cpitch(sum(chans))), dtype, xs[0].device)
    srcs = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
    cc = (C.c_int32 * len(xs))(*chans)
    check(lib_for(out.dtype).p2phd_nchw_cat_to_nhwc(dt_code(out.dtype), srcs, cc, len(xs), ptr(out), N, H * W, out.shape[-1], stream_ptr()),
          "nchw_cat_to_nhwc")
    return out


def to_physical(x_nchw, dtype):
    """f32 NCHW -> NHWC physical."""
    return cat_to_physical([x_nchw], dtype)


def from_physical(x_phys, channels, ch_off=0):
    N, H, W, Cp = x_phys.shape
    out = empty((N, channels, H, W), torch.float32, x_phys.device)
    check(lib_for(x_phys.dtype).p2phd_nhwc_to_nchw(dt_code(x_phys.dtype), ptr(x_phys), ptr(out), N, channels, H * W, Cp, ch_off, stream_ptr()),
          "nhwc_to_nchw")
    return out


class ToPhysical(torch.autograd.Function):
    """NCHW f32 (one or several tensors concatenated along channels) -> physical NHWC."""

    @staticmethod
    def forward(ctx, dtype, *xs):
        ctx.chans = [int(x.shape[1]) for x in xs]
        return cat_to_physical(xs, dtype)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        grads, off = [], 0
        for i, c in enumerate(ctx.chans):
            grads.append(from_physical(g, c, off) if ctx.needs_input_grad[1 + i] else None)
            off += c
        return (None, *grads)


class ToPhysicalPair(torch.autograd.Function):
    """Two channel-concatenated NCHW inputs stacked along the batch: out[:B] = cat(a, b_real), out[B:] = cat(a, b_fake)
    (the discriminator's input for D(real) and D(fake) as one batch of 2B).  Only b_fake gets a gradient."""

    @staticmethod
    def forward(ctx, dtype, a, b_real, b_fake):
        B, ca, H, W = a.shape
        cb = int(b_real.shape[1])
        out = empty((2 * B, H, W, cpitch(ca + cb)), dtype, a.device)
        cat_to_physical((a, b_real), dtype, out=out[:B])
        cat_to_physical((a, b_fake), dtype, out=out[B:])
        ctx.meta = (B, ca, cb)
        return out

    @staticmethod
    def backward(ctx, g):
        B, ca, cb = ctx.meta
        g = g.contiguous()
        if ctx.needs_input_grad[3] and _bwd_range(2 * B) != (B, 2 * B):
            # the Functions below wrote the fake half only if they ran inside backward_on_samples(2B, B, 2B); outside it
            # they wrote everything, which is fine too -- but a DIFFERENT range would have left the fake half unwritten
            if _BWD_RANGE["n_full"] == 2 * B:
                raise _lib.P2PHDError("ToPhysicalPair.backward: the sample-range backward does not cover the fake half")
        return None, None, None, (from_physical(g[B:], cb, ca) if ctx.needs_input_grad[3] else None)


def to_physical_pair(dtype, a, b_real, b_fake):
    """ToPhysicalPair, with the result tagged so that the Functions consuming it (and their outputs, transitively) take part
    in a sample-range backward (backward_on_samples)."""
    out = ToPhysicalPair.apply(dtype, a, b_real, b_fake)
    out._p2phd_pair = True
    return out


class FromPhysical(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_phys, channels):
        _no_lazy(x_phys, "FromPhysical")
        _note_consumer(x_phys)
        ctx.meta = (x_phys.dtype, x_phys.shape[-1])
        return from_physical(x_phys, channels)

    @staticmethod
    def backward(ctx, g):
        dtype, Cp = ctx.meta
        N, Cc, H, W = g.shape
        return cat_to_physical([g], dtype, out=empty((N, H, W, Cp), dtype, g.device)), None


# ------------------------------------------------------------------------------------------
# convolution block: [ReflectionPad] -> Conv2d/ConvTranspose2d -> [InstanceNorm] -> [act] -> [+ residual]
# ------------------------------------------------------------------------------------------

class ConvSpec:
    """Static description of one reference layer; builds the C descriptor for a given input size."""

    def __init__(self, cin, cout, k, stride=1, pad=0, pad_mode=0, transposed=False, opad=0, norm=False, act=ACT_NONE):
        self.cin, self.cout, self.k = cin, cout, k
        self.stride, self.pad, self.pad_mode = stride, pad, pad_mode
        self.transposed, self.opad = transposed, opad
        self.norm, self.act = norm, act
        self._packed = {}
        # fp8 forward (networks.enable_fp8): `fp8` = run this layer's forward on e4m3 operands when the input carries an
        # e4m3 twin; `emit_q8` = the InstanceNorm pass of this layer also writes the e4m3 twin of its output
        self.fp8 = False
        self.emit_q8 = False
        self._q8_out = None

    def fp8_ok(self, N, H, W):
        d = self.desc(N, H, W, torch.bfloat16)
        return bool(lib_for(d.torch_dtype).p2phd_conv_fp8_eligible(C.byref(d)))

    def packed_fp8(self, weight, d):
        """e4m3 weights + per-layer scale (found on the device); re-quantised when the master copy changed."""
        key = ("fp8", d.torch_dtype)
        stamp = (weight._version, weight.data_ptr(), _WEIGHT_EPOCH[0])
        hit = self._packed.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        nbytes = lib_for(d.torch_dtype).p2phd_conv_fp8_packed_bytes(C.byref(d))
        buf = hit[1] if hit is not None and hit[1].numel() == nbytes else empty((nbytes,), torch.uint8, weight.device)
        w = _master_weight(weight, d)
        check(lib_for(d.torch_dtype).p2phd_conv_fp8_pack_weights(C.byref(d), ptr(w), ptr(buf), stream_ptr()), "conv_fp8_pack_weights")
        self._packed[key] = (stamp, buf)
        return buf

    def desc(self, N, H, W, dtype, layout=0):
        d = ConvDesc(N, self.cin, H, W, self.cout, self.k, self.k, self.stride, self.pad, self.pad_mode,
                     int(self.transposed), self.opad, dt_code(dtype), int(layout))
        d.torch_dtype = dtype                                      # (which library serves it: _lib.lib_for)
        return d

    def out_size(self, d):
        ho, wo = C.c_int32(), C.c_int32()
        check(lib_for(d.torch_dtype).p2phd_conv_out_size(C.byref(d), C.byref(ho), C.byref(wo)), "conv_out_size")
        return ho.value, wo.value

    def packed(self, weight, which, d):
        """Packed weights for forward (0) / input gradient (1); re-packed when the master copy changed.
        The dgrad pack of a stride-1 reflect conv depends on H, W only through the descriptor checks."""
        # the layout variant is part of the key: a tap-skipping launch (chosen from N, H, W and an option) reads another tap
        # order than the plain one, so a pack made for the B = 32 training batch must not serve a smaller eval batch
        layout = lib_for(d.torch_dtype).p2phd_conv_pack_layout(C.byref(d), which)
        if layout < 0:
            check(-1, "conv_pack_layout")
        key = (which, d.torch_dtype, layout)
        stamp = (weight._version, weight.data_ptr(), _WEIGHT_EPOCH[0])
        hit = self._packed.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        nbytes = lib_for(d.torch_dtype).p2phd_conv_packed_bytes(C.byref(d), which)
        buf = hit[1] if hit is not None and hit[1].numel() == nbytes else empty((nbytes,), torch.uint8, weight.device)
        w = _master_weight(weight, d)
        check(lib_for(d.torch_dtype).p2phd_conv_pack_weights(C.byref(d), which, ptr(w), ptr(buf), stream_ptr()), "conv_pack_weights")
        self._packed[key] = (stamp, buf)
        return buf


def _master_weight(weight, d):
    """The f32 master weights in the layout d.w_layout names (a K-major parameter is handed over as it lies in memory)."""
    w = weight.detach()
    if d.w_layout == 1:
        if w.dtype != torch.float32 or w_layout(w) != 1:
            raise _lib.P2PHDError("conv: descriptor says K-major master weights but the tensor is not a float32 [K][R][S][C] view")
        return w
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    return w


# Fused first pass of the producer's InstanceNorm backward (p2phd_conv_dgrad_bsum): when the tensor a conv block reads is the
# output of an InstanceNorm block and NOBODY ELSE consumes it, the block's input-gradient kernel also leaves the
# per-(n, c) sums that block's backward needs, and it then runs the apply pass only -- one pass over (g, y) less on the big
# planes.  "Nobody else" cannot be inferred: autograd sums the contributions of several consumers into one tensor, in
# place and without a version bump when it owns the buffer, so neither pointer nor version of the gradient proves it.
# The CALLER states it: conv_block(..., exclusive=True) is passed by networks._run for a step whose input is the
# previous step's output and leaves the chain nowhere else (the generator's sequential chains; never the
# discriminator's exposed features).  Pointer, version and a consumer count of the Functions of this module are kept as
# secondary guards.  P2PHD_BSUM=0 switches the fusion off (A/B runs).
def _bsum_enabled():
    return os.environ.get("P2PHD_BSUM", "1") != "0"


def _note_consumer(t):
    gf = getattr(t, "grad_fn", None)
    if gf is not None and hasattr(gf, "_p2phd_consumers"):
        gf._p2phd_consumers += 1
    return gf


# Lazily normalised activations (round 4).  A conv block asked to `defer` returns its RAW conv output tagged with a LazyNorm
# (statistics, activation, channels): the consumer -- a layer whose forward and weight gradient stage their input through
# registers (csrc/march.hip; p2phd_conv_lazy_ok) -- applies (y - mean) * rstd + activation on load, and the InstanceNorm
# forward pass over that plane (a read and a write of the whole tensor) never runs.  Same values, bit for bit.  Any other
# consumer materialises the tensor first (`materialise`): deferring is always safe, only sometimes useless, so
# networks._run asks for it exactly where the next step is such a layer.
class LazyNorm:
    __slots__ = ("stats", "act", "channels", "gen")

    def __init__(self, stats, act, channels, gen):
        self.stats, self.act, self.channels, self.gen = stats, act, channels, gen


_LAST_LAZY = [None]        # hand-over from ConvBlockFn.forward (which cannot tag its own output) to conv_block


def lazy_static_ok(spec):
    """Shape rule of p2phd_conv_lazy_ok without the geometry (the library checks the rest on every call)."""
    return (spec.k == 3 and spec.stride == 2 and spec.pad == 1 and spec.pad_mode == 0 and not spec.fp8 and
            ((not spec.transposed and spec.cin == 48 and spec.cout == 96) or
             (spec.transposed and spec.opad == 1 and spec.cin == 96 and spec.cout == 48)))


def materialise(x):
    """The normalised + activated tensor a LazyNorm-tagged raw output stands for (p2phd_instnorm_act_fwd)."""
    lz = getattr(x, "_p2phd_lazy", None)
    if lz is None:
        return x
    N, H, W, _ = x.shape
    out = empty_like(x)
    check(lib_for(x.dtype).p2phd_instnorm_act_fwd(dt_code(x.dtype), ptr(x), ptr(lz.stats), None, ptr(out), N, H * W, lz.channels, IN_EPS,
                                       lz.act, stream_ptr()), "instnorm_act_fwd")
    return out


class ConvBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, spec, link=None, exclusive=False, defer=False):
        src = _note_consumer(x)
        if not exclusive:
            src = None
        if residual is not None:
            _note_consumer(residual)
        x = phys(x, "conv input")
        N, H, W, Cp_in = x.shape
        if Cp_in != cpitch(spec.cin):
            raise _lib.P2PHDError(f"conv: input has channel pitch {Cp_in}, layer expects {cpitch(spec.cin)} ({spec.cin} channels)")
        d = spec.desc(N, H, W, x.dtype, w_layout(weight))
        Ho, Wo = spec.out_size(d)
        L = lib_for(x.dtype)
        Cp_out = cpitch(spec.cout)
        lazy_in = getattr(x, "_p2phd_lazy", None)
        if lazy_in is not None and not (x.dtype in HALF_DTYPES and (spec.norm or spec.act == ACT_NONE) and L.p2phd_conv_lazy_ok(C.byref(d))):
            x, lazy_in = materialise(x), None                      # this layer cannot normalise on load after all
        x8 = getattr(x, "_p2phd_q8", None) if (spec.fp8 and x.dtype == torch.bfloat16) else None
        wp = spec.packed_fp8(weight, d) if x8 is not None else spec.packed(weight, 0, d)
        b = None if bias is None else bias.detach().float().contiguous()
        y = empty((N, Ho, Wo, Cp_out), x.dtype, x.device)
        stats = zeros((N, Cp_out, 2), x.device) if spec.norm else None
        fused_act = ACT_NONE if spec.norm else spec.act
        wsb = L.p2phd_conv_fwd_workspace_bytes(C.byref(d))
        ws = workspace(wsb, x.device) if wsb else None
        if x8 is not None:
            _FP8_CALLS[0] += 1
            check(L.p2phd_conv_fwd_fp8(C.byref(d), ptr(x8), ptr(wp), ptr(b), fused_act, ptr(y), ptr(stats), ptr(ws), stream_ptr()), "conv_fwd_fp8")
        elif lazy_in is not None:
            check(L.p2phd_conv_fwd_lazy(C.byref(d), ptr(x), ptr(lazy_in.stats), lazy_in.act, IN_EPS, ptr(wp), ptr(b), ptr(y), ptr(stats),
                                        ptr(ws), stream_ptr()), "conv_fwd_lazy")
        else:
            check(L.p2phd_conv_fwd(C.byref(d), ptr(x), ptr(wp), ptr(b), fused_act, ptr(y), ptr(stats), ptr(ws), stream_ptr()), "conv_fwd")
        _LAST_LAZY[0] = None
        if spec.norm and defer and residual is None and not spec.emit_q8:
            # the consumer normalises on load: hand out the raw output, tagged by conv_block
            out = y
            _LAST_LAZY[0] = LazyNorm(stats, spec.act, spec.cout, arena_generation(x.device))
        elif spec.norm:
            if _STATS_TRACE[0] is not None:
                _STATS_TRACE[0].append((spec, stats.detach().clone(), Ho * Wo))
            res = None if residual is None else phys(residual, "residual")
            out = empty_like(y)
            if spec.emit_q8 and y.dtype == torch.bfloat16:
                out8 = empty(tuple(y.shape), torch.uint8, y.device)
                check(L.p2phd_instnorm_act_fwd_q8(d.dtype, ptr(y), ptr(stats), ptr(res), ptr(out), ptr(out8), N, Ho * Wo, spec.cout,
                                                  IN_EPS, spec.act, stream_ptr()), "instnorm_act_fwd_q8")
                spec._q8_out = out8
            else:
                check(L.p2phd_instnorm_act_fwd(d.dtype, ptr(y), ptr(stats), ptr(res), ptr(out), N, Ho * Wo, spec.cout, IN_EPS,
                                               spec.act, stream_ptr()), "instnorm_act_fwd")
        else:
            if residual is not None:
                raise _lib.P2PHDError("residual add is only fused behind InstanceNorm")
            out = y
        ctx.spec, ctx.d = spec, d
        ctx.in_link = getattr(x, "_p2phd_pool_link", None)          # (armed by conv_block: grad mode is off in here)
        ctx.pair = _is_pair(x)
        ctx.skip_wgrad = _SKIP_WGRAD[0]
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.x, ctx.y, ctx.stats, ctx.weight, ctx.bias = x, y, stats, weight, bias
        ctx.x_lazy = lazy_in                                       # x is RAW: the weight gradient normalises it on load too
        ctx.link = link
        ctx.arena_gen = arena_generation(x.device) if spec.norm else 0
        ctx._p2phd_consumers = 0                                   # forward calls that read `out` (see _note_consumer)
        ctx._bs = None                                             # (bstats, data_ptr, version) left by the consumer's dgrad
        ctx._parked = None                                         # gradient of `out` parked by a loss (LossFn, park=True)
        ctx._dy_done = None                                        # (data_ptr, version) of a gradient that already carries act' 
        ctx.src = src if (src is not None and hasattr(src, "_p2phd_consumers") and getattr(src, "spec", None) is not None) else None
        return out

    @staticmethod
    def backward(ctx, g):
        spec, d = ctx.spec, ctx.d
        x, y, stats, weight = ctx.x, ctx.y, ctx.stats, ctx.weight
        L = lib_for(y.dtype)
        _check_arena(ctx, y.device)
        g = g.contiguous()
        if g.dtype != y.dtype:
            g = g.to(y.dtype)
        rng = _bwd_range(d.N, ctx.pair)
        x_full = x
        if rng is not None:
            # sample-range backward (see backward_on_samples): everything below runs on the sub-batch views
            lo, hi = rng
            x, y, g = x[lo:hi], y[lo:hi], g[lo:hi]
            stats = None if stats is None else stats[lo:hi]
            d = spec.desc(hi - lo, x.shape[1], x.shape[2], x.dtype, d.w_layout)
        parked, ctx._parked = ctx._parked, None
        dy_done, ctx._dy_done = ctx._dy_done, None                 # (consumed once: a marker must not outlive its backward pass)
        # the consumer's input-gradient kernel (p2phd_conv_dgrad_act) may have applied this block's activation derivative
        # to g already; that is only usable when g IS the tensor it wrote (same storage, untouched since)
        carries_act = False
        if dy_done is not None:
            if not (dy_done == (g.data_ptr(), g._version) and g.shape == y.shape and not spec.norm and spec.act != ACT_NONE):
                raise _lib.P2PHDError("conv backward: the consumer's input-gradient kernel already applied this block's activation "
                                      "derivative, but the gradient that arrived is not the tensor it wrote (a second consumer of "
                                      "an `exclusive` chain?); rerun with P2PHD_BSUM=0")
            carries_act = True
        if parked is not None:
            # a loss parked its gradient of this block's output for the consumer's input-gradient kernel to add, and that
            # kernel did not take it (it ran first, or does not exist in this backward pass): add it here
            parked = parked.to(g.dtype)
            if carries_act:
                # g is already dL/d(pre-activation) of the conv path; the parked part is still dL/d(output): give it act' alone
                pk = empty_like(y)
                check(L.p2phd_act_bwd(d.dtype, ptr(parked.contiguous()), ptr(y), ptr(pk), y.numel(), spec.act, stream_ptr()), "act_bwd")
                parked = pk
            g = g + parked
            ctx._bs = None
        N, Ho, Wo, Cp_out = y.shape
        need_w = ((ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])) and not ctx.skip_wgrad
                  and id(weight) not in _BWD_SKIP_WGRAD_IDS)
        # Parameters owned by FlatAdam carry their gradient as a view of its flat buffer: the kernels then add into it
        # directly (p2phd_*_acc) and autograd gets None, which saves a temporary and a `grad += new` launch per parameter.
        direct = need_w and _direct_grad(weight) and (not ctx.has_bias or _direct_grad(ctx.bias))
        # Lazily zeroed gradients (FlatAdam.zero_grad(lazy=True)): a parameter marked fresh holds stale values; its first
        # gradient of the step OVERWRITES them (the non-accumulating entry points), later ones add -- no 410 MB memset per step
        # (biases are zeroed eagerly by that zero_grad -- one launch for all of them -- and always accumulate)
        fresh = need_w and getattr(weight, "_p2phd_fresh", False)
        if fresh and not direct and weight.grad is not None:
            weight.grad.zero_()
        acc = direct and not fresh                                 # weight gradient: add (else overwrite)
        acc_b = direct                                             # bias gradient: always add into the flat buffer
        if need_w:
            weight._p2phd_fresh = False
        gb = None
        if need_w and ctx.has_bias:
            gb = ctx.bias.grad if direct else empty((spec.cout,), torch.float32, y.device)
        gb_done = False
        rx = None                                                  # reflection extras behind dy (p2phd_conv_dgrad_rx)
        if spec.norm:
            bs, ctx._bs = ctx._bs, None
            want_gx = ctx.needs_input_grad[0] and id(spec) not in _BWD_SKIP_DGRAD_SPECS
            n_rx = L.p2phd_conv_reflect_extras_elems(C.byref(d)) if (spec.pad_mode == 1 and want_gx and bs is None) else 0
            if n_rx:
                # residual trunk: the kernel that writes dy appends the pair-sum rows / columns its input-gradient GEMM reads
                buf = empty((y.numel() + n_rx,), y.dtype, y.device)
                dy, rx = buf[:y.numel()].view(y.shape), buf[y.numel():]
                check(L.p2phd_instnorm_act_bwd_rx(d.dtype, ptr(g), ptr(y), ptr(stats), ptr(dy), ptr(gb), 1 if acc_b else 0, N, Ho, Wo,
                                                  spec.cout, IN_EPS, spec.act, ptr(rx), stream_ptr()), "instnorm_act_bwd_rx")
            elif bs is not None and g.data_ptr() == bs[1] and g._version == bs[2] and g.shape == y.shape:
                # the consumer's input-gradient kernel already summed (g', g' * yhat): apply pass only
                dy = empty_like(y)
                check(L.p2phd_instnorm_act_bwd_apply(d.dtype, ptr(g), ptr(y), ptr(stats), ptr(bs[0]), ptr(dy), ptr(gb), 1 if acc_b else 0,
                                                     N, Ho * Wo, spec.cout, IN_EPS, spec.act, stream_ptr()), "instnorm_act_bwd_apply")
                _BSUM_CALLS[0] += 1
            else:
                dy = empty_like(y)
                bstats = empty((N, Cp_out, 2), torch.float32, y.device)
                # the bias gradient (column sums of dy) rides on the apply pass
                bwd = L.p2phd_instnorm_act_bwd_acc if acc_b else L.p2phd_instnorm_act_bwd
                check(bwd(d.dtype, ptr(g), ptr(y), ptr(stats), ptr(bstats), ptr(dy), ptr(gb), N, Ho * Wo, spec.cout, IN_EPS, spec.act,
                          stream_ptr()), "instnorm_act_bwd")
            gb_done = gb is not None
        elif carries_act:
            dy = g                                                 # the consumer's input-gradient kernel applied act' already
            _BSUM_CALLS[0] += 1
        elif spec.act != ACT_NONE:
            dy = empty_like(y)
            if gb is not None:                                     # bias gradient rides on the activation-backward pass
                check(L.p2phd_act_bwd_db(d.dtype, ptr(g), ptr(y), ptr(dy), N * Ho * Wo, spec.cout, spec.act, ptr(gb),
                                         1 if acc_b else 0, stream_ptr()), "act_bwd_db")
                gb_done = True
            else:
                check(L.p2phd_act_bwd(d.dtype, ptr(g), ptr(y), ptr(dy), y.numel(), spec.act, stream_ptr()), "act_bwd")
        else:
            dy = g
        gx = gw = None
        if need_w:
            gw = weight.grad if direct else empty(tuple(weight.shape), torch.float32, y.device)
            wgrad = L.p2phd_conv_wgrad_acc if acc else L.p2phd_conv_wgrad
            dwd = d if w_layout(gw) == d.w_layout else spec.desc(d.N, d.H, d.W, y.dtype, w_layout(gw))   # layout of what is WRITTEN
            ws = workspace(L.p2phd_conv_wgrad_workspace_bytes(C.byref(dwd)), y.device)
            if ctx.x_lazy is not None:
                lz = ctx.x_lazy
                if lz.gen and lz.gen != (_ARENA.get(str(y.device)) or {}).get("gen", 0):
                    raise _lib.P2PHDError("the InstanceNorm statistics of this layer's lazily normalised input were recycled: a new "
                                          "training-step forward started before this backward ran")
                if rng is not None:
                    raise _lib.P2PHDError("sample-range backward through a lazily normalised input is not supported")
                check(L.p2phd_conv_wgrad_lazy(C.byref(dwd), ptr(x), ptr(lz.stats), lz.act, IN_EPS, ptr(dy), ptr(gw),
                                              None if gb_done else ptr(gb), 1 if acc else 0, ptr(ws), stream_ptr()), "conv_wgrad_lazy")
            else:
                check(wgrad(C.byref(dwd), ptr(x), ptr(dy), ptr(gw), None if gb_done else ptr(gb), ptr(ws), stream_ptr()), "conv_wgrad")
            if direct:
                gw = gb = None
        gx_full = None
        if ctx.needs_input_grad[0] and id(spec) not in _BWD_SKIP_DGRAD_SPECS:
            wp = spec.packed(weight, 1, d)
            gx_full = empty_like(x_full)
            gx = gx_full if rng is None else gx_full[rng[0]:rng[1]]
            if rng is not None:
                _zero_unused(gx_full, rng)
            wsb = L.p2phd_conv_dgrad_workspace_bytes(C.byref(d))
            ws = workspace(wsb, y.device) if wsb else None
            # first conv of a residual block: the skip gradient parked by the block's second conv is added inside the
            # dgrad (epilogue / reflect fold) instead of by a separate autograd add
            addend = None
            if ctx.link is not None and ctx.link.role_of(ctx) == "a":
                addend = ctx.link.take()
            if ctx.in_link is not None:
                pk = ctx.in_link.take()                            # gradient of x through the pooling branch (PoolLink)
                if pk is not None:
                    pk = pk if rng is None else pk[rng[0]:rng[1]]
                    addend = pk if addend is None else addend + pk
            src = ctx.src
            if src is not None and src._parked is not None and src._p2phd_consumers == 1:
                # feature-matching gradient of x parked by the loss: summed inside this kernel instead of by autograd
                pk, src._parked = src._parked, None
                if pk.shape != gx.shape:
                    raise _lib.P2PHDError(f"conv backward: a parked loss gradient of shape {tuple(pk.shape)} meets an input gradient of "
                                          f"shape {tuple(gx.shape)} (a half-batch loss needs backward_on_samples around this pass)")
                addend = pk if addend is None else addend + pk
            src_y = src_stats = None
            if src is not None and src.y is not None:
                src_y = src.y if rng is None else src.y[rng[0]:rng[1]]
                if src.stats is not None:
                    src_stats = src.stats if rng is None else src.stats[rng[0]:rng[1]]
            fuse = (src is not None and src.spec.norm and src._p2phd_consumers == 1 and _bsum_enabled() and src_y is not None
                    and src_y.shape == x.shape and src_y.dtype == x.dtype and src.spec.act in (ACT_NONE, ACT_RELU, ACT_LRELU)
                    and L.p2phd_instnorm_act_bwd_two_pass(d.dtype, x.shape[0], x.shape[1] * x.shape[2], spec.cin)
                    and (L.p2phd_conv_dgrad_bsum_ok(C.byref(d)) if _BSUM_ALWAYS else L.p2phd_conv_dgrad_bsum_pays(C.byref(d))))
            if fuse:
                _check_arena(src, y.device)
                bst = empty((x.shape[0], x.shape[3], 2), torch.float32, y.device)
                wsf = workspace(max(L.p2phd_conv_dgrad_bsum_workspace_bytes(C.byref(d)), 256), y.device)
                check(L.p2phd_conv_dgrad_bsum(C.byref(d), ptr(dy), ptr(wp), ptr(addend), ptr(gx), ptr(src_y), ptr(src_stats),
                                              src.spec.act, IN_EPS, ptr(bst), ptr(wsf), stream_ptr()), "conv_dgrad_bsum")
                src._bs = (bst, gx.data_ptr(), gx._version)
            elif (src is not None and not src.spec.norm and src.spec.act in (ACT_RELU, ACT_LRELU) and src._p2phd_consumers == 1
                  and _bsum_enabled() and src_y is not None and src_y.shape == x.shape and src_y.dtype == x.dtype
                  and L.p2phd_conv_dgrad_bsum_ok(C.byref(d))):
                # producer = Conv + (Leaky)ReLU without normalisation: its activation derivative is applied to gx here
                wsf = workspace(max(L.p2phd_conv_dgrad_bsum_workspace_bytes(C.byref(d)), 256), y.device)
                check(L.p2phd_conv_dgrad_act(C.byref(d), ptr(dy), ptr(wp), ptr(addend), ptr(gx), ptr(src_y), src.spec.act, ptr(wsf),
                                             stream_ptr()), "conv_dgrad_act")
                src._dy_done = (gx.data_ptr(), gx._version)
            elif rx is not None:
                check(L.p2phd_conv_dgrad_rx(C.byref(d), ptr(dy), ptr(wp), ptr(addend), ptr(gx), stream_ptr()), "conv_dgrad_rx")
            else:
                check(L.p2phd_conv_dgrad(C.byref(d), ptr(dy), ptr(wp), ptr(addend), ptr(gx), ptr(ws), stream_ptr()), "conv_dgrad")
        if _BWD_TRACE[0] is not None:
            _BWD_TRACE[0].append((spec, g.detach().clone(), dy.detach().clone(), None if gx is None else gx.detach().clone()))
        gres = g if (ctx.has_res and ctx.needs_input_grad[3]) else None
        if gres is not None and rng is not None:
            raise _lib.P2PHDError("sample-range backward through a residual block is not supported")
        if gres is not None and ctx.link is not None and ctx.link.park(gres, ctx):
            gres = None
        return gx_full, gw, gb, gres, None, None, None, None


class SkipLink:
    """Couples the two convs of one ResnetBlock call: out = x + b(a(x)).  In backward, b runs first and parks the
    gradient of the skip path here; a (whose input is the same x) hands it to its dgrad kernel as `addend`, so the
    sum dL/dx = dgrad_a(.) + dL/dout is formed inside that kernel.  One link per forward call of the block."""

    def __init__(self):
        self.g = None
        self.armed = False            # set once conv `a` ran its forward with an input that needs a gradient

    def role_of(self, ctx):
        return "b" if ctx.has_res else "a"

    def park(self, g, ctx):
        if not (self.armed and ctx.has_res):
            return False
        self.g = g
        return True

    def take(self):
        g, self.g = self.g, None
        return g


def conv_block(x, weight, bias, spec, residual=None, link=None, exclusive=False, defer=False):
    """`exclusive`: x is the output of another conv_block and this call is its ONLY consumer (see _bsum_enabled).
    `defer`: the ONLY consumer of this block's output is a layer that normalises on load (lazy_static_ok): return the raw
    conv output tagged with its LazyNorm instead of running the InstanceNorm forward pass."""
    if link is not None and residual is None:
        link.armed = bool(x.requires_grad) and torch.is_grad_enabled()
    pool_link = getattr(x, "_p2phd_pool_link", None)
    if pool_link is not None:
        pool_link.conv_spec = spec if (x.requires_grad and torch.is_grad_enabled()) else None
    out = _tag_pair(ConvBlockFn.apply(x, weight, bias, residual, spec, link, exclusive, defer), x)
    if _LAST_LAZY[0] is not None:
        out._p2phd_lazy, _LAST_LAZY[0] = _LAST_LAZY[0], None
    if spec._q8_out is not None:                                    # e4m3 twin of this output for the next layer's fp8 forward
        out._p2phd_q8, spec._q8_out = spec._q8_out, None
    return out


# ------------------------------------------------------------------------------------------
# AvgPool2d(3, 2, 1, count_include_pad=False)
# ------------------------------------------------------------------------------------------

class PoolLink:
    """Couples the two consumers of a discriminator scale's input (networks.py:311-331: the scale's first conv and the
    AvgPool2d that feeds the next scale).  In backward the pooling branch runs first (its nodes were created later); it
    parks its input gradient here and the conv's input-gradient kernel adds it as `addend` -- the sum autograd would
    otherwise form with a pass of its own over the full-resolution 2B-sample tensor.  One link per forward call;
    attached to the shared input tensor (`t._p2phd_pool_link`)."""

    def __init__(self):
        self.g = None
        self.conv_spec = None         # set by the conv's forward when its input gradient can be wanted
        self.taken = False            # the conv's backward has run: a pooling backward that comes later returns its gradient itself

    def take(self):
        g, self.g, self.taken = self.g, None, True
        return g


def _no_lazy(t, what):
    if getattr(t, "_p2phd_lazy", None) is not None:
        raise _lib.P2PHDError(f"{what}: got a lazily normalised (raw) conv output; only the next conv block of a chain may consume it "
                              "(conv_block(..., defer=True))")
    return t


class AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, channels):
        ctx.link = getattr(x, "_p2phd_pool_link", None)
        x = phys(_no_lazy(x, "avgpool"), "avgpool input")
        N, H, W, Cp = x.shape
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y = empty((N, Ho, Wo, Cp), x.dtype, x.device)
        check(lib_for(x.dtype).p2phd_avgpool3s2_fwd(dt_code(x.dtype), ptr(x), ptr(y), N, H, W, channels, stream_ptr()), "avgpool_fwd")
        ctx.meta = (N, H, W, Cp, channels)
        ctx.pair = _is_pair(x)
        return y

    @staticmethod
    def backward(ctx, g):
        N, H, W, Cp, channels = ctx.meta
        g = g.contiguous()
        dx = empty((N, H, W, Cp), g.dtype, g.device)
        rng = _bwd_range(N, ctx.pair)
        if rng is not None:                                        # sample-range backward (backward_on_samples)
            lo, hi = rng
            _zero_unused(dx, rng)
            check(lib_for(g.dtype).p2phd_avgpool3s2_bwd(dt_code(g.dtype), ptr(g[lo:hi]), ptr(dx[lo:hi]), hi - lo, H, W, channels, stream_ptr()), "avgpool_bwd")
        else:
            check(lib_for(g.dtype).p2phd_avgpool3s2_bwd(dt_code(g.dtype), ptr(g), ptr(dx), N, H, W, channels, stream_ptr()), "avgpool_bwd")
        link = ctx.link
        if (link is not None and not link.taken and link.g is None and link.conv_spec is not None
                and id(link.conv_spec) not in _BWD_SKIP_DGRAD_SPECS):
            link.g = dx                                            # the sibling conv's input-gradient kernel adds it (PoolLink)
            return None, None
        return dx, None


def avgpool(x, channels):
    return _tag_pair(AvgPoolFn.apply(x, channels), x)


# ------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------

class LossFn(torch.autograd.Function):
    """kind 0: mean((a-target)^2) ; kind 1: mean(|a-b|) * coeff.  Returns a 0-dim f32 device tensor.
    `rows` = (n0, n1): the loss is taken over samples [n0, n1) of `a` only (b, if given, has n1 - n0 samples); the
    gradient is zero on the other samples -- or, with park, only the [n0, n1) part exists and is parked."""

    @staticmethod
    def forward(ctx, a, b, kind, target, coeff, channels, park=False, rows=None, into=None):
        gf = getattr(a, "grad_fn", None)
        ctx.park_src = gf if (park and gf is not None and hasattr(gf, "_p2phd_consumers") and hasattr(gf, "_parked")) else None
        if ctx.park_src is None:
            _note_consumer(a)
        a = phys(_no_lazy(a, "loss"), "loss input")
        av = a if rows is None else a[rows[0]:rows[1]]
        if b is not None and tuple(b.shape) != tuple(av.shape):
            raise _lib.P2PHDError(f"loss: operand shapes differ: {tuple(av.shape)} vs {tuple(b.shape)}")
        P = av.numel() // av.shape[-1]
        # `into` (LossAcc): the kernel ADDS its term to that accumulator (it always accumulates: a fresh slot is zero), and
        # the tensor returned is only this term's handle in the autograd graph (LossSum wires the gradients)
        out = zeros((), a.device) if into is None else into
        check(lib_for(a.dtype).p2phd_loss_fwd(kind, dt_code(a.dtype), ptr(av), ptr(b), float(target), P, channels, float(coeff),
                                   ptr(out), stream_ptr()), "loss_fwd")
        ctx.meta = (kind, float(target), float(coeff), channels, P, rows)
        ctx.a, ctx.b = a, b
        return out if into is None else out.detach()

    @staticmethod
    def backward(ctx, g):
        kind, target, coeff, channels, P, rows = ctx.meta
        a, b = ctx.a, ctx.b
        av = a if rows is None else a[rows[0]:rows[1]]
        parked = ctx.park_src is not None
        if rows is None or parked:
            da_full = None
            da = empty_like(av)
        else:
            da_full = torch.zeros_like(a)                          # (last-stage features: a few hundred KB)
            da = da_full[rows[0]:rows[1]]
        g = g.contiguous().float()
        check(lib_for(a.dtype).p2phd_loss_bwd(kind, dt_code(a.dtype), ptr(av), ptr(b), target, P, channels, coeff, ptr(g), ptr(da),
                                   stream_ptr()), "loss_bwd")
        if _BWD_TRACE[0] is not None:
            _BWD_TRACE[0].append((("loss", kind, tuple(av.shape), coeff), g.detach().clone(), da.detach().clone(),
                                  None if b is None else b.detach().clone()))
        if parked:
            # hand the gradient to the block that produced `a`: its exclusive consumer adds it inside its input-gradient
            # kernel (or the block itself does, if that kernel is not part of this backward pass); autograd gets nothing
            src = ctx.park_src
            src._parked = da if src._parked is None else src._parked + da
            return None, None, None, None, None, None, None, None, None
        return (da if da_full is None else da_full), None, None, None, None, None, None, None, None


class LossSum(torch.autograd.Function):
    """`total` = the accumulator the kernels of `terms` added into, as the autograd sum of those terms: no launch in either
    direction (the reference's `loss += term` chains, pix2pixHD_model.py:391-398, networks.py:100-110, cost a kernel per term
    and another per term in the backward pass).  One copy per loss instead."""

    @staticmethod
    def forward(ctx, total, *terms):
        ctx.n = len(terms)
        return total.detach().clone()          # (a copy: the accumulator is an arena slot, recycled by the next step's begin_step)

    @staticmethod
    def backward(ctx, g):
        return (None,) + (g,) * ctx.n


class LossAcc:
    """One scalar loss built from several terms (scales, feature levels): every term's kernel adds into one arena slot."""

    def __init__(self, device):
        self.slot = zeros((), device)
        self.terms = []

    def mse_const(self, a_phys, channels, target, rows=None):
        self.terms.append(LossFn.apply(a_phys, None, 0, target, 1.0, channels, False, rows, self.slot))

    def l1(self, a_phys, b_phys, channels, coeff=1.0, park=False):
        self.terms.append(LossFn.apply(a_phys, b_phys.detach(), 1, 0.0, coeff, channels, park, None, self.slot))

    def l1_halves(self, t_phys, channels, coeff=1.0, park=False):
        n = t_phys.shape[0]
        if n % 2:
            raise _lib.P2PHDError("l1_halves_loss: the batch must hold two equal halves")
        self.terms.append(LossFn.apply(t_phys, t_phys.detach()[:n // 2], 1, 0.0, coeff, channels, park, (n // 2, n), self.slot))

    def total(self):
        if not self.terms:
            return 0
        return LossSum.apply(self.slot, *self.terms)


def mse_const_loss(a_phys, channels, target, rows=None):
    return LossFn.apply(a_phys, None, 0, target, 1.0, channels, False, rows)


def l1_loss(a_phys, b_phys, channels, coeff=1.0, park=False):
    """`park`: a_phys is the output of a conv block whose only other consumer is the next conv block of an exclusive chain
    (networks._run): the loss gradient is parked on the producer and added inside that consumer's input-gradient kernel
    instead of by an autograd accumulation launch (and the InstanceNorm-backward sums can be fused there, see _bsum_enabled)."""
    return LossFn.apply(a_phys, b_phys.detach(), 1, 0.0, coeff, channels, park, None)


def l1_halves_loss(t_phys, channels, coeff=1.0, park=False):
    """mean(|t[B:] - t[:B]|) * coeff for a tensor holding two stacked batches (real half first, then fake): the
    feature-matching term when D(real) and D(fake) ran as one batch.  The gradient exists on the fake half only; with
    `park` it is parked (half-shaped) on the producer for the consumer's input-gradient kernel of a sample-range
    backward pass (backward_on_samples) to add."""
    n = t_phys.shape[0]
    if n % 2:
        raise _lib.P2PHDError("l1_halves_loss: the batch must hold two equal halves")
    return LossFn.apply(t_phys, t_phys.detach()[:n // 2], 1, 0.0, coeff, channels, park, (n // 2, n))
