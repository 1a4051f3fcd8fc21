"""Host-side logic that needs no GPU: the autograd property the fused-backward guards of _ops.py rely on, the sample-range
context of the paired discriminator batch, small bench helpers."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_autograd_accumulation_is_visible_through_pointer_and_version():
    """_ops.ConvBlockFn lets a consumer's input-gradient kernel do part of the producer's backward (InstanceNorm sums,
    activation derivative) and recognises "this gradient is exactly the tensor that kernel wrote" by (data_ptr, _version).
    The round-2 advisor feared that autograd's in-place accumulation of a SECOND consumer's gradient keeps both unchanged.
    It does not (torch 2.10): whichever contribution arrives first, the sum the producer receives either lives in another
    buffer or carries a bumped version -- so a broken `exclusive` promise is detected (fall back / raise), never silent."""
    rec = {}

    class Consumer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, g):
            gx = torch.full_like(g, 3.0)
            rec["wrote"] = (gx.data_ptr(), gx._version)
            return gx

    class Producer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x + 1

        @staticmethod
        def backward(ctx, g):
            rec["got"] = (g.data_ptr(), g._version)
            rec["value"] = float(g.flatten()[0])
            return g

    for second_first in (False, True):
        x = torch.ones(8, requires_grad=True)
        h = Producer.apply(x)
        if second_first:
            side = (h * h).sum(); o = Consumer.apply(h)
        else:
            o = Consumer.apply(h); side = (h * h).sum()
        (o.sum() + side).backward()
        assert rec["value"] == 7.0                                   # 3 (consumer) + 2 h (= 4): both contributions arrived
        assert rec["got"] != rec["wrote"], (second_first, rec)
    # ... and with the consumer alone the gradient IS the tensor it wrote
    x = torch.ones(8, requires_grad=True)
    Consumer.apply(Producer.apply(x)).sum().backward()
    assert rec["got"] == rec["wrote"]


def test_backward_sample_range_context():
    from pix2pixhdaudiosr_amd import _ops
    assert _ops._bwd_range(64) is None
    with _ops.backward_on_samples(64, 32, 64):
        assert _ops._bwd_range(64) == (32, 64)
        assert _ops._bwd_range(32) is None                           # the generator's batch is not restricted
        with _ops.backward_on_samples(8, 4, 8):
            assert _ops._bwd_range(8) == (4, 8) and _ops._bwd_range(64) is None
        assert _ops._bwd_range(64) == (32, 64)
    assert _ops._bwd_range(64) is None


def test_bench_reports_the_cpu_model():
    import bench
    name = bench.cpu_model_name()
    assert isinstance(name, str) and len(name) > 0
