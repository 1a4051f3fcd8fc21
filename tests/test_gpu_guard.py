"""Home-made sanitizer run (GPU AddressSanitizer is not available on this pool): every output buffer of the conv stack is
carved out of a larger allocation with 4 KiB sentinel margins (`_ops._GUARD`), then one whole training step of the tiny
golden model -- every conv geometry, fold path, InstanceNorm variant, loss and layout kernel, forward and both backward
passes -- must leave all margins untouched, in fp32 and bf16."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fp16", [False, True], ids=["f32", "bf16"])
def test_no_kernel_writes_outside_its_output(golden_model, fp16):
    from pix2pixhdaudiosr_amd import _ops
    from test_gpu_model import _model
    g = golden_model
    lr, hr, noise = (torch.from_numpy(g[k]).cuda() for k in ("lr", "hr", "mask_noise"))
    m = _model(g, fp16=fp16)
    _ops._GUARD["on"] = True
    _ops._GUARD["live"] = []
    try:
        m._phase_a_forward(lr, hr, noise)
        n_fwd = _ops.check_guards()
        for run, _ in m._g_stages():
            run()
        n_g = _ops.check_guards()
        m._phase_b()
        n_d = _ops.check_guards()
    finally:
        _ops._GUARD["on"] = False
        _ops._GUARD["live"] = []
    assert n_fwd > 50 and n_g > 50 and n_d > 20, (n_fwd, n_g, n_d)


def test_guard_catches_a_stray_write():
    from pix2pixhdaudiosr_amd import _ops, _lib
    _ops._GUARD["on"] = True
    _ops._GUARD["live"] = []
    try:
        t = _ops.empty((4, 8), torch.float32, "cuda")
        raw = _ops._GUARD["live"][-1][0]
        raw[_ops._GUARD_BYTES + t.numel() * 4 + 3] = 0                 # one byte past the tensor (inside the 256-byte slack)
        with pytest.raises(_lib.P2PHDError):
            _ops.check_guards()
    finally:
        _ops._GUARD["on"] = False
        _ops._GUARD["live"] = []
