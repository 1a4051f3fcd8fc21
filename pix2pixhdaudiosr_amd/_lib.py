"""ctypes binding of libp2phd_hip.so (the C ABI declared in include/p2phd.h).

There is deliberately no CPU fallback: if the shared library is missing or a call fails the
product path raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C pix2pixhdaudiosr_amd/csrc``.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# P2PHD_LIB: load another build of the same ABI (kernel A/B and ablation experiments, tools/ablate_gconv.py)
LIB_PATH = os.environ.get("P2PHD_LIB") or os.path.join(_HERE, "libp2phd_hip.so")
# the same sources with IEEE fp16 as the 16-bit storage type (csrc/common.h; opt.fp16_storage / compute dtype torch.float16)
LIB_PATH_F16 = os.path.join(_HERE, "libp2phd_hip_f16.so")

F32, BF16 = 0, 1
_i64, _i32, _f32, _vp, _sz = C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); must list every symbol of include/p2phd.h (tests check this)
SIGNATURES = {
    "p2phd_last_error": (C.c_char_p, []),
    "p2phd_abi_version": (_i32, []),
    "p2phd_device_info": (_i32, [C.c_char_p, _i32]),
    "p2phd_set_option": (_i32, [C.c_char_p, _i32]),
    "p2phd_reduction_reset": (_i32, [_vp]),
    "p2phd_half_type": (_i32, []),
    "p2phd_probe_gconv": (_i32, [_i32, _i32, _i32, _i32, _i32]),
    "p2phd_probe_gconv_ex": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "p2phd_probe_read": (_i32, [_vp, _i32]),
    "p2phd_launch_count": (_i64, [C.c_char_p, _i32]),
    "p2phd_wait_check": (_i32, [_vp, _i32]),
    "p2phd_mdct4_tables_floats": (C.c_size_t, [_i32]),
    "p2phd_mdct4_tables_fill": (_i32, [_i32, _vp]),
    "p2phd_mdct4_frame_layout": (_i32, [_i64, _i64, _i32, _i32, _i32, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "p2phd_mdct4_fwd": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _i64, _i64, _f32, _vp, _vp]),
    "p2phd_imdct4_fwd": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _i64, _i64, _f32, _vp, _vp]),
    "p2phd_dct_tables_floats": (C.c_size_t, [_i32]),
    "p2phd_dct_tables_fill": (_i32, [_i32, _vp]),
    "p2phd_mdct2_fwd": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _i64, _i64, _f32, _f32, _vp, _vp]),
    "p2phd_imdct2_fwd": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _i64, _i64, _f32, _f32, _vp, _vp]),
    "p2phd_channel_pitch": (_i32, [_i32]),
    "p2phd_conv_out_size": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "p2phd_conv_kmajor_ok": (_i32, [_vp]),
    "p2phd_conv_packed_bytes": (C.c_size_t, [_vp, _i32]),
    "p2phd_conv_pack_weights": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "p2phd_conv_pack_layout": (_i32, [_vp, _i32]),
    "p2phd_conv_fwd_workspace_bytes": (C.c_size_t, [_vp]),
    "p2phd_conv_fwd": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "p2phd_conv_fp8_eligible": (_i32, [_vp]),
    "p2phd_conv_fp8_packed_bytes": (C.c_size_t, [_vp]),
    "p2phd_conv_fp8_pack_weights": (_i32, [_vp, _vp, _vp, _vp]),
    "p2phd_conv_fwd_fp8": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "p2phd_conv_dgrad_workspace_bytes": (C.c_size_t, [_vp]),
    "p2phd_conv_dgrad": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "p2phd_conv_dgrad_bsum_ok": (_i32, [_vp]),
    "p2phd_conv_dgrad_bsum_pays": (_i32, [_vp]),
    "p2phd_conv_dgrad_bsum_workspace_bytes": (_sz, [_vp]),
    "p2phd_conv_dgrad_bsum": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _vp, _vp]),
    "p2phd_conv_dgrad_act": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "p2phd_conv_wgrad_workspace_bytes": (C.c_size_t, [_vp]),
    "p2phd_conv_wgrad": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "p2phd_conv_wgrad_acc": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "p2phd_instnorm_act_fwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _f32, _i32, _vp]),
    "p2phd_instnorm_act_fwd_q8": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _f32, _i32, _vp]),
    "p2phd_instnorm_act_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _f32, _i32, _vp]),
    "p2phd_instnorm_act_bwd_acc": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _f32, _i32, _vp]),
    "p2phd_instnorm_act_bwd_two_pass": (_i32, [_i32, _i32, _i64, _i32]),
    "p2phd_conv_reflect_extras_elems": (C.c_size_t, [_vp]),
    "p2phd_instnorm_act_bwd_rx": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp]),
    "p2phd_conv_dgrad_rx": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "p2phd_conv_lazy_ok": (_i32, [_vp]),
    "p2phd_conv_fwd_lazy": (_i32, [_vp, _vp, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "p2phd_conv_wgrad_lazy": (_i32, [_vp, _vp, _vp, _i32, _f32, _vp, _vp, _vp, _i32, _vp, _vp]),
    "p2phd_instnorm_act_bwd_apply": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _f32, _i32, _vp]),
    "p2phd_act_bwd": (_i32, [_i32, _vp, _vp, _vp, _i64, _i32, _vp]),
    "p2phd_act_bwd_db": (_i32, [_i32, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i32, _vp]),
    "p2phd_avgpool3s2_fwd": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "p2phd_avgpool3s2_bwd": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "p2phd_nchw_to_nhwc": (_i32, [_i32, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _vp]),
    "p2phd_nchw_cat_to_nhwc": (_i32, [_i32, _vp, _vp, _i32, _vp, _i32, _i64, _i32, _vp]),
    "p2phd_nhwc_to_nchw": (_i32, [_i32, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _vp]),
    "p2phd_loss_fwd": (_i32, [_i32, _i32, _vp, _vp, _f32, _i64, _i32, _f32, _vp, _vp]),
    "p2phd_loss_bwd": (_i32, [_i32, _i32, _vp, _vp, _f32, _i64, _i32, _f32, _vp, _vp, _vp]),
    "p2phd_adam_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _i64, _f32, _vp]),
    "p2phd_adam_step_dev": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _f32, _f32, _f32, _f32, _vp]),
    "p2phd_zero_segments": (_i32, [_vp, _vp, _i32, _vp]),
    "p2phd_adam_step_scaled": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _f32, _f32, _f32, _f32, _vp, _i32, _vp]),
    "p2phd_scaler_update": (_i32, [_vp, _f32, _f32, _i32, _vp]),
    "p2phd_spectro_partials_floats": (_i64, [_i64, _i64, _i64]),
    "p2phd_spectro_encode": (_i32, [_vp, _i64, _i64, _i64, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "p2phd_spectro_encode_ex": (_i32, [_vp, _i64, _i64, _i64, _i32, _f32, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "p2phd_spectro_decode": (_i32, [_vp, _vp, _i64, _i64, _i64, _f32, _f32, _vp, _vp]),
    "p2phd_spectro_decode_signed": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _f32, _f32, _vp, _vp]),
    "p2phd_stft_tables_floats": (_sz, [_i32]),
    "p2phd_stft_tables_fill": (_i32, [_i32, _vp]),
    "p2phd_metrics_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32, _i32, _i32]),
    "p2phd_resample_geometry": (_i32, [_i32, _i32, _i32, C.c_double, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "p2phd_resample_kernel_floats": (_sz, [_i32, _i32, _i32, C.c_double]),
    "p2phd_resample_kernel_fill": (_i32, [_i32, _i32, _i32, C.c_double, _vp]),
    "p2phd_resample_out_len": (_i64, [_i64, _i32, _i32]),
    "p2phd_resample_fwd": (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, C.c_double, _vp, _vp, _i64, _vp]),
    "p2phd_audio_metrics": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
}


class ConvDesc(C.Structure):
    """struct p2phd_conv_desc (include/p2phd.h)."""
    _fields_ = [(n, C.c_int32) for n in ("N", "C", "H", "W", "K", "R", "S", "stride", "pad", "pad_mode",
                                         "transposed", "opad", "dtype", "w_layout")]

_libs = {}
_last = [None]            # the library the latest call went to (p2phd_last_error is per library)
_process_options = {}     # options set through set_option_all: every library of the process holds its own copy of them


class P2PHDError(RuntimeError):
    pass


def _load(path, want_half):
    if not os.path.isfile(path):
        raise ImportError(
            f"{path} is missing: the HIP extension has not been built "
            "(run `make -C pix2pixhdaudiosr_amd/csrc` or `__graft_entry__.build()`); "
            "this package has no CPU fallback")
    l = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args
    if l.p2phd_half_type() != want_half:
        raise P2PHDError(f"{path}: 16-bit storage type {l.p2phd_half_type()} (1 bf16, 2 fp16), expected {want_half}")
    # P2PHD_OPTIONS="name=value,...": tuning overrides of p2phd_set_option for A/B runs of whole programs (bench.py, tools)
    for item in filter(None, os.environ.get("P2PHD_OPTIONS", "").split(",")):
        name, _, value = item.partition("=")
        if l.p2phd_set_option(name.strip().encode(), int(value)) != 0:
            raise P2PHDError(f"P2PHD_OPTIONS: {l.p2phd_last_error().decode()}")
    for name, value in _process_options.items():                  # a library loaded late starts with the process's settings
        if l.p2phd_set_option(name, value) != 0:
            raise P2PHDError(f"set_option({name!r}): {l.p2phd_last_error().decode()}")
    return l


def set_option_all(name, value):
    """p2phd_set_option on EVERY library of the process (bf16 and fp16 builds keep separate option globals: a CU count or a
    kernel switch set on one only would leave the other sizing its launches for the wrong machine), remembered for libraries
    that are loaded later."""
    if isinstance(name, str):
        name = name.encode()
    _process_options[name] = int(value)
    for l in list(_libs.values()):
        if l.p2phd_set_option(name, int(value)) != 0:
            raise P2PHDError(f"set_option({name!r}): {l.p2phd_last_error().decode()}")


def lib(kind="bf16"):
    """The loaded library (`kind`: 'bf16' = libp2phd_hip.so, 'f16' = libp2phd_hip_f16.so); raises (never falls back) when it has
    not been built."""
    l = _libs.get(kind)
    if l is None:
        l = _libs[kind] = _load(LIB_PATH if kind == "bf16" else LIB_PATH_F16, 1 if kind == "bf16" else 2)
    _last[0] = l
    return l


def lib_for(dtype):
    """The library whose 16-bit storage type is `dtype` (float32 tensors: the default library -- both hold the same f32 kernels)."""
    return lib("f16" if dtype == torch.float16 else "bf16")


def loaded():
    return list(_libs.values())


def check(rc, what=""):
    if rc != 0:
        msg = (_last[0] or lib()).p2phd_last_error().decode("utf-8", "replace")
        raise P2PHDError(f"{what or 'p2phd'} failed (code {rc}): {msg}")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def require_gpu_tensor(t, name, dtype=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise P2PHDError(f"{name}: expected a tensor on the GPU (this build has no CPU path)")
    if dtype is not None and t.dtype != dtype:
        raise P2PHDError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise P2PHDError(f"{name}: expected a contiguous tensor")
    return t
