"""Shared by the oracle (CPU) and product (GPU) tests of the to_spectro configurations besides the published one:
decodes tests/golden/spectro_modes.npz (reference outputs + the random tensors the reference drew, in call order)."""
import os

import numpy as np
import torch

CASES = {  # name -> option overrides (tools/gen_golden.py::gen_spectro_modes)
    "e_mode0": dict(mask_mode="mode0"),
    "e_mode1": dict(mask_mode="mode1"),
    "p_none_mode2": dict(explicit_encoding=False),
    "p_uni_nomask": dict(explicit_encoding=False, phase_encoding_mode="uni_dist", mask_mode=None),
    "p_norm_mode0": dict(explicit_encoding=False, phase_encoding_mode="norm_dist", mask_mode="mode0"),
    "p_norm2_mode1": dict(explicit_encoding=False, phase_encoding_mode="norm_dist2", mask_mode="mode1"),
    "p_scale_mode2": dict(explicit_encoding=False, phase_encoding_mode="scale"),
}


def load():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "spectro_modes.npz"))


def draws(g, name):
    """(phase_noise, noise, noise_sign, pseudo_pha) as the reference drew them: to_spectro draws the phase noise (if the
    mode has one), then the mask randn (always, pix2pixHD_model.py:202), then the randint of mode1; to_audio one randint."""
    kinds = [str(k) for k in g[f"{name}_draw_kinds"]]
    ts = [torch.from_numpy(g[f"{name}_draw{i}"]) for i in range(len(kinds))]
    n_enc = int(g[f"{name}_n_encode_draws"])
    kw = CASES[name]
    i = 0
    phase_noise = None
    if kw.get("phase_encoding_mode") in ("uni_dist", "norm_dist", "norm_dist2"):
        phase_noise = ts[i]; i += 1
    noise = ts[i]; i += 1
    noise_sign = None
    if kw.get("mask_mode", "mode2") == "mode1":
        noise_sign = (2 * ts[i] - 1).float(); i += 1
    assert i == n_enc, (name, kinds, n_enc)
    pseudo = (2 * ts[i] - 1).float() if i < len(ts) else None
    return phase_noise, noise, noise_sign, pseudo
