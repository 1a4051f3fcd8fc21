#!/usr/bin/env python3
"""In-situ A/B of the bf16 MFMA shape in the gather conv's main loop (tools/ablate_gconv.sh mfma16 "-DP2PHD_ABL_MFMA16"
base ""): the `mfma16` library issues two v_mfma_f32_16x16x32_bf16 per v_mfma_f32_32x32x16_bf16 on the same fragment
registers (same MACs, same LDS / DMA traffic, numerically meaningless) -- does the chip hold a higher clock on that
shape inside THIS kernel (MI355X_MICROARCH.md, DVFS give-back item 7)?  Each library in its own process, interleaved."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(3):
    for name in ("base", "mfma16"):
        env = dict(os.environ, P2PHD_LIB=os.path.join(ROOT, "pix2pixhdaudiosr_amd", "abl", f"libp2phd_{name}.so"))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trunk_only.py")], env=env, capture_output=True, text=True).stdout
        print(rnd, name, out.strip().splitlines()[-1], flush=True)
