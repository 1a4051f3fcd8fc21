#!/usr/bin/env python3
"""Launch only the dominant kernel (trunk Conv3x3 768->768 @32x16, B=32, bf16) a few times: target for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import time_trunk_conv
sec, flops = time_trunk_conv(32, iters=10)
print(f"trunk conv {sec*1e6:.1f} us/launch, {flops/sec/1e12:.0f} TFLOP/s")
