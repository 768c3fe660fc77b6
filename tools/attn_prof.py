#!/usr/bin/env python3
"""A few launches of the bf16 window-attention fast path for one stage — target for rocprofv3 --pmc.
    python tools/attn_prof.py res heads shift [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import ops

res, heads, shift = (int(v) for v in sys.argv[1:4])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
C = heads * 32
torch.manual_seed(0)
qkv = torch.randn(B * res * res, 3 * C, device="cuda").bfloat16()
table = torch.randn(529, heads, device="cuda") * 0.1
dense = ops.shifted_bias_prescaled(table, 12, 32 ** -0.5)
out = torch.empty(B * res * res, C, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    ops.window_attention(qkv, table, B, res, C, heads, 12, shift, out=out, bias_shifted_prescaled=dense)
torch.cuda.synchronize()
