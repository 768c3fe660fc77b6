#!/usr/bin/env python3
"""Isolated timing of odic_window_attention (bf16) per Swin-L stage at batch B: table kernel (v1) vs
dense-bias kernel (v2).  Prints µs per launch and algorithmic TB/s (36,864 B per (window, head))."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
for res, heads in ((96, 6), (48, 12), (24, 24), (12, 48)):
    C = heads * 32
    qkv = torch.randn(B * res * res, 3 * C, device="cuda").bfloat16()
    table = (torch.randn(529, heads, device="cuda") * 0.1)
    dense = ops.shifted_bias_prescaled(table, 12, 32 ** -0.5)
    out = torch.empty(B * res * res, C, device="cuda", dtype=torch.bfloat16)
    inst = B * (res // 12) ** 2 * heads
    for shift in (0, 6 if res > 12 else 0):
        cells = []
        for name, kw in (("table", {}), ("packed", {"bias_shifted_prescaled": dense})):
            for _ in range(3):
                ops.window_attention(qkv, table, B, res, C, heads, 12, shift, out=out, **kw)
            torch.cuda.synchronize()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(20):
                ops.window_attention(qkv, table, B, res, C, heads, 12, shift, out=out, **kw)
            en.record(); torch.cuda.synchronize()
            us = st.elapsed_time(en) * 1e3 / 20
            cells.append(f"{name} {us:7.1f} us {inst * 36864 / us / 1e6:6.2f} TB/s")
        print(f"res {res:3d} heads {heads:2d} shift {shift}  instances {inst:6d} | " + " | ".join(cells))
