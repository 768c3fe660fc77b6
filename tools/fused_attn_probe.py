#!/usr/bin/env python3
"""norm1 -> qkv -> attention core of a width-192 Swin block at batch B: LayerNorm-while-reading product + window
attention (two launches) against odic_swin_qkv_attention (one), isolated, interleaved.   python tools/fused_attn_probe.py [16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
res, C, heads, ws = 96, 192, 6, 12
torch.manual_seed(0)
x = torch.randn(B * res * res, C, device="cuda")
W = torch.randn(3 * C, C, device="cuda") * 0.05
b = torch.randn(3 * C, device="cuda")
Wf, bf, _ = ops.fold_layernorm_bf16(W, b, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"))
table = torch.randn(529, heads, device="cuda") * 0.1
dense = ops.shifted_bias_prescaled(table, ws, 32 ** -0.5)
qkv = torch.empty(B * res * res, 3 * C, device="cuda", dtype=torch.bfloat16)
out = torch.empty(B * res * res, C, device="cuda", dtype=torch.bfloat16)
for shift in (0, 6):
    fns = {"ln+qkv": lambda: ops.gemm(None, Wf, bf, a_ln=x, out=qkv),
           "attention": lambda: ops.window_attention(qkv, table, B, res, C, heads, ws, shift, out=out, bias_shifted_prescaled=dense),
           "fused": lambda: ops.swin_qkv_attention(x, Wf, bf, dense, B, res, C, heads, ws, shift, out=out)}
    t = {k: [] for k in fns}
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    for _ in range(7):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            t[k].append(e0.elapsed_time(e1) * 100)
    print(f"shift {shift}: " + "  ".join(f"{k} {sorted(v)[3]:.1f} us" for k, v in t.items()), flush=True)
