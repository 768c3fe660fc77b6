#!/usr/bin/env python3
"""Time every bf16 GEMM shape of the Swin-L/384 forward (batch B) under each tile configuration.
Run on the GPU box:  python tools/gemm_tune.py [B]   → table of µs and TFLOP/s per (shape, config)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import _hip, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
CFGS = [int(c) for c in os.environ.get('ODIC_CFGS', '0,1,2,7,8,9,10,11').split(',')]
lib = _hip.load()
shapes = []
for s, C in enumerate((192, 384, 768, 1536)):
    M = B * (96 >> s) ** 2
    shapes += [(M, 3 * C, C, "qkv"), (M, C, C, "proj+res"), (M, 4 * C, C, "fc1+gelu"), (M, C, 4 * C, "fc2+res")]
    if s < 3:
        shapes.append((M // 4, 2 * C, 4 * C, "merge"))
torch.manual_seed(0)
print(f"{'shape':>26s} {'kind':>9s} | " + " | ".join(f"cfg{c:<2d} us / TF/s" for c in CFGS))
for M, N, K, kind in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda") if "res" in kind else None
    odt = torch.float32 if ("res" in kind or kind == "merge") else torch.bfloat16
    out = torch.empty(M, N, device="cuda", dtype=odt)
    act = ops.ACT_GELU if "gelu" in kind else ops.ACT_NONE
    cells = []
    for cfg in CFGS:
        lib.odic_gemm_bf16_force_config(cfg)
        try:
            for _ in range(3):
                ops.gemm(A, W, bias, res, out=out, act=act)
        except RuntimeError:
            cells.append("     n/a        ")
            continue
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        st.record()
        for _ in range(n):
            ops.gemm(A, W, bias, res, out=out, act=act)
        en.record()
        torch.cuda.synchronize()
        us = st.elapsed_time(en) * 1e3 / n
        cells.append(f"{us:8.1f} / {2.0 * M * N * K / us / 1e6:6.1f}")
    lib.odic_gemm_bf16_force_config(-1)
    print(f"{M:>8d}x{N:>5d}x{K:>5d} {kind:>9s} | " + " | ".join(cells))
