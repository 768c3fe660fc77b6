#!/usr/bin/env python3
"""The expansion encoder's bf16 products (2304 rows, or 16 batches of 144 / 512 / 992 rows) under each tile configuration:
small outputs that leave most CUs idle under 128-wide tiles.     python tools/enc_gemm_probe.py [--cfgs 0,1,48,49]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--cfgs", default="0,1,7,44,48,49")
a = ap.parse_args()
CFGS = [int(c) for c in a.cfgs.split(",")]
torch.manual_seed(0)
shapes = [(2304, 512, 1536, 1, "input_linear"), (2304, 512, 512, 1, "key / selector"), (2304, 2048, 512, 1, "ff1 + relu"),
          (2304, 512, 2048, 1, "ff2 + res"), (2304, 3072, 512, 1, "cross k/v"), (1024, 144, 512, 16, "class a|b (T)"),
          (992, 144, 512, 16, "z = Q K^T"), (512, 992, 192, 16, "A^T / B^T"), (144, 512, 1024, 16, "backward")]
print(f"{'shape':>28s} | " + " | ".join(f"cfg{c:>3d} us" for c in CFGS))
for M, N, K, batch, name in shapes:
    A = torch.randn(batch, M, K, device="cuda").bfloat16()
    W = (torch.randn(batch, N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(batch, M, N, device="cuda", dtype=torch.float32)
    cells = []
    for cfg in CFGS:
        def f():
            ops.gemm(A, W, out=out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, batch=batch, strideA=M * K, strideW=N * K,
                     strideC=M * N, tile_cfg=cfg)
        try:
            f()
        except RuntimeError:
            cells.append("   n/a   ")
            continue
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)
        cells.append(f"{sorted(ts)[2]:9.1f}")
    print(f"{name:>14s} {M}x{N}x{K}x{batch:<2d} | " + " | ".join(cells), flush=True)
