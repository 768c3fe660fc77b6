#!/usr/bin/env python3
"""Calibration only (not on the product path): what the vendor GEMM (torch.mm → hipBLASLt / rocBLAS) reaches on
the Swin-L bf16 product shapes, plain GEMM without epilogue, next to odic_gemm with its fused epilogue."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import ops

B = 16
shapes = []
for s, C in enumerate((192, 384, 768, 1536)):
    M = B * (96 >> s) ** 2
    shapes += [(M, 3 * C, C, "qkv"), (M, C, C, "proj"), (M, 4 * C, C, "fc1"), (M, C, 4 * C, "fc2")]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(n):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) * 1e3 / n


for M, N, K, kind in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    Wt = W.t()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    us_blas = timeit(lambda: torch.mm(A, Wt, out=out))
    with ops.autotune():
        ops.gemm(A, W, out=out)
    us_ours = timeit(lambda: ops.gemm(A, W, out=out))
    f = 2.0 * M * N * K / 1e6
    print(f"{M:>7d}x{N:>5d}x{K:>5d} {kind:5s} | vendor {us_blas:7.1f} us {f / us_blas:7.1f} TF/s | odic_gemm (plain, bf16 out) "
          f"{us_ours:7.1f} us {f / us_ours:7.1f} TF/s")
