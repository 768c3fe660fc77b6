#!/usr/bin/env python3
"""Decoder-step GEMM shapes (fp32 skinny-M path) timed back to back inside one hipGraph:
µs per launch and effective weight-stream GB/s.   python tools/skinny_bench.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 48
torch.manual_seed(0)
s = torch.cuda.Stream()
for N, K in ((512, 512), (2560, 512), (2048, 512), (512, 2048), (512, 1536), (10000, 512)):
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    Wf, bf, cs = ops.fold_layernorm(W, b, torch.ones(K, device="cuda"), torch.zeros(K, device="cuda"))
    cells = []
    variants = []
    for shp in (0, 1, 2, 3):
        variants.append((f"shape{shp}", lambda shp=shp: ops.gemm(A, W, b, out=out, tile_cfg=shp)))
    variants.append(("auto folded-LN", lambda: ops.gemm(A, Wf, bf, out=out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, ln_fold=(cs, 1e-5))))
    for name, fn in variants:
        with torch.cuda.stream(s):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(64):
                fn()
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            g.replay()
            st.record()
            for _ in range(5):
                g.replay()
            en.record()
        torch.cuda.synchronize()
        us = st.elapsed_time(en) * 1e3 / (5 * 64)
        cells.append(f"{name} {us:6.2f} us")
    print(f"M={M} N={N:5d} K={K:4d} | " + " | ".join(cells))
