#!/usr/bin/env python3
"""One bf16 GEMM shape under one tile config, a few launches — target for rocprofv3 --pmc.
    python tools/gemm_prof.py M N K cfg [act]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import _hip, ops

M, N, K, cfg = (int(v) for v in sys.argv[1:5])
act = int(sys.argv[5]) if len(sys.argv) > 5 else 0
lib = _hip.load()
torch.manual_seed(0)
A = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    ops.gemm(A, W, bias, out=out, act=act, tile_cfg=cfg)
torch.cuda.synchronize()
