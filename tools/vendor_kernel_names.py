#!/usr/bin/env python3
"""Calibration only: which hipBLASLt kernels torch.mm picks on the Swin-L stage-2/3 bf16 shapes (run under
rocprofv3 --kernel-trace; the kernel names carry the macro tile, the MFMA shape and the scheduling mode)."""
import torch

for M, N, K in [(9216, 2304, 768), (9216, 768, 768), (9216, 3072, 768), (9216, 768, 3072), (2304, 6144, 1536)]:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.mm(A, W.t(), out=out)
    torch.cuda.synchronize()
