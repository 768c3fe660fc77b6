#!/usr/bin/env python3
"""Does odic_logsoftmax_topk give the same candidates when other streams keep the chip busy?  Rows of several kinds,
the kernel repeated 3000 times beside a stream of large bf16 GEMMs, every result compared on the device with the first."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import ops

torch.manual_seed(0)
N, V, k = 48, 10000, 3
rows = torch.randn(N, V, device="cuda") * 3.0
rows[8:16] = rows[8:16].round()                       # coarse grid: ties
rows[16:24] = (rows[16:24] * 4).round() / 4
rows[24:32, :5000] = 0.0                              # half the row tied
cv, ci = torch.zeros(N, k, device="cuda"), torch.zeros(N, k, dtype=torch.int32, device="cuda")
ops.logsoftmax_topk(rows, V, None, 0, cv, ci, N, V, k)
ref_v, ref_i = cv.clone(), ci.clone()
A = torch.randn(9216, 768, device="cuda").bfloat16()
Wt = torch.randn(3072, 768, device="cuda").bfloat16()
out = torch.empty(9216, 3072, device="cuda", dtype=torch.bfloat16)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
bad = torch.zeros(2, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for busy in (False, True):
    bad.zero_()
    for it in range(3000):
        if busy and it % 4 == 0:
            with torch.cuda.stream(sb):
                ops.gemm(A, Wt, out=out)
        with torch.cuda.stream(sa):
            cv.zero_(); ci.zero_()
            ops.logsoftmax_topk(rows, V, None, 0, cv, ci, N, V, k)
            bad[0] += (ci != ref_i).any().long()
            bad[1] += (cv != ref_v).any().long()
    torch.cuda.synchronize()
    print(f"busy={busy}: launches with different indices {int(bad[0])}, different values {int(bad[1])} of 3000", flush=True)
