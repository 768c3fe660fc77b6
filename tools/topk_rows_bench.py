#!/usr/bin/env python3
"""odic_logsoftmax_topk time against the number of rows at fixed k, and against k at fixed rows."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import ops
from tools.topk_bench import graph_time

V = 10000
lg = torch.randn(256, V, device="cuda")
for k in (1, 3, 4, 5, 8):
    cells = []
    for N in (16, 48, 80, 128, 256):
        cv, ci = torch.zeros(N, k, device="cuda"), torch.zeros(N, k, dtype=torch.int32, device="cuda")
        cells.append(f"N={N}: {graph_time(lambda: ops.logsoftmax_topk(lg, V, None, 0, cv, ci, N, V, k)):6.2f}")
    print(f"k={k} | " + " | ".join(cells), flush=True)
