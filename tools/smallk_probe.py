#!/usr/bin/env python3
"""Where do the K = 192 / 384 products of Swin stages 0-1 lose their time?  
Each shape with three epilogues (plain store / + bias / + bias + GELU) under every default tile configuration, beside
the pure-traffic floor of the same bytes: a device fill of the output (write only) and a copy of output-sized data
(read + write).  Variants interleaved round by round in one process; median over rounds.

    python tools/smallk_probe.py [--batch 16] [--cfgs 0,1,7,10,44]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--cfgs", default="0,7,10,50,51,52,53")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--inner", type=int, default=10)
a = ap.parse_args()
CFGS = [int(c) for c in a.cfgs.split(",")]
torch.manual_seed(0)
shapes = []
for s, C in enumerate((192, 384)):
    M = a.batch * (96 >> s) ** 2
    shapes += [(M, 3 * C, C, "qkv"), (M, 4 * C, C, "fc1")]


def timeit(fns):
    ev = {k: [] for k in fns}
    for k, f in fns.items():
        f()
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.inner):
                f()
            e1.record()
            torch.cuda.synchronize()
            ev[k].append(e0.elapsed_time(e1) * 1e3 / a.inner)
    return {k: sorted(v)[len(v) // 2] for k, v in ev.items()}


for M, N, K, kind in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    src = torch.randn(M, N, device="cuda").bfloat16()
    fns = {"fill(out)": lambda: out.fill_(1.0), "copy(out)": lambda: out.copy_(src)}
    for cfg in CFGS:
        for ep, (b, act) in {"plain": (None, ops.ACT_NONE), "bias": (bias, ops.ACT_NONE), "gelu": (bias, ops.ACT_GELU)}.items():
            def f(cfg=cfg, b=b, act=act):
                ops.gemm(A, W, b, None, out=out, act=act, tile_cfg=cfg)
            try:
                f()
                fns[f"cfg{cfg}/{ep}"] = f
            except RuntimeError:
                pass
    t = timeit(fns)
    mb = (M * K + N * K + M * N) * 2 / 1e6
    print(f"{M}x{N}x{K} {kind}: algorithmic {mb:.0f} MB ({mb / 5e3 * 1e3 / 1e3:.0f} us at 5 TB/s)", flush=True)
    for k, v in t.items():
        print(f"    {k:>14s} {v:8.1f} us   {mb / v / 1e3 * 1e3 / 1e3:6.2f} TB/s-equivalent", flush=True)
