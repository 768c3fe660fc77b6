#!/usr/bin/env python3
"""norm -> linear of Swin stages 0-1 as two launches (odic_layernorm -> bf16, A-resident product) against the
LayerNorm-while-reading form (odic_gemm_args.a_ln), isolated, interleaved round by round.
    python tools/ln_read_probe.py [--batch 16]          (ODIC_APANEL_BLOCKS=n changes the column split of both forms)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--rounds", type=int, default=7)
a = ap.parse_args()
torch.manual_seed(0)
for s, C in enumerate((192, 384)):
    M = a.batch * (96 >> s) ** 2
    for N, kind, act in ((3 * C, "qkv", ops.ACT_NONE), (4 * C, "fc1", ops.ACT_GELU)):
        x = torch.randn(M, C, device="cuda")
        W = (torch.randn(N, C, device="cuda") * 0.05)
        b = torch.randn(N, device="cuda")
        g1, b1 = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        Wf, bf, _ = ops.fold_layernorm_bf16(W, b, g1, b1)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        xn = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
        cfgs = (50, 52) if C == 192 else (51, 53)
        fns = {"layernorm": lambda: ops.layernorm(x, g1, b1, out=xn)}
        for c in cfgs:
            fns[f"gemm cfg{c}"] = lambda c=c: ops.gemm(xn, Wf, bf, out=out, act=act, tile_cfg=c)
            fns[f"fused cfg{c}"] = lambda c=c: ops.gemm(None, Wf, bf, a_ln=x, out=out, act=act, tile_cfg=c)
        t = {k: [] for k in fns}
        for f in fns.values():
            f()
        torch.cuda.synchronize()
        for _ in range(a.rounds):
            for k, f in fns.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    f()
                e1.record()
                torch.cuda.synchronize()
                t[k].append(e0.elapsed_time(e1) * 100)
        med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
        print(f"{M}x{N}x{C} {kind}: " + "  ".join(f"{k} {v:.1f}" for k, v in med.items()), flush=True)
