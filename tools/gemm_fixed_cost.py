#!/usr/bin/env python3
"""Where a Swin-L bf16 product's time goes: per-K-tile cost vs the fixed cost of a launch (launch gap, pipeline
fill, epilogue, store tail).  For one output shape (M x N) the product is timed at several K and with each
epilogue; T(K) = fixed + slope·K is fitted per (tile config, epilogue).  Variants interleaved in one process.

    python tools/gemm_fixed_cost.py [--M 9216] [--N 3072] [--cfgs 10,12]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=9216)
ap.add_argument("--N", type=int, default=3072)
ap.add_argument("--cfgs", default="1,10,12")
ap.add_argument("--ks", default="256,768,1536,3072")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--inner", type=int, default=10)
a = ap.parse_args()
M, N = a.M, a.N
KS = [int(k) for k in a.ks.split(",")]
torch.manual_seed(0)
bias = torch.randn(N, device="cuda")
res = torch.randn(M, N, device="cuda")
out16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
out32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
EPI = {"plain->bf16": dict(bias=None, res=None, out=out16, act=ops.ACT_NONE),
       "bias->bf16": dict(bias=bias, res=None, out=out16, act=ops.ACT_NONE),
       "bias+gelu->bf16": dict(bias=bias, res=None, out=out16, act=ops.ACT_GELU),
       "bias+res->f32": dict(bias=bias, res=res, out=out32, act=ops.ACT_NONE)}
print(f"M={M} N={N}; T(K) = fixed + slope*K, least squares over K in {KS}")
for cfg in [int(c) for c in a.cfgs.split(",")]:
    for ename, e in EPI.items():
        ts = []
        for K in KS:
            A = torch.randn(M, K, device="cuda").bfloat16()
            W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
            f = lambda: ops.gemm(A, W, e["bias"], e["res"], out=e["out"], act=e["act"], tile_cfg=cfg)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            tt = []
            for _ in range(a.rounds):
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                for _ in range(a.inner):
                    f()
                en.record()
                en.synchronize()
                tt.append(st.elapsed_time(en) * 1e3 / a.inner)
            ts.append(sorted(tt)[len(tt) // 2])
        n = len(KS)
        mk, mt = sum(KS) / n, sum(ts) / n
        slope = sum((k - mk) * (t - mt) for k, t in zip(KS, ts)) / sum((k - mk) ** 2 for k in KS)
        fixed = mt - slope * mk
        inloop = 2.0 * M * N / slope / 1e6 if slope > 0 else float("nan")
        print(f"cfg{cfg:<3d} {ename:>16s} | " + " ".join(f"K={k}: {t:6.1f}us" for k, t in zip(KS, ts)) +
              f" | fixed {fixed:5.1f} us, slope {slope * 64:5.2f} us per 64 of K = {inloop:6.0f} TFLOP/s in-loop", flush=True)
