#!/usr/bin/env python3
"""Time every bf16 GEMM shape of the Swin-L/384 forward (batch B) under each tile configuration, with the
epilogue the product path uses.  Variants are interleaved round by round in ONE process (cdna_hip_programming.md
§5.4 rule 24) on random data; the median over rounds is printed.  `--vendor` adds torch.mm (hipBLASLt; plain GEMM,
calibration only — never on the product path).

    python tools/gemm_tune.py [--batch 16] [--cfgs 0,1,7,10,16,17,23,26] [--stages 2,3] [--rounds 7] [--vendor]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--cfgs", default="0,1,7,10,16,17,23,26")
ap.add_argument("--stages", default="0,1,2,3")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--inner", type=int, default=10)
ap.add_argument("--vendor", action="store_true")
ap.add_argument("--x3", action="store_true", help="split-fp16 operands (gemm_x3.hip tile configurations)")
ap.add_argument("--lib", default=None, help="another build of libodic_hip.so (A/B of kernel changes)")
a = ap.parse_args()
if a.lib:
    from on_device_image_captioning_amd import _hip
    _hip.LIB_PATH = os.path.abspath(a.lib)
CFGS = [int(c) for c in a.cfgs.split(",")]
shapes = []
for s, C in enumerate((192, 384, 768, 1536)):
    if str(s) not in a.stages.split(","):
        continue
    M = a.batch * (96 >> s) ** 2
    shapes += [(M, 3 * C, C, "qkv"), (M, C, C, "proj+res"), (M, 4 * C, C, "fc1+gelu"), (M, C, 4 * C, "fc2+res")]
    if s < 3:
        shapes.append((M // 4, 2 * C, 4 * C, "merge"))
torch.manual_seed(0)
names = [f"cfg{c}" for c in CFGS] + (["vendor"] if a.vendor else [])
print(f"{'shape':>26s} {'kind':>9s} | " + " | ".join(f"{n:>7s} us/TF" for n in names))
for M, N, K, kind in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    if a.x3:
        A, W = ops.cast_h2(A.float()), ops.cast_h2(W.float() * 4096.0)
    Wt = W.t()
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda") if "res" in kind else None
    odt = torch.float32 if ("res" in kind or kind == "merge") else (ops.H2_DTYPE if a.x3 else torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=odt)
    out16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    act = ops.ACT_GELU if "gelu" in kind else ops.ACT_NONE
    fns = {}
    for cfg in CFGS:
        def f(cfg=cfg):
            ops.gemm(A, W, bias if kind != "merge" else None, res, out=out, act=act, tile_cfg=cfg)
        try:
            f()
            fns[f"cfg{cfg}"] = f
        except RuntimeError:
            pass
    if a.vendor and not a.x3:
        fns["vendor"] = lambda: torch.mm(A, Wt, out=out16)
    times = {n: [] for n in fns}
    for n, f in fns.items():
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for n, f in fns.items():
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(a.inner):
                f()
            en.record()
            en.synchronize()
            times[n].append(st.elapsed_time(en) * 1e3 / a.inner)
    cells = []
    for n in names:
        if n not in times:
            cells.append("      n/a     ")
            continue
        t = sorted(times[n])[len(times[n]) // 2]
        cells.append(f"{t:7.1f}/{2.0 * M * N * K / t / 1e6:6.0f}")
    print(f"{M:>8d}x{N:>5d}x{K:>5d} {kind:>9s} | " + " | ".join(cells), flush=True)
