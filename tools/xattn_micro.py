#!/usr/bin/env python3
"""Standalone comparison of the cross-attention step kernels on one random problem (bench shape: 16 images x 3 beams,
S = 144, d = 512, 8 heads): the product library's kernel, the diagnostic build's plain form (same source, other
translation unit) and its software-pipelined form (PF) — bitwise, run to run, and against fp64."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from on_device_image_captioning_amd import ops  # noqa: E402

dbg = C.CDLL(os.path.join(ROOT, "tools", "_build", "libodic_dbg.so"))
P, I32, I64 = C.c_void_p, C.c_int32, C.c_int64
dbg.odic_dbg_cross_attn_step.argtypes = [C.c_int, P, I64, P, I64, I32, I32, P, P, P, I64, I32, I32, I32, I32, I32, P]
torch.manual_seed(0)
dev = "cuda"
n_img, beams, S, d, heads = 16, 3, 144, 512, 8
N = n_img * beams
q = torch.randn(N, d, device=dev)
kv = torch.randn(n_img, S, 6 * d, device=dev)
enc_len = torch.full((n_img,), S, dtype=torch.int32, device=dev)
valid = torch.ones(N, dtype=torch.int32, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(which, koff=2 * d, voff=3 * d):
    out = torch.empty(N, d, device=dev)
    if which == "main":
        ops.cross_attn_step(q, d, kv, 6 * d, koff, voff, enc_len, valid, out, d, N, n_img, S, d, heads)
    else:
        assert dbg.odic_dbg_cross_attn_step(which, q.data_ptr(), d, kv.data_ptr(), 6 * d, koff, voff, enc_len.data_ptr(),
                                            valid.data_ptr(), out.data_ptr(), d, N, n_img, S, d, heads, st) == 0
    torch.cuda.synchronize()
    return out


a, b0, b1 = run("main"), run(0), run(1)
a2 = run("main")
print("main run-to-run identical:", all(torch.equal(a, run("main")) for _ in range(20)), "| second call == first:", torch.equal(a, a2))
print("dbg plain run-to-run     :", all(torch.equal(b0, run(0)) for _ in range(20)))
dm = (a != b0)
if dm.any():
    idx = dm.nonzero()
    print("main vs dbg: rows (seq) histogram:", torch.bincount(idx[:, 0], minlength=N).tolist())
    print("main vs dbg: head histogram:", torch.bincount(idx[:, 1] // 64, minlength=8).tolist())
    ul = (a.view(torch.int32) - b0.view(torch.int32)).abs()[dm]
    print("main vs dbg: ulp histogram:", torch.bincount(ul.clamp(max=8)).tolist())
    print("main err vs fp64 where differing, dbg err:", None)
print("main == dbg plain      :", torch.equal(a, b0), int((a != b0).sum()))
print("dbg plain == dbg PF    :", torch.equal(b0, b1), int((b0 != b1).sum()), "of", a.numel())
print("PF run-to-run identical:", all(torch.equal(b1, run(1)) for _ in range(20)))
K = kv[:, :, 2 * d:3 * d].double().view(n_img, S, heads, 64)
V = kv[:, :, 3 * d:4 * d].double().view(n_img, S, heads, 64)
qq = q.double().view(n_img, beams, heads, 64)
sc = torch.einsum("ibhc,ishc->ibhs", qq, K) / 8.0
ref = torch.einsum("ibhs,ishc->ibhc", torch.softmax(sc, -1), V).reshape(N, d)
for name, t in (("main", a), ("dbg plain", b0), ("dbg PF", b1)):
    print(f"{name:10s} max |err| vs fp64 {float((t.double() - ref).abs().max()):.3e}")
for form in (1, 2, 3, 4):
    bad_runs, bad_el = 0, 0
    for _ in range(200):
        r = run(form)
        ne = int((r != b0).sum())
        bad_runs += ne > 0
        bad_el += ne
    print(f"form {form} alone on the chip: {bad_runs} of 200 launches differ from the plain kernel ({bad_el} elements)")
# beside a GEMM stream
bg = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
bgo = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)
s2 = torch.cuda.Stream()
for kind, form in [(k, f) for k in ("vendor GEMM", "LDS-DMA GEMM of this library (128x128 tiles)", "LDS-DMA GEMM (128x64 tiles, 3 blocks per CU)") for f in (1, 2, 3, 4)]:
    torch.cuda.synchronize()
    with torch.cuda.stream(s2):
        for _ in range(40):
            if kind == "vendor GEMM":
                bg @ bg
            else:
                ops.gemm(bg, bg, out=bgo, tile_cfg=1 if "128x128" in kind else 0)
    bad_runs, bad_el = 0, 0
    outs = []
    for _ in range(200):
        out = torch.empty(N, d, device=dev)
        assert dbg.odic_dbg_cross_attn_step(form, q.data_ptr(), d, kv.data_ptr(), 6 * d, 2 * d, 3 * d, enc_len.data_ptr(),
                                            valid.data_ptr(), out.data_ptr(), d, N, n_img, S, d, heads, st) == 0
        outs.append(out)
    torch.cuda.synchronize()
    for r in outs:
        ne = int((r != b0).sum())
        bad_runs += ne > 0
        bad_el += ne
    print(f"form {form} beside {kind}: {bad_runs} of 200 launches differ ({bad_el} elements)")
df = (b0 != b1)
if df.any():
    idx = df.nonzero()
    print("differing channels (mod 64) histogram:", torch.bincount(idx[:, 1] % 64, minlength=64).tolist())
    print("differing per key-group thread g? rows histogram (first 12):", torch.bincount(idx[:, 0], minlength=N).tolist()[:12])
    ulps = (b0.view(torch.int32) - b1.view(torch.int32)).abs()[df]
    print("ulp distance histogram:", torch.bincount(ulps.clamp(max=8)).tolist())
