#!/usr/bin/env python3
"""Which co-resident kernel makes `v_pk_fma_f32 ... op_sel:[0,1,0]` return a wrong low half in lanes 48-63?
(tools/csrc/pkfma_probe.hip mode 0: registers only, no memory traffic in the probe itself.)"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from on_device_image_captioning_amd import ops  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "tools", "_build", "libpkfma_probe.so"))
P = C.c_void_p
lib.pkfma_probe.argtypes = [C.c_int, P, C.c_int64, C.c_int, C.c_int, P, P, P]
dev = "cuda"
n = 1 << 24
buf = torch.rand(n, device=dev)
sink = torch.zeros(4, device=dev)
a16 = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
o16 = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)
a32 = torch.randn(2048, 2048, device=dev)
ah2 = ops.cast_h2(a32)
x = torch.randn(9216, 768, device=dev)
g, b = torch.ones(768, device=dev), torch.zeros(768, device=dev)
NEIGH = {"none": lambda: None, "vendor bf16 GEMM": lambda: a16 @ a16}
for cfg in (0, 1, 2, 3, 4, 6, 7, 8, 10):
    NEIGH[f"odic bf16 GEMM tile_cfg {cfg}"] = (lambda c: (lambda: ops.gemm(a16, a16, out=o16, tile_cfg=c)))(cfg)
for cfg in (0, 1):
    NEIGH[f"odic x3 GEMM tile_cfg {cfg}"] = (lambda c: (lambda: ops.gemm(ah2, ah2, out_dtype=torch.float32, tile_cfg=c)))(cfg)
NEIGH["odic fp32 GEMM"] = lambda: ops.gemm(a32, a32)
NEIGH["odic layernorm"] = lambda: ops.layernorm(x, g, b, out_dtype=torch.bfloat16)
NEIGH["torch elementwise (a32 * 1.5)"] = lambda: a32 * 1.5
lib.hold_regs.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, P, P]
hold_out = torch.zeros(4, dtype=torch.int32, device=dev)
for nv in (32, 48, 64, 96, 128, 160):
    for blocks_per_cu in (1, 2, 3):
        NEIGH[f"idle neighbour holding {nv} VGPRs, {blocks_per_cu} block(s) of 4 waves per CU"] = (
            lambda v, bpc: (lambda: lib.hold_regs(v, 300, 256 * bpc, 0, hold_out.data_ptr(),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))))(nv, blocks_per_cu)
s2 = torch.cuda.Stream()
rep = {}
for name, fn in NEIGH.items():
    mism = torch.zeros(16, dtype=torch.int32, device=dev)
    fn()
    torch.cuda.synchronize()
    with torch.cuda.stream(s2):
        for _ in range(6 if name.startswith("idle") else 60):
            fn()
    for _ in range(10):
        assert lib.pkfma_probe(0, buf.data_ptr(), n, 400, 1024, mism.data_ptr(), sink.data_ptr(),
                               C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    m = mism.cpu().tolist()
    rep[name] = {"events": m[0], "quarters": m[1:5], "mask": m[8]}
    print(f"{name:64s} events {m[0]:8d} quarters {m[1:5]} result mask {m[8]}", flush=True)
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "pkfma_neighbours.json"), "w"), indent=1)
