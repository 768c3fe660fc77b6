#!/usr/bin/env python3
"""Drives tools/csrc/pkfma_probe.hip: does `v_pk_fma_f32 ... op_sel:[0,1,0]` compute the same sums as its
`op_sel_hi:[1,0,1]` twin and as scalar FMAs — alone, with global loads in flight, and beside a GEMM stream?
Prints one line per (mode, background) with the number of lanes x iterations whose four results disagreed."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "tools", "_build", "libpkfma_probe.so"))
P = C.c_void_p
lib.pkfma_probe.argtypes = [C.c_int, P, C.c_int64, C.c_int, C.c_int, P, P, P]
dev = "cuda"
n = 1 << 26                                              # 256 MiB of floats: the loads miss L2
buf = torch.rand(n, device=dev)
sink = torch.zeros(4, device=dev)
MODES = {0: "FMAs only", 1: "+ 12 global loads in flight", 2: "+ VALU write of the unused source dword after form A",
         3: "loads in flight + that write", 7: "loads + s_nop 7 + that write", 5: "loads + s_nop 7"}
from on_device_image_captioning_amd import ops  # noqa: E402

bg_a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
bg_o = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)
bg_s = torch.cuda.Stream()
report = {}
for background in (False, "vendor GEMM", "LDS-DMA GEMM of this library"):
    for mode, what in MODES.items():
        mism = torch.zeros(16, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        if background:                                  # a GEMM stream beside the probe (other waves' memory traffic)
            with torch.cuda.stream(bg_s):
                for _ in range(60):
                    if background == "vendor GEMM":
                        bg_a @ bg_a
                    else:                               # global_load_lds / buffer_load ... lds staging (csrc/gemm_bf16.hip)
                        ops.gemm(bg_a, bg_a, out=bg_o, tile_cfg=0)
        for _ in range(20):
            rc = lib.pkfma_probe(mode, buf.data_ptr(), n, 400, 1024, mism.data_ptr(), sink.data_ptr(),
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc
        torch.cuda.synchronize()
        m = mism.cpu().tolist()
        key = f"mode {mode} ({what}){' beside: ' + background if background else ''}"
        report[key] = {"events": m[0], "per_16_lane_quarter": m[1:5], "which_results_differed_mask": m[8],
                       "lane_iterations": 20 * 1024 * 256 * 400}
        print(key, report[key], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(report, open(os.path.join(ROOT, "gpurun_out", "pkfma_probe.json"), "w"), indent=1)
