import sys, os
sys.path.insert(0, "/root/repo")
import torch
from on_device_image_captioning_amd import ops
torch.manual_seed(0)
for (M, N, K) in ((36864, 2304, 768), (36864, 3072, 768), (36864, 768, 3072), (147456, 1152, 384), (9216, 6144, 1536)):
    A = (torch.randn(M, K, device="cuda") * 2).clamp(-448, 448).to(torch.float8_e4m3fn)
    W = (torch.randn(N, K, device="cuda") * 3).clamp(-448, 448).to(torch.float8_e4m3fn)
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    cells = []
    for cfg in range(10):
        try:
            for _ in range(3): ops.gemm(A, W, out=out, tile_cfg=cfg)
        except RuntimeError:
            cells.append("   n/a   "); continue
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(10): ops.gemm(A, W, out=out, tile_cfg=cfg)
        en.record(); torch.cuda.synchronize()
        us = st.elapsed_time(en) * 100
        cells.append(f"{us:6.1f}/{2.0*M*N*K/us/1e6:5.0f}")
    print(f"{M}x{N}x{K}: " + " | ".join(cells), flush=True)
