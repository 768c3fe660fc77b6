#!/usr/bin/env python3
"""What does RESIDENCY alone cost?  The encode graph replayed beside N idle (or VALU-busy) resident blocks of 768
threads — the footprint a persistent decoder-step kernel would have — compared with the two decode lanes."""
import ctypes
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from on_device_image_captioning_amd.pipeline import CaptionPipeline

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "spin", "libspin.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "--offload-arch=gfx950",
                           os.path.join(HERE, "spin", "spin.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.spin_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
model, sd, g = bench.build_model(dev, "bf16", "e2e16")
pipe = CaptionPipeline(model, 16, 3, 20, 79, 77, decode_lanes=2)
sink = torch.zeros(1, device=dev)
side = torch.cuda.Stream()


def encode_ms(n=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(pipe.s_enc):
        e0.record()
        for _ in range(n):
            pipe.g_enc.replay()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"encode alone {encode_ms():.3f} ms")
for blocks, threads, mode in ((32, 768, 0), (32, 768, 1), (64, 768, 0), (64, 256, 0), (128, 256, 0), (256, 256, 0), (256, 256, 1)):
    torch.cuda.synchronize()
    lib.spin_launch(blocks, threads, int(100e6 * 0.08), mode, sink.data_ptr(), side.cuda_stream)   # ~80 ms at 100 MHz
    ms = encode_ms()
    torch.cuda.synchronize()
    print(f"{blocks:4d} resident blocks x {threads} threads, {'s_sleep' if mode == 0 else 'busy VALU'}: encode {ms:.3f} ms")
