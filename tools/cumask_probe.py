#!/usr/bin/env python3
"""CU-masked HIP streams (hipExtStreamCreateWithCUMask) for the decode lanes: does a masked queue replay graphs at
full speed, and what does reserving R CUs for the decoder cost / buy?

    python tools/cumask_probe.py
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from on_device_image_captioning_amd import weights as W
from on_device_image_captioning_amd.pipeline import CaptionPipeline

hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(bits):
    """bits: iterable of enabled CU indices (0..255)"""
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask: {rc}")
    return torch.cuda.ExternalStream(h.value)


torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
model, sd, g = bench.build_model(dev, "bf16", "e2e16")
pipe = CaptionPipeline(model, 16, 3, 20, 79, 77)
img = W.synth_images(16, g).to(dev)
pipe(img)
while pipe.outstanding():
    pipe.collect()
torch.cuda.synchronize()


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def enc_on(s):
    def f():
        with torch.cuda.stream(s):
            pipe.g_enc.replay()
    return f


def dec_on(s):
    def f():
        with torch.cuda.stream(s):
            pipe._reset(0)
            pipe.replay_search(0)
    return f


def both(s_enc, s_dec, n_enc=12, n_dec=12):
    """encode graphs back to back on s_enc while every decode lane replays searches on its stream"""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    d = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in s_dec]
    for l, s in enumerate(s_dec):
        with torch.cuda.stream(s):
            d[l][0].record()
    with torch.cuda.stream(s_enc):
        e0.record()
    for i in range(max(n_enc, n_dec)):
        if i < n_enc:
            with torch.cuda.stream(s_enc):
                pipe.g_enc.replay()
        if i < n_dec:
            for l, s in enumerate(s_dec):
                with torch.cuda.stream(s):
                    pipe._reset(l)
                    pipe.replay_search(l)
    with torch.cuda.stream(s_enc):
        e1.record()
    for l, s in enumerate(s_dec):
        with torch.cuda.stream(s):
            d[l][1].record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n_enc, [d[l][0].elapsed_time(d[l][1]) / n_dec for l in range(len(s_dec))]


print(f"plain streams: encode alone {timeit(enc_on(pipe.s_enc)):.3f} ms, search alone {timeit(dec_on(pipe.s_dec[0])):.3f} ms")
e, dd = both(pipe.s_enc, pipe.s_dec)
print(f"plain streams, encode beside two decode lanes: encode {e:.3f} ms, searches " + ", ".join(f"{x:.3f}" for x in dd))
full = masked_stream(range(256))
print(f"all-CU mask: encode alone {timeit(enc_on(full)):.3f} ms, search alone {timeit(dec_on(full)):.3f} ms")
for R, pattern in ((16, "interleaved"), (32, "interleaved"), (32, "low"), (64, "interleaved")):
    if pattern == "interleaved":
        step = 256 // R
        dec_bits = list(range(0, 256, step))
    else:
        dec_bits = list(range(R))
    enc_bits = [b for b in range(256) if b not in set(dec_bits)]
    s_e = masked_stream(enc_bits)
    s_d = [masked_stream(dec_bits), masked_stream(dec_bits)]
    ea, da = timeit(enc_on(s_e)), timeit(dec_on(s_d[0]))
    e, dd = both(s_e, s_d)
    print(f"R={R:3d} {pattern:12s}: encode alone on {256 - R} CUs {ea:.3f} ms, search alone on {R} CUs {da:.3f} ms | together: "
          f"encode {e:.3f} ms, searches " + ", ".join(f"{x:.3f}" for x in dd), flush=True)
