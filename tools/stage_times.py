#!/usr/bin/env python3
"""Isolated stage times of the captioning pipeline on one GPU: encode graph alone, decode loop
alone, and both overlapped (what bench.py measures)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from on_device_image_captioning_amd import weights as W
from on_device_image_captioning_amd.pipeline import CaptionPipeline

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model, sd, g = bench.build_model(dev, "bf16")
pipe = CaptionPipeline(model, B, 3, 20, 79, 77)
img = W.synth_images(B, g).to(dev)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def enc():
    with torch.cuda.stream(pipe.s_enc):
        pipe.g_enc.replay()


def dec():
    with torch.cuda.stream(pipe.s_dec[0]):
        pipe._reset(0)
        for _ in range(pipe.steps):
            pipe.g_step[0].replay()


def both():
    pipe.submit(img)
    if pipe.full():
        pipe.collect()


pipe(img)
print(f"B={B}: encode graph alone {timeit(enc):.3f} ms | decode loop alone {timeit(dec):.3f} ms | "
      f"overlapped step {timeit(both, 20):.3f} ms")
while pipe.outstanding():
    pipe.collect()
