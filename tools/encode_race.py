#!/usr/bin/env python3
"""Is the encode graph bit-reproducible beside the decode lanes?  The K/V it produces for one fixed batch, replay after
replay, compared on the device with the first replay's — alone and with both decode lanes searching continuously."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from on_device_image_captioning_amd import weights as W
from on_device_image_captioning_amd.pipeline import CaptionPipeline

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
model, sd, g = bench.build_model(dev, "bf16", "e2e16")
pipe = CaptionPipeline(model, 16, 3, 20, 79, 77)
img = W.synth_images(16, g, seed=3001).to(dev)
pipe(img)
while pipe.outstanding():
    pipe.collect()
torch.cuda.synchronize()
pipe.imgs[0].copy_(img)
with torch.cuda.stream(pipe.s_enc):
    pipe.g_enc.replay()
torch.cuda.synchronize()
ref = [t.clone() for t in pipe.kv_stages[0]] if isinstance(pipe.kv_stages[0], (list, tuple)) else pipe.kv_stages[0].clone()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for busy in (False, True):
    bad = torch.zeros(1, dtype=torch.int64, device=dev)
    worst = torch.zeros(1, device=dev)
    for it in range(n):
        if busy:
            for l in range(pipe.D):
                with torch.cuda.stream(pipe.s_dec[l]):
                    pipe._reset(l)
                    pipe.replay_search(l)
        with torch.cuda.stream(pipe.s_enc):
            pipe.g_enc.replay()
            cur = pipe.kv_stages[0]
            d = (cur.float() - ref.float()).abs().max()
            bad += (d > 0).long()
            worst.copy_(torch.maximum(worst, d.reshape(1)))
    torch.cuda.synchronize()
    print(f"decode lanes busy={busy}: {int(bad)} of {n} encode replays differ from the first (max |diff| {float(worst):.3e})", flush=True)
