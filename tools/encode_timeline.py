#!/usr/bin/env python3
"""Timeline of the encode graph running alone: every launch of one encode pass with its duration and the gap to its
predecessor (rocprofv3 kernel trace), plus totals per kernel.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/encode_timeline.py run
    python3 tools/encode_timeline.py report $OUT/**/*kernel_trace.csv
"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.decode_timeline import short  # noqa: E402


def run():
    import torch

    import bench
    from on_device_image_captioning_amd import weights as W
    from on_device_image_captioning_amd.pipeline import CaptionPipeline

    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    model, sd, g = bench.build_model(dev, "bf16", "e2e16")
    pipe = CaptionPipeline(model, 16, 3, 20, 79, 77)
    img = W.synth_images(16, g).to(dev)
    pipe(img)
    while pipe.outstanding():
        pipe.collect()
    torch.cuda.synchronize()
    with torch.cuda.stream(pipe.s_enc):
        for _ in range(4):
            pipe.g_enc.replay()
    torch.cuda.synchronize()


def report(path):
    rows = list(csv.DictReader(open(path)))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
    marks = [i for i, k in enumerate(ks) if "patch_embed" in k[2]]
    a, b = marks[-2], marks[-1]
    chain = ks[a:b]
    total = (chain[-1][1] - chain[0][0]) / 1e3
    busy = sum(k[1] - k[0] for k in chain) / 1e3
    print(f"one encode pass alone: {len(chain)} launches, {total:.1f} us wall, {busy:.1f} us inside kernels, "
          f"{total - busy:.1f} us of gaps")
    prev = chain[0][0]
    agg = {}
    for i, k in enumerate(chain):
        s = short(k[2])
        d = agg.setdefault(s, [0, 0.0])
        d[0] += 1
        d[1] += (k[1] - k[0]) / 1e3
        print(f"  {i:3d} gap {(k[0] - prev) / 1e3:6.2f} | run {(k[1] - k[0]) / 1e3:7.2f} | {s[:110]}")
        prev = k[1]
    print("per kernel: launches, total us")
    for s, d in sorted(agg.items(), key=lambda x: -x[1][1]):
        print(f"  {d[0]:4d} {d[1]:8.1f}  {s[:120]}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
