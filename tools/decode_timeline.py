#!/usr/bin/env python3
"""Timeline of ONE decode lane running alone (no encoder beside it): per launch of a decoder step its duration and
the gap to its predecessor, from a rocprofv3 kernel trace.

    # on the GPU box
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/decode_timeline.py run
    python3 tools/decode_timeline.py report $OUT/**/*kernel_trace.csv
"""
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch

    import bench
    from on_device_image_captioning_amd import weights as W
    from on_device_image_captioning_amd.pipeline import CaptionPipeline

    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    wl = os.environ.get("ODIC_WL", "e2e16")
    w = bench.WORKLOADS[wl]
    model, sd, g = bench.build_model(dev, w["precision"], wl)
    pipe = CaptionPipeline(model, w["batch"], w["beam"], w["max_len"], 79, 77)
    if wl == "features48":
        img = torch.randn(w["batch"], 144, g.final_swin_dim, device=dev)
    else:
        img = W.synth_images(w["batch"], g).to(dev)
    pipe(img)
    while pipe.outstanding():
        pipe.collect()
    torch.cuda.synchronize()
    n = int(os.environ.get("ODIC_SEARCHES", "3"))
    with torch.cuda.stream(pipe.s_dec[0]):
        for _ in range(n):
            pipe._reset(0)
            pipe.replay_search(0)
    torch.cuda.synchronize()


def short(name):
    m = re.search(r"(?:\)::)?([A-Za-z_0-9]+)(?:<([^>]*)>)?\(", name)
    return (m.group(1) + (f"<{m.group(2)}>" if m and m.group(2) else "")) if m else name[:40]


def report(path):
    rows = list(csv.DictReader(open(path)))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                  int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0))
                 for r in rows), key=lambda x: x[0])
    # the last searches: find beam_reset launches
    resets = [i for i, k in enumerate(ks) if "beam_reset" in k[2]]
    if len(resets) < 2:
        print("no decode chain found")
        return
    a, b = resets[-2], resets[-1]
    chain = ks[a:b]                                   # one whole search (reset + steps)
    total = (chain[-1][1] - chain[0][0]) / 1e3
    busy = sum(k[1] - k[0] for k in chain) / 1e3
    print(f"one search alone: {len(chain)} launches, {total:.1f} us wall, {busy:.1f} us inside kernels, "
          f"{total - busy:.1f} us of gaps ({(total - busy) / max(1, len(chain) - 1):.2f} us per boundary)")
    gaps = sorted(((chain[i][0] - chain[i - 1][1]) / 1e3, i) for i in range(1, len(chain)))[-5:]
    print("largest gaps (us, launch index): " + ", ".join(f"{g:.1f} @ {i}" for g, i in reversed(gaps)))
    steps = [i + 1 for i, k in enumerate(chain[:-1]) if "beam_search_step" in k[2] or "beam_step_kernel" in k[2]]
    steps = [1] + steps                                # (launch 0 is the reset; a step ends with the beam update)
    which = min(int(os.environ.get("ODIC_STEP", len(steps) // 2)), len(steps) - 2)
    mid = steps[which]
    nxt = steps[which + 1]
    print(f"step {which} of the search ({nxt - mid} launches, {(chain[nxt][0] - chain[mid][0]) / 1e3:.1f} us):")
    prev_end = chain[mid - 1][1]
    for k in chain[mid:nxt]:
        print(f"  gap {(k[0] - prev_end) / 1e3:6.2f} us | run {(k[1] - k[0]) / 1e3:6.2f} us | grid {k[3] // max(1, k[4]):5d} x {k[4]:4d} | {short(k[2])}")
        prev_end = k[1]
    # aggregate by kernel over the whole search
    agg = {}
    for i, k in enumerate(chain):
        s = short(k[2])
        d = agg.setdefault(s, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += (k[1] - k[0]) / 1e3
        if i:
            d[2] += (k[0] - chain[i - 1][1]) / 1e3
    print("per kernel over the search: launches, total run us, total gap-before us")
    for s, d in sorted(agg.items(), key=lambda x: -x[1][1] - x[1][2]):
        print(f"  {d[0]:4d} {d[1]:8.1f} {d[2]:8.1f}  {s}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
