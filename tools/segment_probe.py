#!/usr/bin/env python3
"""Which part of the encode pass pays for the decode lanes?  The pass is captured as five graphs
(Swin stage 0, 1, 2, 3 + expansion encoder / K/V projection); each is replayed back to back alone and
beside two continuously replaying decode lanes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from on_device_image_captioning_amd import ops, weights as W
from on_device_image_captioning_amd.pipeline import CaptionPipeline

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
model, sd, g = bench.build_model(dev, "bf16", "e2e16")
pipe = CaptionPipeline(model, 16, 3, 20, 79, 77, decode_lanes=2)
swin, cap = pipe.swin, pipe.cap
B = 16
img = W.synth_images(B, g).to(dev)
s_enc = pipe.s_enc


def run_stage(x, s):
    blocks, down = swin.stages[s]
    res, C_, heads, ws = g.stage_res(s), g.stage_dim(s), g.swin_num_heads[s], g.stage_window(s)
    x = x.view(B * res * res, C_)
    for w in blocks:
        xn = ops.layernorm(x, w["n1w"], w["n1b"], out_dtype=swin.cdt)
        qkv = ops.gemm(xn, w["qkv_w"], w["qkv_b"])
        att = ops.window_attention(qkv, w["table"], B, res, C_, heads, ws, w["shift"], bias_shifted_prescaled=w["dense"])
        ops.gemm(att, w["proj_w"], w["proj_b"], residual=x, out=x)
        xn = ops.layernorm(x, w["n2w"], w["n2b"], out_dtype=swin.cdt)
        h = ops.gemm(xn, w["fc1_w"], w["fc1_b"], act=ops.ACT_GELU)
        ops.gemm(h, w["fc2_w"], w["fc2_b"], residual=x, out=x)
    if down is not None:
        xm = ops.patch_merge_layernorm(x, down["nw"], down["nb"], B, res, C_, out_dtype=swin.cdt)
        x = ops.gemm(xm.view(-1, 4 * C_), down["red_w"], out_dtype=torch.float32)
    return x


graphs, names = [], []
with torch.cuda.stream(s_enc):
    x = ops.patch_embed(img, swin.pe_w, swin.pe_b, swin.pe_g, swin.pe_beta, g.swin_patch_size)
    xs = [x]
    for s in range(4):
        xs.append(run_stage(xs[-1].clone(), s))
    feats = ops.layernorm(xs[4], swin.fn_w, swin.fn_b, out_dtype=cap.cdt).view(B, 144, -1)
torch.cuda.synchronize()
keep = []                                        # inputs captured by address must outlive the graphs
for s in range(4):
    inp = xs[s].clone()
    keep.append(inp)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s_enc):
        run_stage(inp, s)
    graphs.append(gr); names.append(f"swin stage {s}")
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=s_enc):
    _, mem16 = cap.encode(feats, pipe.enc_lens[0], want_bf16_mem=True)
    cap.project_kv(mem16, out=pipe.kv_stage[0])
graphs.append(gr); names.append("expansion encoder + K/V")
torch.cuda.synchronize()


def time_seg(gr, n, with_decode, keep_src=None):
    torch.cuda.synchronize()
    if with_decode:
        for l in range(2):
            with torch.cuda.stream(pipe.s_dec[l]):
                for i in range(8):
                    pipe._reset(l)
                    pipe.replay_search(l)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s_enc):
        e0.record()
        for i in range(n):
            if keep_src is not None and i % 8 == 0:
                keep_src[0].copy_(keep_src[1])           # the stage works in place: do not let the values run away
            gr.replay()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot_a = tot_b = 0.0
for k, (gr, nm) in enumerate(zip(graphs, names)):
    src = (keep[k], xs[k]) if k < 4 else None
    alone = time_seg(gr, 12, False, src)
    n = max(4, int(60.0 / alone))               # ~60 ms of encode-side work, inside the decode lanes' ~100 ms
    busy = time_seg(gr, n, True, src)
    tot_a += alone; tot_b += busy
    print(f"{nm:28s} alone {alone:7.3f} ms | beside 2 decode lanes {busy:7.3f} ms  (+{busy - alone:.3f}, x{busy / alone:.2f})")
print(f"{'sum':28s} alone {tot_a:7.3f} ms | beside 2 decode lanes {tot_b:7.3f} ms")
