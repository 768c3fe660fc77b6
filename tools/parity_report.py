#!/usr/bin/env python3
"""Caption-level parity of the reduced-precision modes against the fp32 mode (= the reference,
token for token) at the bench shape, with the error attributed per stage.

    python tools/parity_report.py [--images 256] [--variant eos] [--out gpurun_out/parity_report.json]

For every mode: beam-3 captions of `--images` synthetic images through CaptionPipeline (hipGraphs, two
decode lanes, batch 16), CIDEr-D against the fp32 captions (evaluation.caption_agreement), captions/s,
and the teacher-forced log-prob error on the fp32 captions of the first batch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from on_device_image_captioning_amd import weights as W  # noqa: E402
from on_device_image_captioning_amd.End_ExpansionNet_v2 import End_ExpansionNet_v2, make_drop_args  # noqa: E402
from on_device_image_captioning_amd.evaluation import caption_agreement  # noqa: E402
from on_device_image_captioning_amd.pipeline import CaptionPipeline  # noqa: E402

SOS, EOS = 79, 77
MODES = {"fp32": ("fp32", None), "bf16": ("bf16", None), "bf16_backbone_only": ("bf16", "fp32"),
         "bf16_encoder_only": ("fp32", "bf16")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--beam", type=int, default=3)
    ap.add_argument("--max-len", type=int, default=20)
    ap.add_argument("--variant", default="eos")
    ap.add_argument("--modes", default="fp32,bf16,bf16_backbone_only,bf16_encoder_only")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_report.json"))
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    g = W.FULL
    sd = W.synth_state_dict(g, variant=a.variant, eos_idx=EOS)
    m = End_ExpansionNet_v2(**g.model_kwargs(), output_word2idx={i: i for i in range(g.vocab_size)},
                            output_idx2word=list(range(g.vocab_size)), drop_args=make_drop_args(), rank=dev)
    m.load_state_dict(sd, strict=True)
    m.to(dev).eval()
    nb = (a.images + a.batch - 1) // a.batch
    batches = [W.synth_images(a.batch, g, seed=1000 + i).to(dev) for i in range(nb)]
    report = {"variant": a.variant, "images": nb * a.batch, "beam": a.beam, "max_len": a.max_len, "modes": {}}
    ref_caps = None
    ref_dec = None
    for name in a.modes.split(","):
        prec, encp = MODES[name]
        m.set_precision(prec, encp)
        pipe = CaptionPipeline(m, a.batch, a.beam, a.max_len, SOS, EOS)
        caps = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches:
            pipe.submit(b)
            while pipe.full():
                caps += pipe.collect()
        while pipe.outstanding():
            caps += pipe.collect()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        entry = {"captions_per_s": round(len(caps) / dt, 1)}
        if ref_caps is None:
            ref_caps = caps
            T = max(len(c) for c in caps[:a.batch])
            dec = torch.full((a.batch, T), EOS, dtype=torch.long)
            pads = []
            for i, c in enumerate(caps[:a.batch]):
                dec[i, :len(c)] = torch.tensor(c)
                pads.append(T - len(c))
            ref_dec = (dec.to(dev), pads)
        entry.update(caption_agreement(caps, ref_caps))
        # direct (un-pipelined) call of the first batch must equal the pipeline's captions
        toks, _ = m(enc_x=batches[0], enc_x_num_pads=[0] * a.batch, mode="beam_search", beam_size=a.beam,
                    how_many_outputs=1, beam_max_seq_len=a.max_len, sample_or_max="max", sos_idx=SOS, eos_idx=EOS)
        entry["pipeline_equals_direct_call"] = [t[0] for t in toks] == caps[:a.batch]
        mem = m.forward_enc(batches[0], [0] * a.batch)
        lp = m.forward_dec(mem, [0] * a.batch, ref_dec[0], ref_dec[1], apply_log_softmax=True)
        entry["_lp"] = lp.cpu()
        report["modes"][name] = entry
        del pipe
        torch.cuda.empty_cache()
    base = report["modes"].get("fp32", {}).get("_lp")
    for name, e in report["modes"].items():
        lp = e.pop("_lp")
        if base is not None:
            errs, margins = [], []
            for i in range(a.batch):
                n = ref_dec[0].shape[1] - ref_dec[1][i]
                errs.append(float((lp[i, :n] - base[i, :n]).abs().max()))
                top2 = torch.topk(base[i, :n], 2, -1).values
                margins += (top2[:, 0] - top2[:, 1]).tolist()
            e["teacher_forced_logprob_err_max"] = round(max(errs), 5)
            e["teacher_forced_logprob_err_mean_of_max"] = round(sum(errs) / len(errs), 5)
            if name == "fp32":
                ms = sorted(margins)
                report["fp32_top1_top2_margin_nat"] = {"median": round(ms[len(ms) // 2], 4), "p10": round(ms[len(ms) // 10], 4),
                                                        "min": round(ms[0], 5), "positions": len(ms)}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(report, open(a.out, "w"), indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
