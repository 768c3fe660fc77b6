#!/usr/bin/env python3
"""A/B of the two cross-attention step kernels INSIDE the real pipeline (the round-2 incident).

Round 2 met a software-pipelined P·V loop (`PF`: first V batch requested before the score phase, next batch under the
FMAs) that was bit-reproducible alone and gave ~1 wrong caption in 3,000 beside the encode graph; the cause was never
found and the evidence was caption-level only.  This tool catches EVERY occurrence at the source: in the captured step
graphs each cross-attention call site runs the shipped kernel (its output feeds the search, so captions stay right),
then the PF form on the same q / K / V into a buffer of its own, then a device-side bitwise compare; the operands of
the first mismatch (q, both outputs, the image's K and V) are kept on the device and analysed on the host:
which form deviates from the fp64 result, on which (row, channel), and which single replaced V value would explain it.

    bash tools/build_dbg.sh && python tools/xattn_ab.py [--sweeps 600] [--variant xavier]

Needs tools/_build/libodic_dbg.so (decoder_ops.hip built with -DODIC_XATTN_VARIANTS; never part of the product).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from on_device_image_captioning_amd import ops, weights as W  # noqa: E402
from on_device_image_captioning_amd.End_ExpansionNet_v2 import End_ExpansionNet_v2, make_drop_args  # noqa: E402
from on_device_image_captioning_amd.pipeline import CaptionPipeline  # noqa: E402

SOS, EOS = 79, 77


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweeps", type=int, default=600)
    ap.add_argument("--variant", default="xavier")
    ap.add_argument("--forms", default="1,2,3,4", help="PF forms compared against the plain kernel (1 = round 2's)")
    ap.add_argument("--stop-at-first", action="store_true")
    ap.add_argument("--precision", default="bf16", help="encode-pass precision: which kernels run beside the decode lanes")
    ap.add_argument("--lib", default=None, help="another build of libodic_hip.so for the PIPELINE's kernels (e.g. "
                                                "tools/_build/libodic_noprio.so = -DODIC_NO_ENCODE_PRIO)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "xattn_ab.json"))
    a = ap.parse_args()
    if a.lib:
        from on_device_image_captioning_amd import _hip
        assert _hip._lib is None
        _hip.LIB_PATH = os.path.abspath(a.lib)
    dbg = C.CDLL(os.path.join(ROOT, "tools", "_build", "libodic_dbg.so"))
    P, I32, I64 = C.c_void_p, C.c_int32, C.c_int64
    dbg.odic_dbg_cross_attn_step.argtypes = [C.c_int, P, I64, P, I64, I32, I32, P, P, P, I64, I32, I32, I32, I32, I32, P]
    dbg.odic_dbg_compare_snapshot.argtypes = [P, P, I64, P, I64, P, I64, I32, I32, I32, I32, I32, I32, I32, P, P, I32, P]

    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    g = W.FULL
    m = End_ExpansionNet_v2(**g.model_kwargs(), output_word2idx={i: i for i in range(g.vocab_size)},
                            output_idx2word=list(range(g.vocab_size)), drop_args=make_drop_args(), rank=dev)
    m.load_state_dict(W.synth_state_dict(g, variant=a.variant, eos_idx=EOS), strict=True)
    m.to(dev).eval().set_precision(a.precision)

    S, d, beams = 144, g.d_model, 3
    forms = [int(v) for v in a.forms.split(",")]
    state0 = [0, -1, 0, 0, 0, 0] + [0] * 10
    state = torch.tensor(state0, dtype=torch.int32, device=dev)
    snap = torch.zeros(3 * beams * d + 2 * S * d, dtype=torch.float32, device=dev)
    keep, sites = [], []
    orig = ops.cross_attn_step

    def patched(q, ldq, kv, ldkv, koff, voff, enc_len, row_valid, out, ldo, N, n_img, S_, d_, heads):
        orig(q, ldq, kv, ldkv, koff, voff, enc_len, row_valid, out, ldo, N, n_img, S_, d_, heads)
        site = len(sites)
        sites.append((koff // (2 * d_),))
        # both forms from the SAME translation unit (the product library's copy of the plain kernel is compiled
        # separately and differs from it by one fused multiply-add inside expf — tools/xattn_micro.py — so it cannot
        # serve as the bitwise reference)
        outs = {v: torch.empty_like(out) for v in [0] + forms}
        keep.extend(outs.values())
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for variant, dst in outs.items():
            rc = dbg.odic_dbg_cross_attn_step(variant, q.data_ptr(), ldq, kv.data_ptr(), ldkv, koff, voff, enc_len.data_ptr(),
                                              row_valid.data_ptr(), dst.data_ptr(), ldo, N, n_img, S_, d_, heads, st)
            assert rc == 0, rc
        for variant in forms:
            rc = dbg.odic_dbg_compare_snapshot(outs[0].data_ptr(), outs[variant].data_ptr(), ldo, q.data_ptr(), ldq,
                                               kv.data_ptr(), ldkv, koff, voff, N, n_img, S_, d_, site, state.data_ptr(),
                                               snap.data_ptr(), variant, st)
            assert rc == 0, rc

    ops.cross_attn_step = patched
    import on_device_image_captioning_amd.engine as eng_mod
    assert eng_mod.ops is ops
    pipe = CaptionPipeline(m, 16, beams, 20, SOS, EOS)
    torch.cuda.synchronize()
    n_sites_captured = len(sites)
    state.copy_(torch.tensor(state0, dtype=torch.int32))      # forget the warm-up / capture passes
    batches = [W.synth_images(16, g, seed=3000 + i).to(dev) for i in range(4)]
    t0 = time.perf_counter()
    done_sweeps = 0
    for s in range(a.sweeps):
        order = [0, 1, 2, 3] if s % 2 == 0 else [3, 2, 1, 0]
        for i in order:
            while pipe.full():
                pipe.collect()
            pipe.submit(batches[i])
        while pipe.outstanding():
            pipe.collect()
        done_sweeps = s + 1
        if a.stop_at_first and int(state[0].item()) != 0:
            break
        if (s + 1) % 100 == 0:
            print(f"[xattn_ab] {s + 1} sweeps, {int(state[5].item())} call sites compared, 0 mismatches, "
                  f"{time.perf_counter() - t0:.0f}s", flush=True)
    torch.cuda.synchronize()
    st = state.cpu().tolist()
    rep = {"variant": a.variant, "encode_precision": a.precision, "pipeline_library": a.lib or "libodic_hip.so", "sweeps": done_sweeps, "captions": done_sweeps * 64, "call_sites_in_graphs": n_sites_captured,
           "launch_pairs_compared": st[5], "mismatching_elements": st[0], "first_site": st[1], "row": st[2], "col": st[3],
           "snapshot_taken": bool(st[4]), "seconds": round(time.perf_counter() - t0, 1),
           "mismatching_elements_per_form": {str(v): st[6 + v] for v in forms},
           "forms": {"1": "round-2 software-pipelined P·V loop", "2": "1 + s_nop 7 after every FMA group",
                     "3": "1 with scalar FMAs (no v_pk_fma_f32)", "4": "1 with the key's probabilities read and waited for first"}}
    if st[0] and st[4]:
        rep["post_mortem"] = post_mortem(snap.cpu().double(), st[2], st[3], beams, S, d, g.num_heads)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rep, open(a.out, "w"), indent=1)
    print(json.dumps(rep, indent=1))


def post_mortem(snap, row, col, beams, S, d, heads):
    """fp64 attention of the kept operands; which form deviates; and for every deviating (beam, channel) the single
    replaced V value x (at key s) that would explain the error on ALL beams of the image at once:
    err[b] = p[b][s]·(x − V[s][c])."""
    dk = d // heads
    o = 0
    q = snap[o:o + beams * d].view(beams, d); o += beams * d
    A = snap[o:o + beams * d].view(beams, d); o += beams * d
    B = snap[o:o + beams * d].view(beams, d); o += beams * d
    K = snap[o:o + S * d].view(S, d); o += S * d
    V = snap[o:o + S * d].view(S, d)
    ref = torch.zeros(beams, d, dtype=torch.float64)
    probs = torch.zeros(beams, heads, S, dtype=torch.float64)
    for h in range(heads):
        sl = slice(h * dk, (h + 1) * dk)
        sc = (q[:, sl] @ K[:, sl].T) / (dk ** 0.5)
        p = torch.softmax(sc, -1)
        probs[:, h] = p
        ref[:, sl] = p @ V[:, sl]
    eA, eB = (A - ref).abs().max().item(), (B - ref).abs().max().item()
    out = {"max_err_shipped_kernel_vs_fp64": eA, "max_err_pipelined_kernel_vs_fp64": eB,
           "deviating_form": "pipelined (PF)" if eB > eA else "shipped", "first_row_in_image": row % beams, "first_col": col}
    bad = ((A != B).any(0)).nonzero().flatten().tolist()
    out["channels_that_differ"] = bad[:32]
    wrong = B if eB > eA else A
    expl = []
    for c in bad[:8]:
        h = c // dk
        err = wrong[:, c] - ref[:, c]
        best = None
        for s in range(S):
            p = probs[:, h, s]
            if p[0].abs() < 1e-12:
                continue
            x = err[0] / p[0] + V[s, c]
            resid = (err - p * (x - V[s, c])).abs().max().item()
            if best is None or resid < best[0]:
                best = (resid, s, x.item())
        resid, s, x = best
        # where does x occur in the image's V / K?
        hit = ((V - x).abs() < 1e-6 * max(1.0, abs(x))).nonzero().tolist()[:4]
        hitk = ((K - x).abs() < 1e-6 * max(1.0, abs(x))).nonzero().tolist()[:4]
        expl.append({"channel": c, "err_per_beam": err.tolist(), "best_key": s, "replaced_value": x, "true_value": V[s, c].item(),
                     "residual": resid, "same_value_found_in_V_at": hit, "in_K_at": hitk})
    out["single_value_explanations"] = expl
    # hypothesis "ONE probability operand was lost": for the deviating beam b, a single key s whose whole term
    # p[b][s]·V[s][c] is missing (or scaled by f) on every deviating channel at once: err[c] = (f − 1)·p[b][s]·V[s][c]
    dev_rows = ((wrong - ref).abs() > 1e-5).any(1).nonzero().flatten().tolist()
    drops = []
    for b in dev_rows:
        cs = [c for c in bad if abs(float(wrong[b, c] - ref[b, c])) > 1e-6]
        if not cs:
            continue
        h = cs[0] // dk
        errv = torch.stack([wrong[b, c] - ref[b, c] for c in cs])
        best = None
        for s_ in range(S):
            t = torch.stack([probs[b, h, s_] * V[s_, c] for c in cs])
            f = float((errv @ t) / (t @ t + 1e-300))                   # least-squares scale of the term
            resid = float((errv - f * t).abs().max())
            if best is None or resid < best[0]:
                best = (resid, s_, f)
        drops.append({"beam": b, "channels": [cs[0], cs[-1]], "n_channels": len(cs), "key": best[1], "key_group_g": best[1] % 4,
                      "slot_i_in_batch": (best[1] // 4) % 12, "term_scale_minus_1": best[2], "max_residual": best[0],
                      "max_err": float(errv.abs().max())})
    out["lost_probability_operand_fit"] = drops
    return out


if __name__ == "__main__":
    main()
