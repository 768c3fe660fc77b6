#!/usr/bin/env python3
"""odic_logsoftmax_topk / odic_beam_search_step timing against k (beams), N (rows) and V, back to back inside one
hipGraph (the search-step kernel re-armed by odic_beam_reset before every call)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from on_device_image_captioning_amd import _hip, ops

s = torch.cuda.Stream()


def graph_time(fn, reps=32):
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(reps):
            fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        g.replay()
        st.record()
        for _ in range(5):
            g.replay()
        en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) * 1e3 / (5 * reps)


def beam_state(n_img, k, T):
    N = n_img * k
    z = lambda *sh, dt=torch.float32: torch.zeros(*sh, dtype=dt, device="cuda")      # noqa: E731
    t = dict(tokens=z(n_img, k, T, dt=torch.int64), logprobs=z(n_img, k, T), anc=z(N, T, dt=torch.int32), cumul=z(N),
             n_elem=z(N, dt=torch.int32), has_eos=z(N, dt=torch.int32), row_valid=z(N, dt=torch.int32),
             next_tok=z(N, dt=torch.int64), pos=z(1, dt=torch.int32), done=z(1, dt=torch.int32), ctr=z(1, dt=torch.int32))
    st = _hip.BeamState(*(t[n].data_ptr() for n in ("tokens", "logprobs", "anc", "cumul", "n_elem", "has_eos",
                                                    "row_valid", "next_tok", "pos", "done", "ctr")))
    return t, st


def main():
    n_img, T, d = 16, 74, 512
    for V in (10000, 2000):
        for k in (1, 2, 3, 5, 8):
            N = n_img * k
            lg = torch.randn(N, V, device="cuda")
            cv, ci = torch.zeros(N, k, device="cuda"), torch.zeros(N, k, dtype=torch.int32, device="cuda")
            t_topk = graph_time(lambda: ops.logsoftmax_topk(lg, V, None, 0, cv, ci, N, V, k))
            t, st = beam_state(n_img, k, T)
            embed, ptab, y = torch.randn(V, d, device="cuda"), torch.randn(T, d, device="cuda"), torch.zeros(N, d, device="cuda")
            emb = ops.embed_args(embed, ptab, y, d, d, 1.0)
            t_reset = graph_time(lambda: ops.beam_reset(st, n_img, k, T, 3, emb=emb))

            def two_steps():        # reset, step at pos 0 (one row per image), step at pos 1 (k rows per image)
                ops.beam_reset(st, n_img, k, T, 3, emb=emb)
                ops.beam_search_step(lg, V, V, st, n_img, k, T, 4, emb=emb)
                ops.beam_search_step(lg, V, V, st, n_img, k, T, 4, emb=emb)

            def one_step():
                ops.beam_reset(st, n_img, k, T, 3, emb=emb)
                ops.beam_search_step(lg, V, V, st, n_img, k, T, 4, emb=emb)

            t1, t2 = graph_time(one_step), graph_time(two_steps)

            def unfused():
                ops.beam_reset(st, n_img, k, T, 3)
                ops.logsoftmax_topk(lg, V, None, 0, cv, ci, N, V, k)
                ops.beam_step(cv, ci, st, n_img, k, T, 4)
                ops.logsoftmax_topk(lg, V, None, 0, cv, ci, N, V, k)
                ops.beam_step(cv, ci, st, n_img, k, T, 4)
            t_un = graph_time(unfused)
            print(f"V={V:5d} k={k}: logsoftmax_topk ({N} rows) {t_topk:6.2f} us | reset {t_reset:5.2f} | fused step at pos 0 "
                  f"{t1 - t_reset:6.2f}, at pos 1 {t2 - t1:6.2f} | unfused topk+beam_step at pos 1 "
                  f"{t_un - t_reset - (t_topk + 0) - 0:6.2f} (minus one topk: two launches + a step-0 beam_step)", flush=True)


if __name__ == "__main__":
    main()
