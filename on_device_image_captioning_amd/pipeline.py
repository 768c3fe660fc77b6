"""Fixed-shape captioning pipeline: hipGraph-captured encoder pass + decoder step, and the image
shard runner for multi-GPU evaluation.

`CaptionPipeline` serves repeated batches of one shape (B images, beam k, max length T):
  graph 1  images → Swin-L → expansion encoder → per-layer cross-attention K/V      (one replay)
  graph 2  one decoder position + log-softmax/top-k + on-device beam bookkeeping    (T-1 replays)
Both graphs are captured with `torch.cuda.graph` from the very same Python sequencing that runs
eagerly elsewhere (every C-ABI entry point is capture-legal: no allocation, no sync), so replays
carry no per-launch host cost and the host never waits inside a caption.

`shard_indices` / `gather_captions` implement the image-parallel evaluation the north star asks
for: contiguous shards, one process per GPU, and ONE collective (all_gather of fixed-shape token
and length tensors) at the end.  The reference has no counterpart (test.py:300-316 makes every rank
evaluate everything).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import os

import torch

from . import ops
from .captioning_model import CaptioningModel


class CaptionPipeline:
    """Software pipeline over consecutive batches on HIP streams:

        s_enc      encode graph of batch i+1          (compute-bound, fills the chip)
        s_dec[0]   decode-step graph x(T-1), group j   \\  ~40 tiny dependent kernels per step: latency-bound,
        s_dec[1]   decode-step graph x(T-1), group j-1 //  so two chains in flight hide each other's waits

    A decode GROUP is `decode_group` consecutive batches searched together (G·B images, G·B·k rows per
    kernel): the step kernels are latency-bound at 48 rows, so G batches share one chain of launches.
    Images never interact inside the search, so the captions are those of the per-batch search.

    `submit()` only enqueues; `collect()` returns the captions of the oldest outstanding batch (and
    launches a partially filled group if that batch is waiting in one).  Every batch is still
    processed exactly as in the reference (encoder and search per image); only the scheduling
    across batches is overlapped."""

    def __init__(self, model: CaptioningModel, batch: int, beam_size: int, max_seq_len: int, sos_idx: int,
                 eos_idx: int, use_graphs: bool = True, done_poll: int = 0, decode_lanes: int = 2,
                 streams=None, decode_group: int = 1, encode_lanes: int = 1, feat_len: int = 144,
                 decode_cus: Optional[int] = None, keep_scores: bool = False):
        """done_poll = 0: never look at the `done` flag (fixed work per batch — benchmark mode with
        weights that never emit EOS); n > 0: host checks every n steps and stops early.
        `model` is an End_ExpansionNet_v2 (inputs: images [B,3,H,W]) or a features-only ExpansionNet_v2
        (inputs: features [B, feat_len, F] + per-sample trailing pad counts, ExpansionNet_v2.py:50-70)."""
        # keep_scores (verification runs): the best beam's per-token log-probs travel with its tokens, so that a
        # pipelined search can be compared with the direct call bit for bit (collect_scored())
        self.keep_scores = keep_scores
        self.model, self.B, self.k = model, batch, beam_size
        self.steps = max(1, max_seq_len - 1)
        self.T = self.steps + 1
        self.sos, self.eos = sos_idx, eos_idx
        self.done_poll = done_poll
        self.D = max(1, decode_lanes)
        self.G = max(1, decode_group)
        self.E = max(1, encode_lanes)                              # encode graphs that may be in flight together
        self.NB = self.G * batch                                   # images per search launch
        self.RG = self.D + 1                                       # result ring, in groups
        self.R = self.RG * self.G                                  # batches that may be outstanding (upper bound)
        # members: one (backbone engine, captioner engine) pair per model; an EsembleCaptioningModel contributes
        # all of its models (every member encodes and decodes on its own, ONE shared beam state per lane,
        # per-step distribution = log(mean softmax) — ensemble_captioning_model.py:66-83)
        models = list(getattr(model, "models_list", None) or [model])
        self.members = [m._engines() if hasattr(m, "swin_transf") else (None, m._captioner_engine()) for m in models]
        self.M = len(self.members)
        swin, cap = self.members[0]
        if any((sw is None) != (swin is None) or c.g != cap.g for sw, c in self.members):
            raise RuntimeError("ensemble members must share one geometry and input kind")
        self.swin, self.cap = swin, cap
        g, dv = cap.g, cap.device
        self.device = dv
        if swin is not None:
            in_shape = (batch, g.swin_in_chans, g.swin_img_size, g.swin_img_size)
            S = g.stage_res(len(g.swin_depths) - 1) ** 2
        else:
            in_shape = (batch, feat_len, g.final_swin_dim)
            S = feat_len
        self.S = S
        self.imgs = [torch.zeros(*in_shape, dtype=torch.float32, device=dv) for _ in range(self.E)]
        # encoder lengths: one buffer per encode lane (read by its graph) + one per decode lane (its group)
        self.enc_lens = [torch.full((batch,), S, dtype=torch.int32, device=dv) for _ in range(self.E)]
        self.enc_len_grps = [torch.full((self.NB,), S, dtype=torch.int32, device=dv) for _ in range(self.D)]
        self.host_len_in = [torch.full((batch,), S, dtype=torch.int32).pin_memory() for _ in range(self.E)]
        nkv = 2 * g.N_dec * g.d_model
        # K/V of every member: [member] stacked in the leading dim, written by the encode graphs
        self.kv_stages = [torch.empty(self.M, batch, S, nkv, dtype=torch.float32, device=dv) for _ in range(self.E)]
        self.kv = [torch.empty(self.M, self.NB, S, nkv, dtype=torch.float32, device=dv) for _ in range(self.D)]
        self.member_states = []                                    # [lane][member]; member 0 owns the beam state
        for l in range(self.D):
            sts = [c.new_state(self.NB, beam_size, self.T, self.kv[l][mi], self.enc_len_grps[l])
                   for mi, (_, c) in enumerate(self.members)]
            for st in sts[1:]:
                st.anc, st.row_valid, st.next_tok, st.pos = sts[0].anc, sts[0].row_valid, sts[0].next_tok, sts[0].pos
            self.member_states.append(sts)
        self.states = [sts[0] for sts in self.member_states]
        self.avg_logp = [torch.empty(self.NB * beam_size, g.vocab_size, dtype=torch.float32, device=dv)
                         for _ in range(self.D)] if self.M > 1 else None
        self.order = [torch.empty(self.NB, beam_size, dtype=torch.int32, device=dv) for _ in range(self.D)]
        self.score = [torch.empty(self.NB, beam_size, dtype=torch.float32, device=dv) for _ in range(self.D)]
        if streams is not None:                                    # ([encode streams], [decode streams]) supplied by the caller
            self.s_encs = list(streams[0]) if isinstance(streams[0], (list, tuple)) else [streams[0]]
            self.s_dec = list(streams[1])
        elif decode_cus:                                           # the decode lanes on compute units of their own
            from .cu_streams import split_streams
            self.s_encs, self.s_dec = split_streams(dv, decode_cus, self.E, self.D)
        else:
            self.s_encs = [torch.cuda.Stream(device=dv) for _ in range(self.E)]
            self.s_dec = [torch.cuda.Stream(device=dv) for _ in range(self.D)]
        self.ev_enc = [torch.cuda.Event() for _ in range(self.E)]
        self.ev_kv_taken = [torch.cuda.Event() for _ in range(self.E)]
        self.ev_len_up = [torch.cuda.Event() for _ in range(self.E)]      # pinned length buffer free to rewrite
        for ev in self.ev_kv_taken + self.ev_len_up:
            ev.record()
        # results ring (device, one slot per group) + pinned host mirrors
        self.out_tok = [torch.zeros(self.NB, self.T, dtype=torch.int32, device=dv) for _ in range(self.RG)]
        self.out_len = [torch.zeros(self.NB, dtype=torch.int32, device=dv) for _ in range(self.RG)]
        self.host_tok = [torch.zeros(self.NB, self.T, dtype=torch.int32).pin_memory() for _ in range(self.RG)]
        self.host_len = [torch.zeros(self.NB, dtype=torch.int32).pin_memory() for _ in range(self.RG)]
        if keep_scores:
            self.host_lp = [torch.zeros(self.NB, self.T, dtype=torch.float32).pin_memory() for _ in range(self.RG)]
            self._rows = torch.arange(self.NB, device=dv)
        self.ev_done = [torch.cuda.Event() for _ in range(self.RG)]
        self.ev_res_free = [torch.cuda.Event() for _ in range(self.RG)]   # device-side consumers are done with the slot
        for ev in self.ev_res_free:
            ev.record()
        self._submitted = self._collected = 0
        self._gi = 0                 # index of the group being filled
        self._gfill = 0              # batches already staged into it
        self._where: List[Tuple[int, int]] = []    # outstanding batches, oldest first: (group index, position in group)
        self.g_encs: List[Optional[torch.cuda.CUDAGraph]] = [None] * self.E
        self.g_step: List[Optional[torch.cuda.CUDAGraph]] = [None] * self.D
        # steps per graph launch: the whole search when the host never polls `done`, else one poll interval; every
        # step reads its position from device memory, so a graph of `chunk` steps serves any part of the search (a
        # graph-to-graph boundary costs 8.5 us on the decode stream, a node-to-node boundary nothing measurable)
        self.chunk = self.steps if not done_poll else max(1, min(done_poll, self.steps))
        if os.environ.get("ODIC_STEP_CHUNK"):                    # measurement switch: steps per graph launch
            self.chunk = max(1, min(int(os.environ["ODIC_STEP_CHUNK"]), self.steps))
        self.g_tail: List[Optional[torch.cuda.CUDAGraph]] = [None] * self.D     # steps % chunk
        if use_graphs:
            self._capture()

    # compatibility with single-lane callers (tests, tools)
    @property
    def state(self):
        return self.states[0]

    @property
    def s_enc(self):
        return self.s_encs[0]

    @property
    def g_enc(self):
        return self.g_encs[0]

    @property
    def img(self):
        return self.imgs[0]

    @property
    def kv_stage(self):
        return self.kv_stages[0]

    # -- the captured regions -------------------------------------------------------------------
    def _encode(self, e: int = 0) -> None:
        for mi, (swin, cap) in enumerate(self.members):
            feats = swin.forward(self.imgs[e], out_dtype=cap.cdt) if swin is not None else self.imgs[e]
            if cap.cdt != torch.float32:                          # bf16 / split-fp16 encoder: K/V from the rounded memory
                _, mem16 = cap.encode(feats, self.enc_lens[e], want_bf16_mem=True)
                cap.project_kv(mem16, out=self.kv_stages[e][mi])
            else:
                cap.project_kv(cap.encode(feats, self.enc_lens[e]), out=self.kv_stages[e][mi])

    def _step(self, lane: int) -> None:
        if self.M == 1:
            self.cap.beam_step(self.states[lane], self.eos)
            return
        sts = self.member_states[lane]
        lead = sts[0]
        for (_, cap), st in zip(self.members, sts):
            cap.step_logits(st)                                   # private caches, shared next_tok / pos / ancestors
        ops.ensemble_logprobs([st.logits for st in sts], self.avg_logp[lane])
        ops.topk_rows(self.avg_logp[lane], lead.cand_val, lead.cand_idx, lead.beams)
        ops.beam_step(lead.cand_val, lead.cand_idx, lead.beam_state, lead.n_img, lead.beams, lead.T, self.eos)

    def replay_search(self, lane: int) -> None:
        """All steps of one search on the current stream, no polling (measurement tools)."""
        t = 0
        while t < self.steps:
            if self.steps - t >= self.chunk:
                self.g_step[lane].replay()
                t += self.chunk
            else:
                self.g_tail[lane].replay()
                t = self.steps

    def _reset(self, lane: int) -> None:
        st = self.states[lane]
        # single model: the start token's embedding is part of the reset, every later input row part of the step's
        # last launch (CaptionerEngine.beam_step); ensemble members embed for themselves in step_logits
        ops.beam_reset(st.beam_state, st.n_img, st.beams, st.T, self.sos, emb=st.emb if self.M == 1 else None)

    def _capture(self) -> None:
        torch.cuda.synchronize()
        with torch.cuda.stream(self.s_encs[0]), ops.autotune():  # warm-up: allocator, code load, tile autotune
            self._encode(0)
        for lane in range(self.D):
            with torch.cuda.stream(self.s_dec[lane]):
                self.s_dec[lane].wait_stream(self.s_encs[0])
                for p in range(self.G):
                    self.kv[lane][:, p * self.B:(p + 1) * self.B].copy_(self.kv_stages[0])
                self._reset(lane)
                self._step(lane)
        torch.cuda.synchronize()
        for e in range(self.E):                                  # one graph (and private activation pool) per encode lane
            self.g_encs[e] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_encs[e], stream=self.s_encs[e]):
                self._encode(e)
        for lane in range(self.D):
            self._reset(lane)
            self.g_step[lane] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_step[lane], stream=self.s_dec[lane]):
                for _ in range(self.chunk):
                    self._step(lane)
            if self.steps % self.chunk:
                self._reset(lane)
                self.g_tail[lane] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_tail[lane], stream=self.s_dec[lane]):
                    for _ in range(self.steps % self.chunk):
                        self._step(lane)
        torch.cuda.synchronize()

    # -- public ---------------------------------------------------------------------------------
    def submit(self, images: torch.Tensor, enc_num_pads: Optional[Sequence[int]] = None) -> None:
        """Enqueue one batch; never blocks the host (unless done_poll > 0).  `enc_num_pads` (features-only
        model): trailing padded positions per sample, as the reference's enc_x_num_pads."""
        if enc_num_pads is not None and self.swin is not None and any(int(p) for p in enc_num_pads):
            raise AssertionError("End to End case have no padding")
        if self.full():
            raise RuntimeError(f"result ring full ({self.outstanding()} batches outstanding); call collect() first")
        gi, gpos = self._gi, self._gfill
        lane = gi % self.D
        e = self._submitted % self.E
        cur = torch.cuda.current_stream()
        se = self.s_encs[e]
        with torch.cuda.stream(se):
            se.wait_stream(cur)                                  # `images` may have been produced there
            se.wait_event(self.ev_kv_taken[e])                   # this lane's previous K/V hand-off finished
            if images.is_cuda:
                images.record_stream(se)                         # the caller may drop `images` right after submit()
            self.imgs[e].copy_(images, non_blocking=True)
            if self.swin is None:                                # per-sample encoder lengths ride along
                hl = self.host_len_in[e]
                pads = [0] * self.B if enc_num_pads is None else [int(p) for p in enc_num_pads]
                if len(pads) != self.B:
                    raise RuntimeError(f"expected {self.B} pad counts, got {len(pads)}")
                self._len_host_free(e)
                for i, pz in enumerate(pads):
                    hl[i] = self.S - pz
                self.enc_lens[e].copy_(hl, non_blocking=True)
                self.ev_len_up[e].record()
            if self.g_encs[e] is not None:
                self.g_encs[e].replay()
            else:
                self._encode(e)
            self.ev_enc[e].record()
        sd = self.s_dec[lane]
        with torch.cuda.stream(sd):
            sd.wait_event(self.ev_enc[e])
            self._hand_over(self.kv_stages[e], self.kv[lane][:, gpos * self.B:(gpos + 1) * self.B])
            if self.swin is None:
                self.enc_len_grps[lane][gpos * self.B:(gpos + 1) * self.B].copy_(self.enc_lens[e])
            self.ev_kv_taken[e].record()
        self._where.append((gi, gpos))
        self._submitted += 1
        self._gfill += 1
        if self._gfill == self.G:
            self._launch_group()

    @staticmethod
    def _hand_over(src: torch.Tensor, dst: torch.Tensor) -> None:
        """K/V of a batch from the encode stream's staging buffer into the lane's buffer (current stream)."""
        if os.environ.get("ODIC_HANDOVER", "kernel") == "runtime" or not (src.is_contiguous() and dst.is_contiguous()):
            dst.copy_(src)                                       # hipMemcpyAsync (blit kernel of the runtime)
        else:
            ops.copy(src, dst)

    def _len_host_free(self, e: int) -> None:
        self.ev_len_up[e].synchronize()                          # the previous upload from this pinned buffer is done

    def flush(self) -> None:
        """Search a partially filled group now (its empty positions repeat the last staged batch)."""
        if self._gfill:
            self._launch_group()

    def _launch_group(self) -> None:
        gi, filled = self._gi, self._gfill
        lane, gslot = gi % self.D, gi % self.RG
        st = self.states[lane]
        with torch.cuda.stream(self.s_dec[lane]):
            for p in range(filled, self.G):
                self.kv[lane][:, p * self.B:(p + 1) * self.B].copy_(self.kv[lane][:, (filled - 1) * self.B:filled * self.B])
                if self.swin is None:
                    self.enc_len_grps[lane][p * self.B:(p + 1) * self.B].copy_(
                        self.enc_len_grps[lane][(filled - 1) * self.B:filled * self.B])
            self._reset(lane)
            if self.g_step[lane] is not None:
                t = 0
                while t < self.steps:
                    if self.steps - t >= self.chunk:
                        self.g_step[lane].replay()
                        t += self.chunk
                    else:
                        self.g_tail[lane].replay()
                        t = self.steps
                    if self.done_poll and t < self.steps and int(st.done.item()):
                        break
            else:
                for t in range(self.steps):
                    ops.set_step_hint(t)                         # prices the step's cache reads in a profile pass
                    self._step(lane)
                    ops.set_step_hint(None)
                    if self.done_poll and t >= 1 and (t + 1) % self.done_poll == 0 and t + 1 < self.steps \
                            and int(st.done.item()):
                        break
            self.s_dec[lane].wait_event(self.ev_res_free[gslot])   # collect_device() readers of the slot's last use
            ops.beam_finalize_best(st.beam_state, self.order[lane], self.score[lane], self.out_tok[gslot],
                                   self.out_len[gslot], self.NB, self.k, self.T, self.eos)
            self.host_tok[gslot].copy_(self.out_tok[gslot], non_blocking=True)
            self.host_len[gslot].copy_(self.out_len[gslot], non_blocking=True)
            if self.keep_scores:                                    # (two ATen launches; verification runs only)
                best = self.order[lane][:, 0].long()
                self.host_lp[gslot].copy_(st.logprobs.view(self.NB, self.k, self.T)[self._rows, best], non_blocking=True)
            self.ev_done[gslot].record()
        self._gi += 1
        self._gfill = 0

    def outstanding(self) -> int:
        return self._submitted - self._collected

    def full(self) -> bool:
        """True when the next submit() would reuse the result slot of a group not yet collected."""
        return bool(self._where) and self._gi - self._where[0][0] >= self.RG

    def _oldest(self) -> Tuple[int, int]:
        if not self._where:
            raise RuntimeError("nothing outstanding")
        gi, gpos = self._where[0]
        if gi == self._gi:                                       # still waiting in a partially filled group
            self.flush()
        return gi, gpos

    def collect_device(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Device tensors (int32 [B,T] EOS-padded tokens, int32 [B] lengths) of the oldest outstanding
        batch, ordered after its decode on the CURRENT stream (for a following collective)."""
        gi, gpos = self._oldest()
        gslot = gi % self.RG
        cur = torch.cuda.current_stream()
        cur.wait_event(self.ev_done[gslot])
        self._where.pop(0)
        self._collected += 1
        rows = slice(gpos * self.B, (gpos + 1) * self.B)
        toks, lens = self.out_tok[gslot][rows].clone(), self.out_len[gslot][rows].clone()
        self.ev_res_free[gslot].record(cur)                      # the ring slot may be rewritten after these copies
        return toks, lens

    def collect(self) -> List[List[int]]:
        """Captions (token-id lists) of the oldest outstanding batch; blocks until it is decoded."""
        gi, gpos = self._oldest()
        gslot = gi % self.RG
        self.ev_done[gslot].synchronize()
        self._where.pop(0)
        self._collected += 1
        toks, lens = self.host_tok[gslot], self.host_len[gslot]
        return [toks[b, :int(lens[b])].tolist() for b in range(gpos * self.B, (gpos + 1) * self.B)]

    def collect_scored(self) -> Tuple[List[List[int]], List[torch.Tensor]]:
        """collect() plus the per-token log-probs (fp32, bit for bit what the search kept) of every caption; needs
        keep_scores=True."""
        if not self.keep_scores:
            raise RuntimeError("construct the pipeline with keep_scores=True")
        gi, gpos = self._oldest()
        gslot = gi % self.RG
        self.ev_done[gslot].synchronize()
        self._where.pop(0)
        self._collected += 1
        toks, lens, lps = self.host_tok[gslot], self.host_len[gslot], self.host_lp[gslot]
        rows = range(gpos * self.B, (gpos + 1) * self.B)
        return ([toks[b, :int(lens[b])].tolist() for b in rows], [lps[b, :int(lens[b])].clone() for b in rows])

    def __call__(self, images: torch.Tensor, enc_num_pads: Optional[Sequence[int]] = None) -> List[List[int]]:
        self.submit(images, enc_num_pads)
        while self.outstanding() > 1:
            self.collect()
        return self.collect()


# =================================================================================================
# image sharding across ranks
# =================================================================================================
def shard_indices(n_items: int, rank: int, world: int) -> Tuple[int, int, int]:
    """Contiguous shard [lo, hi) of rank `rank`; every rank is given `per` = ceil(n/world) slots
    (the tail shard is padded by the caller and trimmed after the gather).  → (lo, hi, per)."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    hi = min(lo + per, n_items)
    return lo, hi, per


def gather_captions(tokens: torch.Tensor, lengths: torch.Tensor, n_items: int, group=None
                    ) -> Optional[List[List[int]]]:
    """tokens int [per, T] / lengths int [per] of THIS rank's shard (rows beyond the shard padded) →
    on every rank the full list of `n_items` captions in global order.  ONE all_gather (RCCL over
    xGMI on GPUs, gloo on CPU); payload = world·per·(T+1)·4 bytes, latency-bound."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        toks, lens = tokens.cpu(), lengths.cpu()
        return [toks[i, :int(lens[i])].tolist() for i in range(min(n_items, toks.shape[0]))]
    world = dist.get_world_size(group)
    # one collective: lengths ride in an extra column of the token tensor
    packed = torch.cat([tokens.to(torch.int32), lengths.to(torch.int32)[:, None]], dim=1).contiguous()
    parts = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(parts, packed, group=group)
    allp = torch.cat(parts, 0).cpu()
    T = tokens.shape[1]
    return [allp[i, :int(allp[i, T])].tolist() for i in range(n_items)]


def run_shard(pipe: CaptionPipeline, n_local: int, fetch, pad_idx: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """This rank's shard through the software pipeline: items [0, n_local) in sub-batches of pipe.B (the
    ragged tail padded with repeats of its last item and trimmed), batches submitted as long as the result
    ring has room, results collected in order on the device.
      fetch(lo, hi) → inputs of local items [lo, hi) (device tensor)
    → (tokens int32 [n_local, T] padded with pad_idx, lengths int32 [n_local]) — the operands of gather_captions."""
    B, T = pipe.B, pipe.T
    toks = torch.full((n_local, T), pad_idx, dtype=torch.int32, device=pipe.device)
    lens = torch.zeros(n_local, dtype=torch.int32, device=pipe.device)
    pending: List[Tuple[int, int]] = []

    def drain_one():
        lo, n = pending.pop(0)
        t, l = pipe.collect_device()
        toks[lo:lo + n] = t[:n]
        lens[lo:lo + n] = l[:n]

    for lo in range(0, n_local, B):
        hi = min(lo + B, n_local)
        x, n = fetch(lo, hi), hi - lo
        if n < B:
            x = torch.cat([x, x[-1:].expand(B - n, *x.shape[1:])], 0)
        while pipe.full():
            drain_one()
        pipe.submit(x.contiguous())
        pending.append((lo, n))
    while pending:
        drain_one()
    return toks, lens


def run_steps(pipe: CaptionPipeline, ring: Sequence[torch.Tensor], n_steps: int, pad_idx: int
              ) -> Tuple[torch.Tensor, torch.Tensor]:
    """`n_steps` batches (ring[i % len(ring)]) through the software pipeline with the FINISHED token rows kept on the
    device: → (tokens int32 [n_steps·B, T] padded with pad_idx, lengths int32 [n_steps·B]) in submission order — the
    operands of ONE gather_captions at the end of a multi-GPU run (the north star's "gather of finished token-ID
    tensors"; a collective per step would synchronise the ranks every few milliseconds)."""
    B, T = pipe.B, pipe.T
    toks = torch.full((n_steps * B, T), pad_idx, dtype=torch.int32, device=pipe.device)
    lens = torch.zeros(n_steps * B, dtype=torch.int32, device=pipe.device)
    got = 0

    def take():
        nonlocal got
        t, l = pipe.collect_device()
        toks[got * B:(got + 1) * B] = t
        lens[got * B:(got + 1) * B] = l
        got += 1

    for i in range(n_steps):
        while pipe.full():
            take()
        pipe.submit(ring[i % len(ring)])
    while got < n_steps:
        take()
    return toks, lens


def caption_sharded(caption_batch, n_items: int, fetch, batch: int, T: int, eos_idx: int, device,
                    rank: int, world: int, group=None) -> List[List[int]]:
    """Evaluate `n_items` images split over `world` ranks.
      caption_batch(images[batch,...]) → (tokens [batch,T] int, lengths [batch] int) device tensors
      fetch(lo, hi) → images of global indices [lo, hi) (a tensor on `device`)
    Returns the captions of ALL items in global order on every rank."""
    lo, hi, per = shard_indices(n_items, rank, world)
    toks = torch.full((per, T), eos_idx, dtype=torch.int32, device=device)
    lens = torch.zeros(per, dtype=torch.int32, device=device)
    for s in range(lo, hi, batch):
        e = min(s + batch, hi)
        imgs = fetch(s, e)
        n = e - s
        if n < batch:                                   # pad the ragged tail with repeats of the last image
            imgs = torch.cat([imgs, imgs[-1:].expand(batch - n, *imgs.shape[1:])], 0)
        t, l = caption_batch(imgs)
        toks[s - lo:e - lo] = t[:n].to(torch.int32)
        lens[s - lo:e - lo] = l[:n].to(torch.int32)
    return gather_captions(toks, lens, n_items, group)
