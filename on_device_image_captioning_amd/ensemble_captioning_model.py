"""Ensemble beam search (reference models/ensemble_captioning_model.py:5-291, same class
name and call shape: `EsembleCaptioningModel(models_list, rank)`, `forward(..., mode='beam_search')`).

Every member encodes and decodes on its own step engine; the per-step distribution is
log(mean_m softmax(logits_m)) (`odic_ensemble_logprobs`), its row-wise top-k (`odic_topk_rows`) feeds the
same on-device beam bookkeeping (`odic_beam_step`) as the single-model search.  The members share ONE beam
state (tokens, ancestor table, positions, finished flags) — only their caches and logits are private, so
the incremental decoding stays exact for each of them.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn as nn

from . import ops
from .captioning_model import CaptioningModel, _DONE_POLL


class EsembleCaptioningModel(CaptioningModel):
    def __init__(self, models_list, rank):
        super().__init__()
        self.num_models = len(models_list)
        self.models_list = models_list
        self.rank = rank
        self.dummy_linear = nn.Linear(1, 1)                       # reference :13 (keeps .parameters() non-empty)
        for model in self.models_list:
            model.eval()

    def forward(self, enc_x, dec_x=None, enc_x_num_pads=[0], dec_x_num_pads=[0], apply_log_softmax=False,
                mode="beam_search", **kwargs):
        assert mode == "beam_search", "this class supports only beam search."
        sos_idx = kwargs.get("sos_idx", -999)
        eos_idx = kwargs.get("eos_idx", -999)
        return self.ensemble_beam_search(enc_x, enc_x_num_pads, sos_idx=sos_idx, eos_idx=eos_idx,
                                         beam_size=kwargs.get("beam_size", 5),
                                         how_many_outputs=kwargs.get("how_many_outputs", 1),
                                         max_seq_len=kwargs.get("beam_max_seq_len", 20),
                                         sample_or_max=kwargs.get("sample_or_max", "max"))

    def forward_enc(self, enc_input, enc_input_num_pads):
        return [m.forward_enc(enc_input, enc_input_num_pads) for m in self.models_list]

    def ensemble_beam_search(self, enc_input, enc_input_num_pads, sos_idx, eos_idx, beam_size=3, how_many_outputs=1,
                             max_seq_len=20, sample_or_max="max") -> Tuple[List[List[List[int]]], torch.Tensor]:
        assert (how_many_outputs <= beam_size), "requested output per sequence must be lower than beam width"
        assert (sample_or_max == "max" or sample_or_max == "sample"), \
            "argument must be chosen between 'max' and 'sample'"
        if sample_or_max != "max":
            raise NotImplementedError("the ensemble search is built for sample_or_max='max'")
        mems = self.forward_enc(enc_input, enc_input_num_pads)
        engs = [m._captioner_engine() for m in self.models_list]
        dv = engs[0].device
        B, S, _ = mems[0].shape
        k = beam_size
        steps = max(1, max_seq_len - 1)
        T = steps + 1
        states = []
        for m, eng, mem in zip(self.models_list, engs, mems):
            st = eng.new_state(B, k, T, eng.project_kv(mem), m._enc_lens(B, S, enc_input_num_pads))
            if states:                                           # one beam state for all members
                lead = states[0]
                st.anc, st.row_valid, st.next_tok, st.pos = lead.anc, lead.row_valid, lead.next_tok, lead.pos
            states.append(st)
        lead = states[0]
        lead.tokens[:, :, 0] = sos_idx
        lead.next_tok.fill_(sos_idx)
        V = engs[0].g.vocab_size
        avg = torch.empty(lead.N, V, dtype=torch.float32, device=dv)
        for t in range(steps):
            for eng, st in zip(engs, states):
                eng.step_logits(st)                              # reads the shared next_tok / pos / ancestor table
            ops.ensemble_logprobs([st.logits for st in states], avg)
            ops.topk_rows(avg, lead.cand_val, lead.cand_idx, k)
            ops.beam_step(lead.cand_val, lead.cand_idx, lead.beam_state, lead.n_img, lead.beams, lead.T, eos_idx)
            if t >= 1 and (t + 1) % _DONE_POLL == 0 and t + 1 < steps and int(lead.done.item()):
                break
        order = torch.empty(B, k, dtype=torch.int32, device=dv)
        score = torch.empty(B, k, dtype=torch.float32, device=dv)
        ops.beam_finalize(lead.beam_state, order, score, B, k)
        order_h, n_elem_h, tokens_h = order.cpu(), lead.n_elem.view(B, k).cpu(), lead.tokens.cpu()
        res_tok: List[List[List[int]]] = []
        lp_rows = []
        for b in range(B):
            per = []
            for j in range(how_many_outputs):
                i = int(order_h[b, j])
                n = int(n_elem_h[b, i])
                per.append(tokens_h[b, i, :n].tolist())
                lp_rows.append(lead.logprobs[b, i, :n])
            res_tok.append(per)
        lp = torch.nn.utils.rnn.pad_sequence(lp_rows, batch_first=True).view(B, how_many_outputs, -1)
        return res_tok, lp
