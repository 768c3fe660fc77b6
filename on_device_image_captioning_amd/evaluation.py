"""Caption-level agreement metrics for the north star's "beam=3 captions within ±0.1 CIDEr-D" bar.

The reference scores predictions with `COCOEvalCap` (test.py:230-257), whose CIDEr-D is
eval/cider/cider_scorer.py (restated in cider.py).  README.md:98-106 quotes 140.4 for the single
model: that is 100 x the scorer's `compute_score` output (≈1.404 on Karpathy test), so

    README units = 100 x CiderD.compute_score(...)[0]        (a caption identical to its only
                                                               reference scores 10.0 → 1000 README units)

and "±0.1 CIDEr-D" means ±0.001 of `compute_score`.  Host-side, not on the GPU hot path.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

from .cider import CiderD

README_SCALE = 100.0


def prediction_words(tokens: Sequence[int], idx2word: Optional[Sequence] = None) -> str:
    """Prediction string as test.py:216-224 builds it: words[1:-1] (SOS and EOS dropped) joined by spaces.
    `idx2word=None` spells token id i as "w<i>" (synthetic vocabularies)."""
    body = list(tokens)[1:-1]
    if idx2word is None:
        return " ".join(f"w{int(t)}" for t in body)
    return " ".join(str(idx2word[int(t)]) for t in body)


def cider_d_readme(candidates: Sequence[Sequence[int]], references: Sequence[Sequence[Sequence[int]]],
                   idx2word: Optional[Sequence] = None) -> float:
    """CIDEr-D (README units) of one candidate caption per image against that image's reference captions
    (all given as token-id lists incl. SOS/EOS)."""
    if len(candidates) != len(references):
        raise ValueError("one candidate per image")
    gts: Dict[int, List[str]] = {i: [prediction_words(r, idx2word) for r in refs] for i, refs in enumerate(references)}
    res: Dict[int, List[str]] = {i: [prediction_words(c, idx2word)] for i, c in enumerate(candidates)}
    return README_SCALE * CiderD().compute_score(gts, res)[0]


def caption_agreement(candidates: Sequence[Sequence[int]], reference_captions: Sequence[Sequence[int]],
                      idx2word: Optional[Sequence] = None) -> dict:
    """How far `candidates` (e.g. bf16-mode captions) are from `reference_captions` (the fp32-mode captions,
    which equal the reference implementation's token for token) with the latter as the ONLY ground truth:
      cider_d          CIDEr-D of the candidates against them (README units)
      cider_d_self     the same score for the reference captions themselves (the attainable maximum)
      cider_d_delta    cider_d_self − cider_d: 0 iff every caption is identical; the north-star bar is 0.1
      identical        fraction of images whose caption is token-identical
      mean_prefix      mean length of the common token prefix / mean reference length
    """
    refs = [[list(r)] for r in reference_captions]
    score = cider_d_readme(candidates, refs, idx2word)
    self_score = cider_d_readme(reference_captions, refs, idx2word)
    same = sum(1 for c, r in zip(candidates, reference_captions) if list(c) == list(r))
    pref, tot = 0, 0
    for c, r in zip(candidates, reference_captions):
        n = 0
        for x, y in zip(c, r):
            if x != y:
                break
            n += 1
        pref += n
        tot += len(r)
    return {"cider_d": round(score, 3), "cider_d_self": round(self_score, 3),
            "cider_d_delta": round(self_score - score, 3), "unit": "README units = 100 x compute_score",
            "images": len(candidates), "identical": round(same / max(1, len(candidates)), 4),
            "mean_prefix": round(pref / max(1, tot), 4)}
