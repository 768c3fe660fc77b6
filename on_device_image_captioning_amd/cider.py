"""CIDEr-D scorer — the metric the north star's "within ±0.1 CIDEr-D" criterion is
stated in.  Clean-room restatement of the algorithm of reference eval/cider/cider_scorer.py:51-201 and
eval/cider/cider.py:13-52 (itself the standard coco-caption CIDEr-D): TF-IDF weighted n-gram (n = 1..4)
cosine similarity with count clipping, a Gaussian length penalty (sigma 6) and a x10 scale.  Host-side,
pure Python/numpy like the reference's; not part of the GPU hot path.

Reproduced quirks (pinned by tests/golden/cider.json, generated from the reference's scorer):
  * IDF uses log(number of images) - log(max(1, document frequency)), document frequency counted
    over the reference sets only;
  * the "length" fed to the Gaussian penalty is the number of BIGRAM tokens of a sentence
    (the reference accumulates term frequencies where len(ngram) - 1 == 1, cider_scorer.py:146-147);
  * similarity accumulates min(hyp, ref)·ref per n-gram of the hypothesis, then divides by the norms.
"""
from __future__ import annotations

import math
from collections import Counter, defaultdict
from typing import Dict, List, Sequence, Tuple

import numpy as np


def _ngrams(sentence: str, n: int = 4) -> Counter:
    words = sentence.split()
    c: Counter = Counter()
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            c[tuple(words[i:i + k])] += 1
    return c


class CiderD:
    def __init__(self, n: int = 4, sigma: float = 6.0):
        self.n, self.sigma = n, sigma

    def compute_score(self, gts: Dict[object, List[str]], res: Dict[object, List[str]]
                      ) -> Tuple[float, np.ndarray]:
        """gts: image id → list of (pre-tokenised, space-separated) reference captions;
        res: image id → [one hypothesis].  → (corpus score, per-image scores)."""
        assert gts.keys() == res.keys()
        ids = list(gts.keys())
        refs = [[_ngrams(r, self.n) for r in gts[i]] for i in ids]
        hyps = []
        for i in ids:
            assert isinstance(res[i], list) and len(res[i]) == 1 and len(gts[i]) > 0
            hyps.append(_ngrams(res[i][0], self.n))
        df: Dict[tuple, float] = defaultdict(float)
        for rs in refs:
            for g in set(g for r in rs for g in r):
                df[g] += 1.0
        log_n = math.log(float(len(ids)))

        def vectorise(cnt: Counter):
            vec = [dict() for _ in range(self.n)]
            norm = [0.0] * self.n
            length = 0
            for g, tf in cnt.items():
                k = len(g) - 1
                w = float(tf) * (log_n - math.log(max(1.0, df.get(g, 0.0))))
                vec[k][g] = w
                norm[k] += w * w
                if k == 1:
                    length += tf
            return vec, [math.sqrt(v) for v in norm], length

        scores = []
        for h, rs in zip(hyps, refs):
            hv, hn, hl = vectorise(h)
            total = np.zeros(self.n)
            for r in rs:
                rv, rn, rl = vectorise(r)
                val = np.zeros(self.n)
                for k in range(self.n):
                    acc = 0.0
                    for g, w in hv[k].items():
                        rw = rv[k].get(g, 0.0)
                        acc += min(w, rw) * rw
                    if hn[k] != 0 and rn[k] != 0:
                        acc /= hn[k] * rn[k]
                    val[k] = acc * math.exp(-((hl - rl) ** 2) / (2.0 * self.sigma ** 2))
                total += val
            scores.append(float(np.mean(total)) / len(rs) * 10.0)
        arr = np.array(scores)
        return float(arr.mean()), arr
