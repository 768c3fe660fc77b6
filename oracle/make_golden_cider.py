"""tests/golden/cider.json: CIDEr-D scores of the REFERENCE's scorer (eval/cider/cider.py) on synthetic
pre-tokenised caption sets.  Build container only (imports /root/reference/eval)."""
import json
import os
import random
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
from eval.cider.cider import Cider          # noqa: E402  (pure Python + numpy)

rng = random.Random(1234)
vocab = [f"w{i}" for i in range(60)]


def sent(lo=4, hi=12):
    return " ".join(rng.choice(vocab[: rng.choice((10, 30, 60))]) for _ in range(rng.randint(lo, hi)))


cases = []
for n_img, n_ref in ((5, 3), (12, 5), (30, 2), (8, 1)):
    gts, res = {}, {}
    for i in range(n_img):
        refs = [sent() for _ in range(n_ref)]
        gts[str(i)] = refs
        words = refs[rng.randrange(n_ref)].split()
        if rng.random() < 0.5:                       # perturb a copy of a reference
            for _ in range(rng.randint(0, 3)):
                words[rng.randrange(len(words))] = rng.choice(vocab)
            if rng.random() < 0.3:
                words = words[: max(2, len(words) - 3)]
        else:
            words = sent().split()
        res[str(i)] = [" ".join(words)]
    score, scores = Cider().compute_score(gts, res)
    cases.append({"gts": gts, "res": res, "score": float(score), "scores": [float(s) for s in scores]})
json.dump(cases, open(os.path.join(ROOT, "tests", "golden", "cider.json"), "w"))
print([round(c["score"], 4) for c in cases])
