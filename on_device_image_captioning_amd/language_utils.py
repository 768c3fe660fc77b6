"""Caller-side text helpers (SURVEY §8 rows A20, F1 prerequisites).

Behaviour follows reference utils/language_utils.py:75-93 and test.py:216-224; the 10000-word COCO
vocabulary (data, identical to reference vocab/coco_vocab_idx_dict.json; SOS=79, EOS=77) ships as
data/coco_vocab_words.json.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Sequence, Tuple

_VOCAB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "coco_vocab_words.json")


def load_vocab(path: str = _VOCAB) -> Tuple[Dict[str, int], List[str]]:
    """→ (word2idx, idx2word) exactly as demo.py:57-63 builds them from its pickle."""
    with open(path) as f:
        idx2word = json.load(f)
    return {w: i for i, w in enumerate(idx2word)}, idx2word


def convert_vector_idx2word(sentence: Sequence[int], idx2word_list: Sequence[str]) -> List[str]:
    return [idx2word_list[i] for i in sentence]


def convert_allsentences_idx2word(sentences, idx2word_list):
    return [convert_vector_idx2word(s, idx2word_list) for s in sentences]


def tokens2description(tokens: Sequence[int], idx2word_list: Sequence[str], sos_idx: int, eos_idx: int) -> str:
    """ids → "A … ." : drop SOS, cut at the first EOS, full stop glued to the last word,
    capitalise (reference utils/language_utils.py:82-93)."""
    kept = []
    for tok in tokens:
        if tok == sos_idx:
            continue
        if tok == eos_idx:
            break
        kept.append(idx2word_list[tok])
    kept[-1] += "."
    return " ".join(kept).capitalize()


def test_style_sentence(tokens: Sequence[int], idx2word_list: Sequence[str]) -> str:
    """The evaluation-time form of test.py:216-224: words[1:-1] joined by spaces."""
    return " ".join(convert_vector_idx2word(tokens, idx2word_list)[1:-1])
