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


# ------------------------------------------------------------------------------------------------
# Caption normalisation used in front of the scorers (reference
# utils/language_utils.py:4-72 — the regex cleaner that stands in for the PTB tokenizer, whose Java
# jar is not available).  Outputs are pinned by tests/golden/text_cleaner.json, recorded from the
# reference's functions (oracle/make_golden_text.py).
# ------------------------------------------------------------------------------------------------
import re as _re

#: stand-alone tokens dropped by remove_punctuations (reference :17-31)
PUNCTUATIONS = frozenset(["''", "'", "``", "`", ".", "?", "!", ",", ":", "-", "--", "...", ";"])
_NON_WORD = _re.compile(r"([^\w0-9])")


def compute_num_pads(list_bboxes) -> List[int]:
    """Padding needed to bring every item to the longest one's length (reference :4-13)."""
    longest = max((len(b) for b in list_bboxes), default=-1)
    return [longest - len(b) for b in list_bboxes]


def remove_punctuations(sentences: Sequence[str]) -> List[str]:
    """Drops the single-space-separated tokens that are pure punctuation (reference :16-39; empty
    tokens from doubled spaces survive, exactly as there)."""
    return [" ".join(w for w in s.split(" ") if w not in PUNCTUATIONS) for s in sentences]


def lowercase_and_clean_trailing_spaces(sentences: Sequence[str]) -> List[str]:
    return [s.lower().rstrip() for s in sentences]


def add_space_between_non_alphanumeric_symbols(sentences: Sequence[str]) -> List[str]:
    """Every character that is not a word character gets a space on both sides (reference :46-47)."""
    return [_NON_WORD.sub(r" \1 ", s) for s in sentences]


def tokenize(list_sentences: Sequence[str]) -> List[List[str]]:
    """Split on single spaces and drop the empty strings (reference :50-57)."""
    return [[w for w in s.split(" ") if w != ""] for s in list_sentences]


def convert_vector_word2idx(sentence: Sequence[str], word2idx_dict: Dict[str, int]) -> List[int]:
    return [word2idx_dict[w] for w in sentence]


def convert_allsentences_word2idx(sentences, word2idx_dict):
    return [convert_vector_word2idx(s, word2idx_dict) for s in sentences]


def clean_captions(sentences: Sequence[str]) -> List[str]:
    """The cleaning chain the reference's data loaders apply to raw captions before scoring / vocabulary
    lookup (data/vizwiz_dataset.py:273-276, losses/reward.py:19-23: lowercase → separate symbols → drop punctuation → tokenise), returned
    re-joined with single spaces — the pre-tokenised form cider.CiderD takes."""
    s = lowercase_and_clean_trailing_spaces(sentences)
    s = add_space_between_non_alphanumeric_symbols(s)
    s = remove_punctuations(s)
    return [" ".join(t) for t in tokenize(s)]
