"""tests/golden/text_cleaner.json: outputs of the REFERENCE's caption-cleaning helpers
(utils/language_utils.py:4-72) on a fixed set of raw captions.  Build container only (imports
/root/reference/utils/language_utils.py, pure Python).  Test infrastructure, never shipped."""
import json
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
from utils import language_utils as L      # noqa: E402

RAW = [
    "A man riding a wave on top of a surfboard.",
    "Two dogs,  one brown -- one white; both running!   ",
    "The sign says \"STOP\" (it's red) ...",
    "a person's hand holding an i-phone 6s: what's on it?",
    "  leading spaces & symbols #1 @home ",
    "Plain lowercase sentence without punctuation",
    "`` quoted '' text ` with ' odd tokens - and - dashes",
    "Numbers 12,5 and 3.14 are kept apart; 100% sure.",
    "",
    "ÀÉ unicode café naïve — em-dash",
]

out = {"raw": RAW}
lo = L.lowercase_and_clean_trailing_spaces(RAW)
sp = L.add_space_between_non_alphanumeric_symbols(lo)
rp = L.remove_punctuations(sp)
tk = L.tokenize(rp)
out["lowercase_and_clean_trailing_spaces"] = lo
out["add_space_between_non_alphanumeric_symbols"] = sp
out["remove_punctuations"] = rp
out["tokenize"] = tk
out["remove_punctuations_direct"] = L.remove_punctuations(RAW)
out["tokenize_direct"] = L.tokenize(RAW)
lists = [[1, 2, 3], [], [4], [5, 6, 7, 8, 9]]
out["compute_num_pads_in"] = lists
out["compute_num_pads"] = L.compute_num_pads(lists)
w2i = {w: i for i, w in enumerate(sorted({w for s in tk for w in s}))}
out["word2idx"] = w2i
out["convert_allsentences_word2idx"] = L.convert_allsentences_word2idx(tk, w2i)
i2w = sorted(w2i, key=w2i.get)
out["convert_allsentences_idx2word"] = L.convert_allsentences_idx2word(out["convert_allsentences_word2idx"], i2w)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "text_cleaner.json"), "w"), ensure_ascii=False, indent=1)
print("wrote", len(RAW), "cases")
