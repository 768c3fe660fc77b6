#!/usr/bin/env python3
"""profiles/r03_pmc_traffic_<workload>_<mode>.json from two rocprofv3 counter_collection.csv files (FETCH_SIZE pass, WRITE_SIZE
pass): HBM bytes per launch of each kernel family bench.py reports = (2·FETCH_SIZE + WRITE_SIZE)·1024 (gfx950
correction of MI355X_MICROARCH.md §HBM).    python tools/pmc_traffic_json.py fetch.csv write.csv out.json "<cmd>" """
import csv
import json
import re
import sys
from collections import defaultdict

FAMILIES = {"gemm_bf16": r"gemm_bf16_", "gemm_x3": r"gemm_x3_nt_kernel", "window_attention_x3": r"window_attention_h2_kernel",
            "gemm_f32": r"gemm_f32_", "gemm_fp8": r"gemm_lowp_nt_kernel<(\d+, ){6}1,",
            "gemm_f16": r"gemm_lowp_nt_kernel<(\d+, ){6}2,", "layernorm": r"\blayernorm_kernel",
            "window_attention_bf16": r"window_attention_bf16", "patch_embed": r"patch_embed_kernel",
            "patch_merge_layernorm": r"layernorm_kernel<\d+, true", "cross_attn_step": r"cross_attn_step_kernel",
            "dynexp_step": r"dynexp_", "logsoftmax_topk": r"logsoftmax_topk_kernel",
            "beam_search_step": r"beam_search_step_kernel"}


def load(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for fam, pat in FAMILIES.items():
            if re.search(pat, r["Kernel_Name"]):
                if fam == "layernorm" and re.search(FAMILIES["patch_merge_layernorm"], r["Kernel_Name"]):
                    continue
                tot[fam] += float(r["Counter_Value"])
                cnt[fam] += 1
                break
    return tot, cnt


fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, nw = load(sys.argv[2], "WRITE_SIZE")
out = {}
for fam in fetch:
    rd = 2.0 * fetch[fam] * 1024 / nf[fam]
    wr = write.get(fam, 0.0) * 1024 / max(1, nw.get(fam, 1))
    out[fam] = {"hbm_bytes_per_launch": round(rd + wr), "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                "launches_sampled": nf[fam],
                "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per MI355X_MICROARCH.md "
                          "§HBM; " + (sys.argv[4] if len(sys.argv) > 4 else "")}
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_hash: which HIP sources these counters describe)
out["_meta"] = {"kernel_source_hash": bench.kernel_source_hash(), "command": sys.argv[4] if len(sys.argv) > 4 else ""}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    if k != "_meta":
        print(k, v["hbm_bytes_per_launch"], v["launches_sampled"])
