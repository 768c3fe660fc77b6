#!/usr/bin/env python3
"""Average the counters of the LAST launches of each kernel in rocprofv3 counter_collection.csv files."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "gemm_bf16" not in name and "window_attention" not in name:
            continue
        acc[name[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        tail = v[-3:]
        print(f"    {c:32s} {sum(tail) / len(tail):16.1f}")
