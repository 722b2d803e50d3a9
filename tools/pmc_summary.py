#!/usr/bin/env python3
"""Average per-dispatch counter values of the ccd GEMM kernel from rocprofv3 --pmc csv files."""
import collections
import csv
import sys

csv.field_size_limit(1 << 30)
acc = collections.OrderedDict()
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            if "gemm_bf16_kernel" not in r["Kernel_Name"]:
                continue
            d = acc.setdefault(r["Counter_Name"], [0.0, 0])
            d[0] += float(r["Counter_Value"])
            d[1] += 1
for k, (v, n) in acc.items():
    print(f"{k:34s} {v / n:16.0f}   (n={n})")
