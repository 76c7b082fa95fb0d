#!/usr/bin/env python3
"""Average per-launch value of every counter in a rocprofv3 counter_collection.csv, per kernel.  usage: pmc_generic.py CSV..."""
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        m = re.search(r"(conv3x3_bias_relu_kernel<[^>]*>|upsample2x_kernel|convert_input_kernel|convt2x2_kernel)", k)
        k = m.group(1) if m else k[:40]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, c in acc.items():
    print(k)
    for n, (v, cnt) in sorted(c.items()):
        print(f"   {n:32s} {v / cnt:16.1f}  (n={cnt})")
