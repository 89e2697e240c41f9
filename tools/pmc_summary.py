#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter CSVs per kernel: tools/pmc_summary.py <dir>"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "oval::" not in k or "init_results" in k or "chunk_table" in k:
            continue
        short = k.split("oval::")[1].split("(")[0]
        agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
kernels = sorted({k for k, _ in agg})
counters = sorted({c for _, c in agg})
print("%-34s" % "counter" + "".join("%26s" % k[:25] for k in kernels))
for c in counters:
    print("%-34s" % c + "".join("%26.4g" % (sum(agg[(k, c)]) / max(1, len(agg[(k, c)]))) for k in kernels))
