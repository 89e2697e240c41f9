#!/usr/bin/env python3
"""Condenses the three rocprofv3 runs of tools/profile_round.sh:

    tools/traffic_summary.py <dir> <passes>

<dir>/stats  kernel trace + stats (average duration per kernel)
<dir>/fetch  --pmc FETCH_SIZE      <dir>/write  --pmc WRITE_SIZE
Counters are summed over all launches of a kernel and divided by the number of
passes over the batch.  Both counters are in KiB.  gfx950 correction: FETCH_SIZE
tallies the 128-byte requests of a wide coalesced streaming read at 64 bytes
each, so read bytes = 2 x FETCH_SIZE x 1024 for the kernels that read that way
(validate_kernel's bodies, the table pre-pass, the builders); the fill kernel reads
nothing but 64-byte records with scalar loads, which the counter tallies at full
size (tools/scalar_fetch_probe.hip), so its read bytes = FETCH_SIZE x 1024.
WRITE_SIZE is exact for 16-byte-per-lane streaming stores."""
import collections
import csv
import glob
import json
import sys


def short(name):
    return name.split("oval::")[1].split("(")[0] if "oval::" in name else None


def counter_sum(d, counter):
    agg, launches = collections.defaultdict(float), collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == counter:
                agg[k] += float(r["Counter_Value"])
                launches[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    return agg, {k: len(v) for k, v in launches.items()}


def main():
    d, passes = sys.argv[1], int(sys.argv[2])
    print("# kernel durations (rocprofv3 --kernel-trace --stats)")
    for f in glob.glob(d + "/stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if k:
                print("%-34s calls %6s  avg %12.1f ns  total %14s ns  %5.1f %%" % (
                    k[:34], r["Calls"], float(r["AverageNs"]), r["TotalDurationNs"], float(r["Percentage"])))
    fetch, nf = counter_sum(d + "/fetch", "FETCH_SIZE")
    write, nw = counter_sum(d + "/write", "WRITE_SIZE")
    print("# HBM traffic per pass over the batch (%d passes), PMC counters, gfx950 corrections applied" % passes)
    out = {}
    for k in sorted(set(fetch) | set(write)):
        rd = (1.0 if k.startswith("fill_kernel") else 2.0) * fetch.get(k, 0.0) * 1024 / passes
        wr = write.get(k, 0.0) * 1024 / passes
        print("%-34s launches/pass %5.1f  read %15.0f B  written %15.0f B" % (
            k[:34], max(nf.get(k, 0), nw.get(k, 0)) / passes, rd, wr))
        out[k] = {"read_bytes": int(rd), "write_bytes": int(wr)}
    print("# json " + json.dumps(out))


if __name__ == "__main__":
    main()
