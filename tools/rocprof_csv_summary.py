#!/usr/bin/env python
"""Summarise rocprofv3 CSV output (kernel trace + optional PMC passes) into a markdown table.
usage: rocprof_csv_summary.py <kernel_trace.csv> [<fetch_counter_collection.csv> <write_counter_collection.csv>]
FETCH_SIZE/WRITE_SIZE are in KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports 1/2 of the
bytes of wide coalesced reads, so the 'fetch x2' column doubles it."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:96]


kt = sys.argv[1]
agg = defaultdict(lambda: [0, 0])
with open(kt) as f:
    for r in csv.DictReader(f):
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
pmc = {}
for path in sys.argv[2:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            d = pmc.setdefault(short(r["Kernel_Name"]), defaultdict(float))
            d[r["Counter_Name"]] += float(r["Counter_Value"])
            d["n_" + r["Counter_Name"]] += 1
tot = sum(v[1] for v in agg.values())
print(f"# rocprofv3 summary: {kt}\n\ntotal kernel time {tot / 1e6:.3f} ms over {sum(v[0] for v in agg.values())} dispatches\n")
hdr = "| kernel | calls | total ms | avg us | % |"
if pmc:
    hdr += " fetch KiB/launch (x2 corrected) | write KiB/launch |"
print(hdr)
print("|---|---|---|---|---|" + ("---|---|" if pmc else ""))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    line = f"| `{k}` | {n} | {t / 1e6:.3f} | {t / n / 1e3:.2f} | {100 * t / tot:.1f} |"
    if pmc:
        d = pmc.get(k, {})
        fe = d.get("FETCH_SIZE", 0) / max(1, d.get("n_FETCH_SIZE", 1))
        wr = d.get("WRITE_SIZE", 0) / max(1, d.get("n_WRITE_SIZE", 1))
        line += f" {fe:.1f} ({2 * fe:.1f}) | {wr:.1f} |"
    print(line)
