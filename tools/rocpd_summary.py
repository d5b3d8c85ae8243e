#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg duration.
usage: python tools/rocpd_summary.py gpurun_out/prof/xxx_results.db > profiles/xxx_kernel_stats.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    short = re.sub(r"\(.*", "", name)
    short = re.sub(r"^void ", "", short)
    a = agg.setdefault(short, [0, 0])
    a[0] += 1
    a[1] += (e - s)
tot = sum(v[1] for v in agg.values())
print(f"# rocprofv3 --kernel-trace summary ({sys.argv[1]})\n")
print(f"total kernel time {tot / 1e6:.3f} ms over {len(rows)} dispatches\n")
print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"| `{k[:110]}` | {n} | {t / 1e6:.3f} | {t / n / 1e3:.2f} | {100 * t / tot:.1f} |")
