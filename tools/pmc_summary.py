"""Summarise rocprofv3 --pmc passes over tools/pmc_forward.py (one U-Net forward per batch size) per kernel family.

    python tools/pmc_summary.py <dir with the *_counter_collection.csv of every pass> [--json profiles/r02_pmc_forward.json]

Counters (separate passes, as MI355X_MICROARCH.md prescribes): FETCH_SIZE (KiB; DOUBLED on gfx950), WRITE_SIZE (KiB),
and the SQ set SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * #CU-ish
normalisation): reported both raw and as a ratio to SQ_BUSY_CYCLES (the per-SE busy window)."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)


def fam(n):
    m = re.match(r"(conv_gemm_x6_kernel<\d+, \d+|conv_gemm_kernel<\d+, \d+|lin_gemm_kernel<\d+, \d+, \d+|conv_gemm_wsk_kernel|attention_split_kernel<\d+|"
                 r"attention_kernel<\d+|gn_small_kernel|gn_stats_kernel|gn_apply_kernel|splitk_reduce_kernel)", n)
    return m.group(1) + (">" if "<" in m.group(1) else "") if m else n[:40]


d = sys.argv[1]
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = fam(short(r["Kernel_Name"]))
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
names = sorted({c for k in agg for c in agg[k]})
print(f"# rocprofv3 --pmc summary of `{d}` (counters: {', '.join(names)})\n")
print("| kernel family | launches | fetch MB (x2) | write MB | MFMA busy / SQ busy | MFMA busy / GRBM active | wave cycles waiting (WAIT_ANY) | issue stalls (WAIT_INST_ANY) |")
print("|---|---|---|---|---|---|---|---|")


def g(k, c):
    return agg[k].get(c, 0.0)


order = sorted(agg, key=lambda k: -(g(k, "FETCH_SIZE") + g(k, "SQ_BUSY_CYCLES")))
tot_f = tot_w = 0.0
n_gemm = 0
for k in order:
    n = cnt[k].get("FETCH_SIZE") or min(cnt[k].values())      # a counter present in two passes must not double the count
    f = g(k, "FETCH_SIZE") * 2 * 1024 / 1e6
    w = g(k, "WRITE_SIZE") * 1024 / 1e6
    mb = g(k, "SQ_VALU_MFMA_BUSY_CYCLES")
    sb = g(k, "SQ_BUSY_CYCLES")
    ga = g(k, "GRBM_GUI_ACTIVE")
    wc = g(k, "SQ_WAVE_CYCLES")
    r1 = f"{mb / sb:.3f}" if sb else "-"
    r2 = f"{mb / ga:.1f}" if ga else "-"
    r3 = f"{g(k, 'SQ_WAIT_ANY') / wc:.2f}" if wc else "-"
    r4 = f"{g(k, 'SQ_WAIT_INST_ANY') / wc:.2f}" if wc else "-"
    print(f"| `{k}` | {n} | {f:.1f} | {w:.1f} | {r1} | {r2} | {r3} | {r4} |")
    if k.startswith("conv_gemm") or k.startswith("lin_gemm") or k.startswith("splitk"):
        tot_f += f
        tot_w += w
        n_gemm += n
if n_gemm:
    print(f"\nconv_gemm + lin_gemm family: {n_gemm} launches, fetch {tot_f:.1f} MB (x2-corrected) + write {tot_w:.1f} MB "
          f"= {(tot_f + tot_w) / n_gemm:.2f} MB per launch")
if out_json:
    h = hashlib.sha1()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f_ in ("conv_gemm.hip", "conv_gemm_x6.hip", "lin_gemm.hip", "cg_params.h", "aed_common.h"):     # = bench.csrc_hash()
        h.update(open(os.path.join(root, "audioeditingcode_amd", "csrc", f_), "rb").read())
    # --alg-total-bytes: the sum tools/pmc_forward.py prints ("algorithmic bytes", all forwards of the profiled command);
    # measured and algorithmic are divided by the SAME launch count so their ratio is the ratio of the totals
    alg_tot = float(sys.argv[sys.argv.index("--alg-total-bytes") + 1]) if "--alg-total-bytes" in sys.argv else None
    json.dump(dict(csrc_hash=h.hexdigest()[:12], counters="rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, "
                   "separate passes", measured_bytes_per_launch=1e6 * (tot_f + tot_w) / max(1, n_gemm),
                   algorithmic_bytes_per_launch=(alg_tot / max(1, n_gemm) if alg_tot else None),
                   measured_over_algorithmic=(1e6 * (tot_f + tot_w) / alg_tot if alg_tot else None), launches=n_gemm,
                   source=f"tools/pmc_forward.py via tools/pmc_summary.py ({os.path.basename(d.rstrip('/'))})"),
              open(out_json, "w"), indent=1)
