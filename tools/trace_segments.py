"""Segment a rocprofv3 --kernel-trace CSV of bench.py into the diffusion steps of the clip and summarise each class.

    rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o kt --output-format csv -- \
        python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batched
    python tools/trace_segments.py gpurun_out/kt/**/kt_kernel_trace.csv > profiles/r02_kernel_trace_headline.md

A segment ends with the fused step kernel(s) of the loop (reverse_step_kernel = one edit step at U-Net batch 2;
a run of invert_step_kernel = one inversion call at U-Net batch 2G) and starts after the previous one; only segments
with a full U-Net forward (>= 300 launches) are kept.  Per class: launches, sum of kernel durations, wall span
(first start -> last end), per-family breakdown, and the conv/lin GEMM family's algorithmic TFLOP/s (2*M*N*K summed over
the tape of that batch shape, built on the CPU -- no GPU needed to run this tool)."""
import collections
import os
import csv
import re
import sys

PEAK = 157.3
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n


def family(n):
    if n.startswith("conv_gemm") or n.startswith("lin_gemm") or n.startswith("splitk_reduce"):
        return "gemm (conv_gemm + lin_gemm)"
    if n.startswith("attention"):
        return "attention"
    if n.startswith("gn_"):
        return "groupnorm"
    return "other"


PEAK_BF16 = 2500.0


def tape_flops(B, share=None):
    """(algorithmic GEMM flops, algorithmic forward flops, #ops, executed fp32-equivalent flops of the records the split-bf16
    kernel takes, ... of the records on fp32-input MFMAs) of the U-Net tape at batch B, laid out on the CPU in the product's
    arithmetic (tape.arith_mode("bf16x6"), whole-chip tile tables) AND with the product's CFG row sharing (share = 2: the traced
    engines compute the context-free head once per [uncond | prompt] pair, so their EXECUTED flops are 3.6 % below the unshared
    tape's -- round 5's trace summary counted the unshared tape and overstated the executed rate by that much; `AED_TRACE_SHARE=1`
    counts the unshared tape for traces taken with --no-share-cfg-rows)."""
    import os
    if share is None:
        share = int(os.environ.get("AED_TRACE_SHARE", "2"))
    import torch
    from audioeditingcode_amd import configs, tape as tape_mod, weights
    from audioeditingcode_amd.unet import UNetEngine
    fam = configs.FAMILIES["audioldm2"]
    sd = {k: torch.zeros(s) for k, s in weights.unet_param_shapes(fam["unet"]).items()}
    with tape_mod.arith_mode("bf16x6"):
        eng = UNetEngine(fam["unet"], sd, "cpu", B, 256, 16, ctx_len0=8, ctx_len1=16, share=share if B % share == 0 else 1)
    conv = sum(m["flops"] for m in eng.tape.meta if m["code"] == 1)
    x6 = sum(m["exec_flops"] for op, m in zip(eng.tape.ops, eng.tape.meta) if m["code"] == 1 and (op.flags & 4) and op.i[29] < 10)
    f32 = sum(m["exec_flops"] for op, m in zip(eng.tape.ops, eng.tape.meta) if m["code"] == 1 and not ((op.flags & 4) and op.i[29] < 10))
    return conv, eng.tape.flops, len(eng.tape.ops), x6, f32


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
segs, cur = [], []
i = 0
while i < len(rows):
    s, e, n = rows[i]
    cur.append(rows[i])
    if n.startswith("reverse_step_kernel") or n.startswith("invert_step_kernel") or n.startswith("ddim_step"):
        kind = n.split("_kernel")[0]
        j = i + 1
        nstep = 1
        while j < len(rows) and (rows[j][2].startswith(kind) or rows[j][2].startswith("advance")):
            cur.append(rows[j])
            nstep += rows[j][2].startswith(kind)
            j += 1
        segs.append((kind, nstep, cur))
        cur = []
        i = j
        continue
    i += 1

classes = collections.OrderedDict()
for kind, nstep, ks in segs:
    if len(ks) < 300:
        continue
    classes.setdefault((kind, nstep), []).append(ks)

print(f"# per-step segmentation of `{sys.argv[1]}`\n")
print(f"{len(rows)} dispatches, {sum(e - s for s, e, _ in rows) / 1e6:.1f} ms of kernel time; "
      f"{len(segs)} step segments, classes with a full U-Net forward:\n")
for (kind, nstep), lst in classes.items():
    B = 2 * nstep
    conv_fl, all_fl, n_ops, x6_fl, f32_fl = tape_flops(B)
    n = len(lst)
    launches = sum(len(k) for k in lst) / n
    busy = sum(sum(e - s for s, e, _ in k) for k in lst) / n / 1e6
    span = sum(k[-1][1] - k[0][0] for k in lst) / n / 1e6
    print(f"## {kind} x{nstep} per segment -> U-Net batch {B}: {n} segments\n")
    print(f"* launches per segment {launches:.0f} (U-Net tape: {n_ops} ops); sum of kernel durations **{busy:.3f} ms**; "
          f"wall span first-start -> last-end **{span:.3f} ms**")
    print(f"* whole segment: {all_fl / 1e9:.1f} GF algorithmic -> {all_fl / span / 1e9:.1f} TFLOP/s = "
          f"{all_fl / span / 1e9 / PEAK:.3f} of the {PEAK} TF fp32-MFMA peak\n")
    fam_t = collections.defaultdict(lambda: [0, 0.0])
    ker_t = collections.defaultdict(lambda: [0, 0.0])
    for k in lst:
        for s, e, nm in k:
            fam_t[family(nm)][0] += 1
            fam_t[family(nm)][1] += (e - s) / 1e6
            ker_t[nm][0] += 1
            ker_t[nm][1] += (e - s) / 1e6
    print("| family | launches / segment | ms / segment | share | avg us |")
    print("|---|---|---|---|---|")
    for fm, (c, t) in sorted(fam_t.items(), key=lambda kv: -kv[1][1]):
        extra = ""
        if fm.startswith("gemm"):
            extra = f" ({conv_fl / (t / n) / 1e9:.1f} TFLOP/s algorithmic = {conv_fl / (t / n) / 1e9 / PEAK:.3f} of peak)"
        print(f"| {fm}{extra} | {c / n:.0f} | {t / n:.3f} | {100 * t / n / busy:.1f} % | {1e3 * t / c:.2f} |")
    # PHYSICAL fractions (round 5): executed MFMA flops of a kernel class over the peak of the instruction it issues
    t_x6 = sum(t for nm, (c, t) in ker_t.items() if nm.startswith("conv_gemm_x6")) / n
    t_f32 = sum(t for nm, (c, t) in ker_t.items() if nm.startswith("conv_gemm_kernel") or nm.startswith("lin_gemm") or
                nm.startswith("splitk_reduce")) / n
    if t_x6 > 0:
        print(f"\n* `conv_gemm_x6_kernel<*>` (v_mfma_f32_32x32x16_bf16, six piece products per fp32 product): {x6_fl / 1e9:.1f} GF "
              f"fp32-equivalent per segment = {6 * x6_fl / 1e12:.2f} TF of executed bf16 MFMA flops in {t_x6:.3f} ms -> "
              f"**{6 * x6_fl / t_x6 / 1e9:.1f} TFLOP/s = {6 * x6_fl / t_x6 / 1e9 / PEAK_BF16:.3f} of the {PEAK_BF16:.0f} TF bf16 MFMA peak** "
              f"({x6_fl / t_x6 / 1e9:.1f} TFLOP/s fp32-equivalent)")
    if t_f32 > 0:
        print(f"* fp32-input MFMA GEMM kernels (`conv_gemm_kernel`, `lin_gemm_kernel`, + `splitk_reduce`): {f32_fl / 1e9:.1f} GF in "
              f"{t_f32:.3f} ms -> {f32_fl / t_f32 / 1e9:.1f} TFLOP/s = {f32_fl / t_f32 / 1e9 / PEAK:.3f} of the {PEAK} TF fp32 MFMA peak")
    print("\n| kernel | launches / segment | ms / segment | avg us |")
    print("|---|---|---|---|")
    for nm, (c, t) in sorted(ker_t.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"| `{nm[:90]}` | {c / n:.1f} | {t / n:.3f} | {1e3 * t / c:.2f} |")
    print()
