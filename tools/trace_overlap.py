"""Are the two kernel classes of the clip pipeline resident at once?  Reads a rocprofv3 --kernel-trace CSV of a pipelined
bench run and reports, per hardware queue: dispatches, busy time (union of kernel intervals), the dominant kernels; and
for the two busiest queues that carry the loops -- the one running `invert_step_kernel` (front stage: inversion at U-Net
batch 2G) and the one running `reverse_step_kernel` (back stage: edit loop at batch 2) -- the time during which BOTH have
a kernel in flight.

    rocprofv3 --kernel-trace --stats -d gpurun_out/kt_r03 -o kt --output-format csv -- \
        python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-batched
    python tools/trace_overlap.py gpurun_out/kt_r03/**/kt_kernel_trace.csv > profiles/r03_kernel_trace_pipeline.md"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)


def union(iv):
    iv = sorted(iv)
    out, cur_s, cur_e = [], None, None
    for s, e in iv:
        if cur_s is None:
            cur_s, cur_e = s, e
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            out.append((cur_s, cur_e))
            cur_s, cur_e = s, e
    if cur_s is not None:
        out.append((cur_s, cur_e))
    return out


def intersect(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


path = sys.argv[1]
rows = collections.defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        rows[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
t_min = min(s for v in rows.values() for s, _, _ in v)
t_max = max(e for v in rows.values() for _, e, _ in v)
print(f"# per-queue view of `{path}`\n")
print(f"{sum(len(v) for v in rows.values())} dispatches on {len(rows)} hardware queues over {(t_max - t_min) / 1e6:.1f} ms.\n")
print("| queue | dispatches | busy [ms] (union of kernel intervals) | first..last [ms] | top kernels by time |")
print("|---|---|---|---|---|")
info = {}
for q, v in sorted(rows.items(), key=lambda kv: -len(kv[1])):
    u = union([(s, e) for s, e, _ in v])
    busy = sum(e - s for s, e in u)
    by = collections.Counter()
    for s, e, n in v:
        by[n] += e - s
    top = ", ".join(f"`{n}` {t / 1e6:.1f}" for n, t in by.most_common(3))
    info[q] = dict(u=u, busy=busy, names=by, n=len(v))
    print(f"| {q} | {len(v)} | {busy / 1e6:.1f} | {(v[0][0] - t_min) / 1e6:.0f}..{(max(e for _, e, _ in v) - t_min) / 1e6:.0f} | {top} |")
# ---- the pipeline's queues by role (round 5: one inversion queue, SEVERAL edit lanes)
front = max((q for q in info if info[q]["names"].get("invert_step_kernel")), key=lambda q: info[q]["names"]["invert_step_kernel"], default=None)
lanes = sorted((q for q in info if info[q]["names"].get("reverse_step_kernel") and q != front),
               key=lambda q: -info[q]["names"]["reverse_step_kernel"])
lanes = [q for q in lanes if info[q]["names"]["reverse_step_kernel"] >= 0.05 * info[lanes[0]]["names"]["reverse_step_kernel"]] if lanes else []
if front is None or not lanes:
    print("\nno queue set with invert_step_kernel / reverse_step_kernel found (not a partition-pipeline trace?)")
    sys.exit(0)


def steps(q, name):
    return sum(1 for _, _, n in rows[q] if n.startswith(name))


print("\n## the pipeline's queues\n")
fb = info[front]["busy"]
print(f"* queue {front} = front stage (inversion at U-Net batch 2G: `conv_gemm_x6_kernel<256|128, ...>`, `attention_x6_kernel`, "
      f"`invert_step_kernel`): busy {fb / 1e6:.1f} ms, {steps(front, 'invert_step_kernel')} inversion step kernels;")
for q in lanes:
    bb = info[q]["busy"]
    both = intersect(info[front]["u"], info[q]["u"])
    print(f"* queue {q} = edit lane (batch-2 loop: `lin_gemm_kernel`, small `conv_gemm_x6_kernel` tiles, `reverse_step_kernel`): busy "
          f"{bb / 1e6:.1f} ms, {steps(q, 'reverse_step_kernel')} edit steps; **{both / 1e6:.1f} ms with a kernel of this lane AND the "
          f"front stage in flight** = {100 * both / bb:.1f} % of the lane's busy time;")
if len(lanes) >= 2:
    l01 = intersect(info[lanes[0]]["u"], info[lanes[1]]["u"])
    # all three: intersect the pairwise-intersection intervals with the front's
    ab, i, j = [], 0, 0
    a, b = info[lanes[0]]["u"], info[lanes[1]]["u"]
    while i < len(a) and j < len(b):
        s_, e_ = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e_ > s_:
            ab.append((s_, e_))
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    all3 = intersect(ab, info[front]["u"])
    print(f"* the two edit lanes together: {l01 / 1e6:.1f} ms with a kernel of BOTH in flight; front stage + both lanes: "
          f"**{all3 / 1e6:.1f} ms = {100 * all3 / (t_max - t_min):.1f} % of the traced wall time** with three kernels resident at once.")
wall = (t_max - t_min) / 1e6
print(f"* busy fractions of the traced wall time ({wall:.0f} ms, fill / warm-up / drain included): front {100 * fb / 1e6 / wall:.1f} %, "
      + ", ".join(f"lane {q} {100 * info[q]['busy'] / 1e6 / wall:.1f} %" for q in lanes) + ".")
# a concrete pair: the longest inversion kernel and the edit kernels that ran entirely inside it
fv = sorted(rows[front], key=lambda x: x[0] - x[1])[0]
inside = [(s_, e_, n) for s_, e_, n in rows[lanes[0]] if s_ >= fv[0] and e_ <= fv[1]]
print(f"* example: `{fv[2]}` on queue {front} ran for {(fv[1] - fv[0]) / 1e3:.0f} us; {len(inside)} kernels of queue {lanes[0]} started "
      f"and finished inside that interval" + (f" (e.g. `{inside[0][2]}`, {(inside[0][1] - inside[0][0]) / 1e3:.1f} us)." if inside else "."))
# per-clip device time of each stage from the step kernels: an inversion = 2 runs of invert_step kernels, an edit loop = 100 steps
