#!/bin/bash
# round 6, lease 17: what would a loader WITHOUT the bf16 split buy on the batch-200 shapes?  x6_bench `quick` (tiles 8 / 9, 16- and
# 32-wide chunks) with the product library and with two DIAGNOSTIC builds of conv_gemm_x6.hip (-DX6_DIAG_NOSPLIT=1: the W rows are not
# split, =2: neither operand is; wrong values, representative time): the upper bound of pre-split operands in HBM
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06q; mkdir -p $O
X=audioeditingcode_amd/x6_bench
$X 20 quick > $O/product.jsonl 2> $O/product.err; echo "product rc=$?"
# (LD_PRELOAD of the diagnostic library had no effect on the first try -- the feature matrix still passed at 1e-6: x6_bench resolves
# libaed.so through its $ORIGIN runpath, so each diagnostic library gets a private copy of the bench binary beside it)
for v in 1 2; do D=/tmp/x6_nosplit$v; mkdir -p $D; cp $X $D/x6_bench; cp scratch/libaed_nosplit$v.so $D/libaed.so; done
/tmp/x6_nosplit1/x6_bench 20 quick > $O/nosplit_w.jsonl 2> $O/nosplit_w.err; echo "nosplit W rc=$? (feature matrix: $(grep 'feature cases' $O/nosplit_w.err))"
/tmp/x6_nosplit2/x6_bench 20 quick > $O/nosplit_aw.jsonl 2> $O/nosplit_aw.err; echo "nosplit A+W rc=$? (feature matrix: $(grep 'feature cases' $O/nosplit_aw.err))"
$X 20 quick > $O/product2.jsonl 2> /dev/null
python - <<PY
import json
def load(f):
    d={}
    for ln in open('$O/'+f):
        if ln.startswith('{') and '"shape"' in ln:
            r=json.loads(ln); d[(r['shape'], r['variant'])]=r
    return d
p, p2, w, aw = load('product.jsonl'), load('product2.jsonl'), load('nosplit_w.jsonl'), load('nosplit_aw.jsonl')
print('rel L2 vs the fp32 kernel (must be ~1e-6 for the product, ~4e-3 for the unsplit diagnostics):', [round(next(iter(d.values()))['rel_l2_vs_fp32_kernel'] if d else -1, 7) for d in (p, w, aw)])
print('| shape (M x N x K) | variant | product us (TF/s fp32-eq) | repeat us | W unsplit us (x) | A+W unsplit us (x) |')
print('|---|---|---|---|---|---|')
for k, r in p.items():
    if not r['variant'].startswith('x6'): continue
    a, b, c = p2.get(k), w.get(k), aw.get(k)
    f=lambda q: '-' if q is None else f"{q['us']:.1f} ({r['us']/q['us']:.3f}x)"
    print(f"| {k[0]} ({r['M']} x {r['N']} x {r['K']}) | {k[1]} | {r['us']:.1f} ({r['tflops']:.1f}) | {'-' if a is None else a['us']} | {f(b)} | {f(c)} |")
PY
