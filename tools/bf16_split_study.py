"""CPU study for the split-bf16 GEMM arithmetic (csrc/conv_gemm_x6.hip): how far is an fp32 product computed as a sum of
bf16 x bf16 partial products (exact in fp32, accumulated in fp32) from the fp64 result, next to a plain fp32 GEMM?

An fp32 value is cut into bf16 pieces a = a1 + a2 + a3 (24 significand bits = 3 x 8, the split is EXACT); the product a*b is
the sum of 9 piece products a_i*b_j of relative size 2^-8(i+j-2).  Variants: x3 = {11, 12, 21}; x6 = x3 + {22, 13, 31};
x9 = all.  Each piece product is exact in fp32 (8 x 8 significand bits); what differs from an fp32 FMA chain is which terms
are dropped and the order of the fp32 additions.  Emulation here: every term is an fp32 numpy matmul of bf16-valued fp32
arrays (products exact, fp32 accumulation in BLAS order) -- the MFMA's internal adder tree is not modelled.

    python tools/bf16_split_study.py            # prints a markdown table
"""
import numpy as np


def bf16_round(x, mode):
    b = x.view(np.uint32)
    if mode == "trunc":
        return (b & np.uint32(0xFFFF0000)).view(np.float32)
    r = b + np.uint32(0x7FFF) + ((b >> np.uint32(16)) & np.uint32(1))          # round to nearest even on bit 16
    return (r & np.uint32(0xFFFF0000)).view(np.float32)


def split3(x, mode):
    x = np.ascontiguousarray(x, dtype=np.float32)
    h = bf16_round(x, mode)
    r1 = x - h
    m = bf16_round(r1, mode)
    r2 = r1 - m
    lo = bf16_round(r2, mode)
    return h, m, lo, float(np.abs(r2 - lo).max())


TERMS = {"x3": [(0, 0), (0, 1), (1, 0)],
         "x6": [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)],
         "x9": [(i, j) for i in range(3) for j in range(3)]}


def emulate(A, W, kind, mode):
    a = split3(A, mode)
    w = split3(W, mode)
    acc = np.zeros((A.shape[0], W.shape[0]), np.float32)
    for i, j in sorted(TERMS[kind], key=lambda t: -(t[0] + t[1])):      # small terms first
        acc += a[i] @ w[j].T
    return acc, max(a[3], w[3])


def main():
    rng = np.random.default_rng(0)
    rows = []
    for K in (320, 2560, 5760):
        for dist in ("normal", "mixed-scale"):
            A = rng.standard_normal((512, K)).astype(np.float32)
            W = (rng.standard_normal((256, K)) / np.sqrt(K)).astype(np.float32)
            if dist == "mixed-scale":                                   # activations after SiLU / with outlier channels
                A *= np.exp(rng.standard_normal((1, K)) * 1.5).astype(np.float32)
            ref = A.astype(np.float64) @ W.astype(np.float64).T
            nrm = np.linalg.norm(ref)
            err = lambda c: float(np.linalg.norm(c.astype(np.float64) - ref) / nrm)      # noqa: E731
            row = {"K": K, "dist": dist, "fp32": err(A @ W.T)}
            for mode in ("rne", "trunc"):
                for kind in ("x3", "x6", "x9"):
                    c, resid = emulate(A, W, kind, mode)
                    assert resid == 0.0, "the 3-way split must be exact"
                    row[f"{kind}/{mode}"] = err(c)
            rows.append(row)
    cols = ["fp32", "x3/rne", "x6/rne", "x9/rne", "x3/trunc", "x6/trunc", "x9/trunc"]
    print("| K | data | " + " | ".join(cols) + " |")
    print("|---|---|" + "---|" * len(cols))
    for r in rows:
        print(f"| {r['K']} | {r['dist']} | " + " | ".join(f"{r[c]:.2e}" for c in cols) + " |")


if __name__ == "__main__":
    main()
