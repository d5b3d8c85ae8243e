"""A/B of the self-attention kernels at the inversion's shapes: fp32 transposed-score kernel (attention.hip) against the
split-bf16 kernel (attention_x6.hip), whole chip and on a 128-CU stream (the pipeline's inversion partition).

    PYTHONPATH=. python tools/attn_x6_ab.py > gpurun_out/attn_x6_ab.json"""
import json
import sys
import time

import torch

from audioeditingcode_amd.streams import PartitionStream
from audioeditingcode_amd.tape import Tape

DEV = "cuda:0"
# AudioLDM2 level 1 / level 2 at batch 200; DiT batch 16; the edit loop's batch 2 (fp32 side = the key-split kernel there)
SHAPES = [(200, 8, 1024, 32), (200, 8, 256, 48), (16, 24, 1024, 64), (2, 8, 1024, 32), (2, 8, 256, 48)]
full = PartitionStream.acquire(DEV)
half = PartitionStream.acquire(DEV, cus=range(128, 256))
lane = PartitionStream.acquire(DEV, cus=range(0, 64))
out = []
for B, H, N, D in SHAPES:
    C = H * D
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B, N, 3 * C, generator=g).to(DEV)
    rec = dict(B=B, H=H, N=N, D=D, gflop=4e-9 * B * H * N * N * D)
    res = {}
    for name, variant in (("f32", 0), ("x6", 3)):
        tp = Tape(DEV)
        o = tp.alloc(B, N, C)
        tp.attention(qkv, qkv[..., C:], qkv[..., 2 * C:], o, B=B, H=H, Nq=N, Nk=N, D=D, ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C,
                     bsq=N * 3 * C, bsk=N * 3 * C, bsv=N * 3 * C, bso=N * C, scale=D ** -0.5, variant=variant)
        for label, ps in (("chip", full), ("cus128", half), ("cus64", lane)):
            with torch.cuda.stream(ps.stream):
                for _ in range(3):
                    tp.run()
                ps.stream.synchronize()
                t0 = time.perf_counter()
                R = 20 if B > 4 else 200
                for _ in range(R):
                    tp.run()
                ps.stream.synchronize()
                ms = 1e3 * (time.perf_counter() - t0) / R
            rec[f"{name}_{label}_ms"] = round(ms, 4)
            rec[f"{name}_{label}_tflops"] = round(rec["gflop"] / ms, 1)
        res[name] = o.cpu()
    rec["rel_l2_x6_vs_f32"] = float((res["x6"].double() - res["f32"].double()).norm() / res["f32"].double().norm())
    out.append(rec)
    print(json.dumps(rec), flush=True)
