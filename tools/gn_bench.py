"""GroupNorm(+SiLU) at the inversion's batch: time and algorithmic GB/s of the stats + apply pair per shape, for several grid sizes of
gn_apply (tape.GN_APPLY_BLOCKS_PER_CU).

    PYTHONPATH=. python tools/gn_bench.py [B=200] > gpurun_out/gn_bench.jsonl"""
import json
import sys

import torch

from audioeditingcode_amd import _lib as L, tape as tape_mod
from audioeditingcode_amd.tape import Tape

B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
DEV = "cuda:0"
st = torch.cuda.Stream(DEV)
for HW, C, C1 in ((4096, 128, 0), (4096, 256, 128), (1024, 256, 0), (1024, 512, 256), (256, 384, 0), (256, 768, 384), (64, 640, 0),
                  (64, 1280, 640)):
    x = torch.randn(B, HW, C1 or C, device=DEV)
    x2 = torch.randn(B, HW, C - C1, device=DEV) if C1 else None
    ga, be = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
    ref = None
    for bpc in (2, 4, 8, 16, 32):
        tape_mod.GN_APPLY_BLOCKS_PER_CU = bpc
        tp = Tape(DEV)
        out = tp.alloc(B, HW, C)
        tp.groupnorm(x, ga, be, out, B=B, HW=HW, C=C, G=32, act=L.ACT_SILU, x2=x2, C1=C1)
        tp.finalize()
        with torch.cuda.stream(st):
            for _ in range(3):
                tp.run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(20):
                tp.run()
            e1.record(st)
            st.synchronize()
        ms = e0.elapsed_time(e1) / 20
        same = True if ref is None else bool(torch.equal(out, ref))
        ref = out.clone() if ref is None else ref
        print(json.dumps(dict(B=B, HW=HW, C=C, two_source=bool(C1), launches=len(tp.ops), apply_blocks_per_cu=bpc, us=round(ms * 1e3, 1),
                              algorithmic_gb_per_s=round(12 * B * HW * C / (ms * 1e-3) / 1e9, 1), bit_identical_to_first=same)), flush=True)
