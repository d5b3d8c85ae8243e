"""How does one AudioLDM2 U-Net forward scale with the batch on a CU-masked stream?  (Round 5: should an edit lane step SEVERAL
clips in lockstep -- U-Net batch 2g for g clips -- instead of one?)

For every (CUs, batch): the engine is built as the clip pipeline would build it (tile regime of the lane size, split-bf16
arithmetic), captured in a hipGraph and replayed on the masked stream; reported: ms per forward, ms per clip-step (batch / 2
clips per forward) and the fp32-equivalent TF/s.

    PYTHONPATH=. python tools/batch_scaling.py [cus list, e.g. 64,128,256] [batches, e.g. 2,4,8,16] > gpurun_out/batch_scaling.jsonl"""
import json
import sys

import torch

from audioeditingcode_amd import configs, tape as tape_mod, weights
from audioeditingcode_amd.streams import PartitionStream
from audioeditingcode_amd.unet import PackedUNetWeights, UNetEngine

CUS = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "64,128,256").split(",")]
BATCHES = [int(b) for b in (sys.argv[2] if len(sys.argv) > 2 else "2,4,8,16").split(",")]
ARITH = sys.argv[3] if len(sys.argv) > 3 else "bf16x6"
fam = configs.FAMILIES["audioldm2"]
sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
pw = PackedUNetWeights(sd, "cuda:0")
g = torch.Generator().manual_seed(1)
for cus in CUS:
    regime = {64: "cus64", 128: "cus128"}.get(cus)
    ps = PartitionStream.acquire("cuda:0", cus=None if cus >= 256 else range(cus))
    for B in BATCHES:
        with tape_mod.tile_regime(regime), tape_mod.arith_mode(ARITH):
            eng = UNetEngine(fam["unet"], pw, "cuda:0", B, 256, 16, ctx_len0=8, ctx_len1=16)
        eng.set_conditioning(ehs0=torch.randn(B, 8, 768, generator=g), ehs1=torch.randn(B, 16, 1024, generator=g),
                             bias1=torch.zeros(B, 16))
        eng.x_in.copy_(torch.randn(B, 256, 16, 8, generator=g))
        eng.set_timestep(500)
        with torch.cuda.stream(ps.stream):
            for _ in range(2):
                eng.forward()
            ps.stream.synchronize()
            eng.tape.capture()
            for _ in range(3):
                eng.tape.replay()
            ps.stream.synchronize()
            n = max(4, 40 // B)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ps.stream)
            for _ in range(n):
                eng.tape.replay()
            e1.record(ps.stream)
            ps.stream.synchronize()
        ms = e0.elapsed_time(e1) / n
        lin = sum(1 for op in eng.tape.ops if op.code == 1 and op.i[29] >= 10)
        x6 = sum(1 for op in eng.tape.ops if op.code == 1 and op.flags & 4)
        print(json.dumps(dict(cus=cus, batch=B, clips=B // 2, ms_per_forward=round(ms, 3),
                              ms_per_clip_step=round(ms / (B // 2), 3), tflops=round(eng.tape.flops / ms / 1e9, 1),
                              ops=len(eng.tape.ops), lin_gemm_ops=lin, x6_ops=x6)), flush=True)
        del eng
        torch.cuda.empty_cache()
