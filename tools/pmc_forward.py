#!/usr/bin/env python
"""One AudioLDM2 U-Net forward at the two batch shapes of the bench (2*G and 2), for rocprofv3 --pmc passes:
    rocprofv3 --pmc FETCH_SIZE -d out -o f --output-format csv -- python tools/pmc_forward.py 40 2
Counter passes serialise every dispatch (~10 ms each), so this runs exactly one forward per batch size."""
import os
import sys

import torch

from audioeditingcode_amd import configs, tape as tape_mod, weights
from audioeditingcode_amd.unet import PackedUNetWeights, UNetEngine

# AED_PMC_ARITH: arithmetic the engines are built under (default = the product's, bf16x6); AED_PMC_TAPMAJOR=1: the round-1..3
# tap-major K traversal (op flag 32) instead of the grouped one, for the traffic A/B (cg_params.h kgroup)
ARITH = os.environ.get("AED_PMC_ARITH", "bf16x6")
TAPMAJOR = os.environ.get("AED_PMC_TAPMAJOR", "0") == "1"
NFASTEST = os.environ.get("AED_PMC_NFASTEST", "0") == "1"      # round 6: the n-fastest tile order of rounds 1-5 (op flag 1024), traffic A/B

fam = configs.FAMILIES["audioldm2"]
sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
pw = PackedUNetWeights(sd, "cuda:0")
g = torch.Generator().manual_seed(1)
for B in [int(a) for a in sys.argv[1:]] or [40, 2]:
    with tape_mod.arith_mode(ARITH):
        # the inversion's engine computes the context-free head once per [uncond | prompt] row pair (round 5); the edit loop's does not
        eng = UNetEngine(fam["unet"], pw, "cuda:0", B, 256, 16, ctx_len0=8, ctx_len1=16, share=2 if B > 2 else 1)
    if TAPMAJOR or NFASTEST:
        for op in eng.tape.ops:
            if op.code == 1:
                op.flags |= (32 if TAPMAJOR else 0) | (1024 if NFASTEST else 0)
        eng.tape._arr = None
    eng.set_conditioning(ehs0=torch.randn(B, 8, 768, generator=g), ehs1=torch.randn(B, 16, 1024, generator=g),
                         bias1=torch.zeros(B, 16))
    eng.x_in.copy_(torch.randn(B, 256, 16, 8, generator=g))
    eng.set_timestep(500)
    eng.forward()
    torch.cuda.synchronize()
    conv = [m for m in eng.tape.meta if m["code"] == 1]
    print("arith", ARITH, "tap-major" if TAPMAJOR else "grouped K order", flush=True)
    print("forward done", B, "conv_gemm launches", len(conv), "algorithmic bytes", sum(m["bytes"] for m in conv), flush=True)
