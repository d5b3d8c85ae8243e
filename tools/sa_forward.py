"""One Stable Audio Open DiT forward (full width, 24 layers, 1025 tokens, 130-token context) per batch size, for rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -o kt --output-format csv -- python tools/sa_forward.py 2 40
    rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ... -- python tools/sa_forward.py 2 40
Three forwards per batch size under the kernel trace (the first builds / warms), exactly one under --pmc (AED_ONE=1)."""
import os
import sys

import torch

from audioeditingcode_amd import configs, weights
from audioeditingcode_amd.stable_audio import DiTEngine, PackedDiTWeights

cfg = dict(configs.FAMILIES["stable_audio"]["dit"])
sd = weights.random_state_dict(weights.dit_param_shapes(cfg), seed=0)
packed = PackedDiTWeights(sd, cfg, "cuda:0")
del sd
g = torch.Generator().manual_seed(1)
S = 130
for B in [int(a) for a in sys.argv[1:]] or [2, 40]:
    eng = DiTEngine(cfg, packed, "cuda:0", B, S)
    ctx = torch.randn(B, S, cfg["cross_attention_input_dim"], generator=g)
    ctx[::2] = 0
    eng.set_conditioning(ctx, torch.randn(B, cfg["global_states_input_dim"], generator=g))
    eng.set_timestep(0.37)
    eng.x_in.copy_(torch.randn(B, cfg["sample_size"], cfg["in_channels"], generator=g))
    for _ in range(1 if os.environ.get("AED_ONE") == "1" else 3):
        eng.forward()
    torch.cuda.synchronize()
    assert torch.isfinite(eng.v).all()
    print("forward done", B, "launches", len(eng.tape.ops), "algorithmic GFLOP", eng.tape.flops / 1e9, flush=True)
    del eng
    torch.cuda.empty_cache()
