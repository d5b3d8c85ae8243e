"""How many independent batch-2 U-Net chains (edit / reference-order inversion steps of DIFFERENT clips) does it take to
fill the MI355X?  L engines, L HIP streams, n hipGraph replays of one forward each, all started together; reports the
per-lane forward time t(L) and the chip time per clip-step t(L)/L.  Run it under different GPU_MAX_HW_QUEUES settings
(the HIP runtime multiplexes streams onto that many hardware queues; default 4).

    [GPU_MAX_HW_QUEUES=8] PYTHONPATH=. python tools/lanes.py [Lmax] -> gpurun_out/lanes_q<queues>.json"""
import json
import os
import sys

import torch

from audioeditingcode_amd import configs, weights
from audioeditingcode_amd.unet import PackedUNetWeights, UNetEngine

LMAX = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
fam = configs.FAMILIES["audioldm2"]
sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
pw = PackedUNetWeights(sd, dev)
gen = torch.Generator().manual_seed(1)


def mk(B):
    eng = UNetEngine(fam["unet"], pw, dev, B, 256, 16, ctx_len0=8, ctx_len1=16)
    eng.set_conditioning(ehs0=torch.randn(B, 8, 768, generator=gen), ehs1=torch.randn(B, 16, 1024, generator=gen),
                         bias1=torch.zeros(B, 16))
    eng.x_in.copy_(torch.randn(B, 256, 16, 8, generator=gen))
    eng.set_timestep(500)
    return eng


engs = [mk(2) for _ in range(LMAX)]
streams = [torch.cuda.Stream() for _ in range(LMAX)]
for e, s in zip(engs, streams):
    with torch.cuda.stream(s):
        e.forward()
        s.synchronize()
        e.tape.capture()
        e.tape.replay()
        s.synchronize()
flops = engs[0].tape.flops
q = os.environ.get("GPU_MAX_HW_QUEUES", "default")
out = {"GPU_MAX_HW_QUEUES": q, "algorithmic_gflop_per_forward": flops / 1e9, "lanes": {}}
n = 60
for L in [l for l in (1, 2, 3, 4, 5, 6, 8, 10, 12) if l <= LMAX]:
    torch.cuda.synchronize()
    evs = []
    for e, s in zip(engs[:L], streams[:L]):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            a.record(s)
            for _ in range(n):
                e.tape.replay()
            b.record(s)
        evs.append((a, b))
    torch.cuda.synchronize()
    per_lane = [a.elapsed_time(b) / n for a, b in evs]
    t = max(per_lane)
    out["lanes"][L] = dict(forward_ms_per_lane=per_lane, chip_ms_per_clip_step=t / L, tflops=L * flops / t / 1e9)
    print(f"queues={q} L={L:2d}: {t:7.3f} ms per lane-step -> {t / L:6.3f} ms of chip time per clip-step "
          f"({L * flops / t / 1e9:6.1f} TF/s algorithmic)", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/lanes_q{q}.json", "w"), indent=1)
