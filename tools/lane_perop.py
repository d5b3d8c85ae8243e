"""Where does an edit lane's step go?  Per-op HIP-event profile of the batch-2 AudioLDM2 U-Net forward as the clip pipeline's
edit lanes build it (tile regime of the lane size, split-bf16 arithmetic) on a CU-masked stream, summed per op category and
U-Net level.

    PYTHONPATH=. python tools/lane_perop.py [cus=64] [arith] [share=2] > gpurun_out/lane_perop_cus64.json"""
import collections
import json
import re
import sys

import torch

from audioeditingcode_amd import configs, tape as tape_mod, weights
from audioeditingcode_amd.streams import PartitionStream
from audioeditingcode_amd.unet import PackedUNetWeights, UNetEngine

CUS = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ARITH = sys.argv[2] if len(sys.argv) > 2 else "bf16x6"
FUSE = next((int(a.split("=")[1]) for a in sys.argv[3:] if a.startswith("fuse=")), 1)       # 0: split-K with reduce launches (A/B)
tape_mod.FUSE_SPLITK = FUSE
SHARE = next((int(a.split("=")[1]) for a in sys.argv[3:] if a.startswith("share=")), 1)     # 2: the edit loop's CFG-shared head
fam = configs.FAMILIES["audioldm2"]
sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
pw = PackedUNetWeights(sd, "cuda:0")
regime = {64: "cus64", 128: "cus128"}.get(CUS)
ps = PartitionStream.acquire("cuda:0", cus=None if CUS >= 256 else range(CUS))
B = 2
with tape_mod.tile_regime(regime), tape_mod.arith_mode(ARITH):
    eng = UNetEngine(fam["unet"], pw, "cuda:0", B, 256, 16, ctx_len0=8, ctx_len1=16, share=SHARE)
g = torch.Generator().manual_seed(1)
eng.set_conditioning(ehs0=torch.randn(B, 8, 768, generator=g), ehs1=torch.randn(B, 16, 1024, generator=g), bias1=torch.zeros(B, 16))
eng.x_in.copy_(torch.randn(B, 256, 16, 8, generator=g))
eng.set_timestep(500)
with torch.cuda.stream(ps.stream):
    for _ in range(3):
        eng.forward()
    ps.stream.synchronize()
    runs = [eng.tape.profile() for _ in range(5)]
    ms = [min(r[i] for r in runs) for i in range(len(runs[0]))]
    eng.tape.capture()
    for _ in range(3):
        eng.tape.replay()
    ps.stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ps.stream)
    for _ in range(30):
        eng.tape.replay()
    e1.record(ps.stream)
    ps.stream.synchronize()
graph_ms = e0.elapsed_time(e1) / 30


def category(name, code):
    if code == 5:
        return "attention"
    if code in (2, 3, 22):
        return "groupnorm:" + ("transformer" if re.search(r"attentions\.\d+\.norm", name) else "resnet/out")
    for pat, cat in ((r"qkv", "qkv+ln"), (r"attn1\.to_out", "attn1.to_out"), (r"attn2\.to_out", "attn2.to_out"),
                     (r"scores\+softmax", "xattn scores+softmax"), (r"PV\+to_out", "xattn PV+to_out"), (r"ff1", "ff1+ln+geglu"),
                     (r"ff2", "ff2+proj_out"), (r"proj_in", "proj_in"), (r"conv_shortcut", "conv_shortcut"),
                     (r"conv1", "resnet conv1"), (r"conv2", "resnet conv2"), (r"samplers", "down/upsample conv"),
                     (r"time_emb|time_embedding|time_embed", "time embedding"), (r"conv_in|conv_out", "conv_in/out")):
        if re.search(pat, name):
            return cat
    return "other:" + name.split(".")[-1]


agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
rows = []
for op, mt, t in zip(eng.tape.ops, eng.tape.meta, ms):
    M = op.i[0] if op.code == 1 else (op.i[0] * op.i[2] if op.code == 5 else op.i[0] * op.i[1])
    key = (category(mt["name"], op.code), int(M))
    a = agg[key]
    a[0] += 1
    a[1] += t
    a[2] += mt["exec_flops"]
    rows.append(dict(name=mt["name"], code=op.code, M=int(M), N=int(op.i[1]) if op.code == 1 else None,
                     K=int(op.i[2]) if op.code == 1 else None, tile=int(op.i[29]) if op.code == 1 else None,
                     flags=int(op.flags), ms=round(t, 4)))
tot = sum(ms)
out = dict(cus=CUS, arith=ARITH, regime=regime, share=SHARE, fuse_splitk=FUSE, ops=len(ms), per_op_sum_ms=round(tot, 3), graph_replay_ms=round(graph_ms, 3),
           by_category=[dict(category=k[0], rows=k[1], launches=v[0], ms=round(v[1], 3), share=round(v[1] / tot, 4),
                             tflops=round(v[2] / (v[1] * 1e-3) / 1e12, 1) if v[1] else None)
                        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])], rows=rows)
print(json.dumps(out))
