"""Why does the back stage's codec (VAE decode + 2 vocoder passes: 45 ms alone on 256 CUs) cost ~200 ms inside the clip
pipeline?  Times the two engines on (a) an unmasked stream, idle chip; (b) a 128-CU stream, idle chip; (c) the 128-CU
stream while the batch-200 inversion forward replays on the other 128 CUs; (d) an unmasked stream beside that inversion.
Per-op profile of the slowest case.

    PYTHONPATH=. python tools/codec_partition.py -> gpurun_out/codec_partition.json"""
import collections
import json
import os
import time

import torch

from audioeditingcode_amd import models
from audioeditingcode_amd.streams import PartitionStream

dev = "cuda:0"
m = models.load_model("cvssp/audioldm2", dev, 200, allow_synthetic=True)
g = torch.Generator().manual_seed(0)
w = torch.randn(1, 8, 256, 16, generator=g).to(dev)
mel = torch.randn(1, 1, 1024, 64, generator=g).to(dev)
full = PartitionStream.acquire(dev)
lo = PartitionStream.acquire(dev, cus=range(128))
hi = PartitionStream.acquire(dev, cus=range(128, 256))
with torch.inference_mode():
    m.vae_decode(w)
    m.decode_to_mel(mel)
dec, voc = m._vae_dec(1, 256, 16), m._vocoder(1, 1024)
# a batch-200 inversion forward as background load
ed = m.editor(256, 16)
eng = ed.unet(200, 8, 16)
eng.set_conditioning(ehs0=torch.randn(200, 8, 768, generator=g), ehs1=torch.randn(200, 16, 1024, generator=g),
                     bias1=torch.zeros(200, 16))
eng.x_in.copy_(torch.randn(200, 256, 16, 8, generator=g))
eng.set_timestep(500)
with torch.cuda.stream(hi.stream):
    eng.forward()
    hi.stream.synchronize()
    eng.tape.capture()
out = {}


def codec(stream, n=5):
    with torch.cuda.stream(stream):
        for tp in (dec.tape, voc.tape, voc.tape):
            tp.run()
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            for tp in (dec.tape, voc.tape, voc.tape):
                tp.run()
        stream.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n


def with_load(stream):
    with torch.cuda.stream(hi.stream):
        for _ in range(4):
            eng.tape.replay()
    time.sleep(0.05)
    ms = codec(stream, 3)
    torch.cuda.synchronize()
    return ms


out["codec_ms_unmasked_idle_chip"] = codec(full.stream)
out["codec_ms_128cus_idle_chip"] = codec(lo.stream)
out["codec_ms_128cus_beside_inversion_on_other_128"] = with_load(lo.stream)
out["codec_ms_unmasked_beside_inversion_on_128"] = with_load(full.stream)
for k, v in out.items():
    print(f"{k}: {v:.1f} ms", flush=True)
for name, tp in (("vae_decode", dec.tape), ("vocoder", voc.tape)):
    for label, st in (("256", full.stream), ("128", lo.stream)):
        with torch.cuda.stream(st):
            tp.profile()
            ms = tp.profile()
        agg = collections.defaultdict(lambda: [0, 0.0])
        for mt, t in zip(tp.meta, ms):
            key = mt["name"].split(".")[0] + ":" + (mt["name"].split(".")[-1] if mt["code"] != 1 else "conv")
            agg[key][0] += 1
            agg[key][1] += t
        top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:6]
        out[f"{name}_perop_{label}"] = dict(total_ms=sum(ms), top=[(k, n, round(t, 3)) for k, (n, t) in top])
        print(f"{name} on {label} CUs: per-op sum {sum(ms):.2f} ms ({len(ms)} ops); top: "
              + ", ".join(f"{k} x{n} {t:.2f}" for k, (n, t) in top), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/codec_partition.json", "w"), indent=1)
