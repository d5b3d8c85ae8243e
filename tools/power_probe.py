"""Is the chip at its power budget?  Samples `rocm-smi` (socket power, shader clock) every ~200 ms on a host thread while (a) the
batch-200 U-Net forward replays on the whole chip, (b) on a 128-CU stream, (c) the same beside two batch-2 edit-loop graphs on
64-CU lanes (the clip pipeline's steady state), and prints the averages.

    PYTHONPATH=. python tools/power_probe.py > gpurun_out/power_probe.jsonl"""
import json
import re
import subprocess
import threading
import time

import torch

from audioeditingcode_amd import configs, tape as tape_mod, weights
from audioeditingcode_amd.streams import PartitionStream, separate_queues
from audioeditingcode_amd.unet import PackedUNetWeights, UNetEngine

fam = configs.FAMILIES["audioldm2"]
sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
pw = PackedUNetWeights(sd, "cuda:0")
g = torch.Generator().manual_seed(1)


def mk(B, regime):
    with tape_mod.tile_regime(regime), tape_mod.arith_mode("bf16x6"):
        eng = UNetEngine(fam["unet"], pw, "cuda:0", B, 256, 16, ctx_len0=8, ctx_len1=16)
    eng.set_conditioning(ehs0=torch.randn(B, 8, 768, generator=g), ehs1=torch.randn(B, 16, 1024, generator=g),
                         bias1=torch.zeros(B, 16))
    eng.x_in.copy_(torch.randn(B, 256, 16, 8, generator=g))
    eng.set_timestep(500)
    return eng


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True,
                                     timeout=5).stdout
                d = json.loads(out)
                card = next(iter(d.values()))
                pw_ = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
                sclk = next((v for k, v in card.items() if "sclk" in k.lower()), None)
                m = re.search(r"(\d+)Mhz", str(sclk))
                self.rows.append((time.perf_counter(), pw_, int(m.group(1)) if m else None))
            except Exception:                       # noqa: BLE001
                pass
            time.sleep(0.15)


def measure(name, work, seconds=6.0):
    s = Sampler()
    torch.cuda.synchronize()
    s.start()
    t0 = time.perf_counter()
    n = work(seconds)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s.stop = True
    s.join()
    rows = [r for r in s.rows if r[0] - t0 > 1.0]            # skip the ramp
    p = [r[1] for r in rows if r[1] is not None]
    c = [r[2] for r in rows if r[2] is not None]
    print(json.dumps(dict(case=name, seconds=round(dt, 2), work=n, samples=len(rows),
                          power_w_avg=round(sum(p) / len(p), 1) if p else None, power_w_max=max(p) if p else None,
                          sclk_mhz_avg=round(sum(c) / len(c)) if c else None)), flush=True)


full = PartitionStream.acquire("cuda:0")
front = PartitionStream.acquire("cuda:0", cus=range(128, 256))
l0 = PartitionStream.acquire("cuda:0", cus=range(0, 64), index=0)
l1 = PartitionStream.acquire("cuda:0", cus=range(64, 128), index=1)
front, l0, l1 = separate_queues([front, l0, l1])
e200, e200p = mk(200, None), mk(200, "cus128")
e2a, e2b = mk(2, "cus64"), mk(2, "cus64")
for eng, ps in ((e200, full), (e200p, front), (e2a, l0), (e2b, l1)):
    with torch.cuda.stream(ps.stream):
        eng.forward()
        ps.stream.synchronize()
        eng.tape.capture()
        eng.tape.replay()
        ps.stream.synchronize()


def loop(pairs):
    def work(seconds):
        t0, n = time.perf_counter(), [0] * len(pairs)
        while time.perf_counter() - t0 < seconds:
            for k, (eng, ps, reps) in enumerate(pairs):
                with torch.cuda.stream(ps.stream):
                    for _ in range(reps):
                        eng.tape.replay()
                n[k] += reps
            for _, ps, _ in pairs:
                ps.stream.synchronize()
        return n
    return work


measure("idle", lambda s: time.sleep(3) or 0, 3.0)
measure("batch-200 forward, whole chip", loop([(e200, full, 1)]))
measure("batch-200 forward, 128-CU partition alone", loop([(e200p, front, 1)]))
measure("batch-2 forwards on two 64-CU lanes alone", loop([(e2a, l0, 25), (e2b, l1, 25)]))
measure("pipeline steady state: batch-200 on 128 CUs + two batch-2 lanes", loop([(e200p, front, 1), (e2a, l0, 27), (e2b, l1, 27)]))
