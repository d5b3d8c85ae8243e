"""Do the pipeline's CU partitions disturb each other?  (Round 4: two 64-CU edit lanes beside the split-bf16 inversion ran their
clips 1.8x slower than the same lanes alone.)  Builds the partition pipeline's engines, then replays the captured U-Net
forward graphs of the stages in isolation and side by side, timing each stream with the WALL CLOCK from a common start
(NOTES.md: per-stream event pairs under-report on oversubscribed queues).

    GPU_MAX_HW_QUEUES=8 PYTHONPATH=. python tools/lane_interference.py [--edit-lanes 2] [--arith bf16x6]   -> one JSON line"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--edit-lanes", type=int, default=2)
ap.add_argument("--edit-cus", type=int, default=128)
ap.add_argument("--arith", default="bf16x6")
ap.add_argument("--front-arith", default=None, help="arithmetic of the inversion stage's engines only (A/B: is it the bf16 stream?)")
ap.add_argument("--T", type=int, default=200)
a = ap.parse_args()

from audioeditingcode_amd import models                                   # noqa: E402
from audioeditingcode_amd.pipeline import ClipPipeline                    # noqa: E402
from audioeditingcode_amd.utils import prepare_waveform, synthetic_clip   # noqa: E402

dev = torch.device("cuda:0")
m = models.load_model("cvssp/audioldm2", dev, a.T, allow_synthetic=True)
m.arith = a.arith
pipe = ClipPipeline(m, plan="partition", edit_cus=a.edit_cus, edit_lanes=a.edit_lanes, timestep_group=100)
if a.front_arith:
    for w in pipe.workers:
        if w.stage == "front":
            w.view.arith = a.front_arith
wave = torch.clip(torch.from_numpy(prepare_waveform(synthetic_clip(10.0, seed=1), 1024 * 160))[None], -1, 1).to(dev)


def to_mel(view, w):
    mel, _, _ = view.get_fn_STFT().mel_spectrogram(w)
    return mel[0].T[:1024][None, None].contiguous()


pipe.warm_up(wave, ["a piano"], ["a guitar"], [""], [3.0], [12.0], a.T, a.T // 2, prepare=to_mel, seeds=[1])
torch.cuda.synchronize()


def engine(w, B):
    ed = w.view.editor(256, 16)
    cand = sorted(((len(k), e) for k, e in ed._unets.items() if k[0] == B), key=lambda t: -t[0])
    with torch.inference_mode():
        for pl in ed._plans.values():
            pl["state"].zero_()
    return cand[0][1]


class Clocks:
    """sclk samples (rocm-smi) while a scenario runs: is the slowdown a clock drop under the split-bf16 stream's power?"""

    def __init__(self):
        import subprocess
        import threading
        self.samples, self.stop = [], threading.Event()

        def loop():
            while not self.stop.is_set():
                try:
                    r = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
                    d = json.loads(r.stdout)
                    card = next(iter(d.values()))
                    v = [val for key, val in card.items() if "sclk" in key.lower()]
                    if v:
                        self.samples.append(str(v[0]))
                except Exception as e:                  # noqa: BLE001
                    self.samples.append(f"err:{e!r}"[:40])
                    return
        self.th = threading.Thread(target=loop, daemon=True)

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *a):
        self.stop.set()
        self.th.join(timeout=6)


front = [(engine(w, 200), w.lane.stream) for w in pipe.workers if w.stage == "front"]
back = [(engine(w, 2), w.lane.stream) for w in pipe.workers if w.stage == "back"]
with torch.inference_mode():
    for eng, st in front + back:
        with torch.cuda.stream(st):
            eng.tape.capture()
            eng.tape.replay()
torch.cuda.synchronize()


def run(members):
    """members: [(eng, stream, n)] -- enqueue everything round-robin, return per-member ms per replay (wall clock to the
    member's own completion) from a common start."""
    evs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    left = [n for _, _, n in members]
    while any(left):
        for k, (eng, st, _) in enumerate(members):
            if left[k]:
                with torch.cuda.stream(st):
                    eng.tape.replay()
                left[k] -= 1
    for _, st, _ in members:
        e = torch.cuda.Event()
        e.record(st)
        evs.append(e)
    out = [None] * len(members)
    pending = set(range(len(members)))
    while pending:
        for k in list(pending):
            if evs[k].query():
                out[k] = 1e3 * (time.perf_counter() - t0) / members[k][2]
                pending.discard(k)
        time.sleep(0.0005)
    return [round(x, 3) for x in out]


res = dict(env_GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES"), arith=a.arith, front_arith=a.front_arith or a.arith,
           edit_lanes=a.edit_lanes, edit_lane_cus=pipe.edit_lane_cus)


def scenario(name, members, labels):
    with Clocks() as ck:
        out = run(members)
    res[name] = dict(zip(labels, out), sclk=ck.samples[:6])
    print(name, res[name], file=sys.stderr, flush=True)


with torch.inference_mode():
    f, b = front[0], back
    nb, nf = 60, 5
    bl = [f"back{k}" for k in range(len(b))]
    scenario("back_lanes_alone", [(e, s, nb) for e, s in b], bl)
    scenario("one_back_lane_alone", [(b[0][0], b[0][1], nb)], bl[:1])
    scenario("front_alone", [(f[0], f[1], nf)], ["front"])
    scenario("front_and_back_lanes", [(f[0], f[1], nf)] + [(e, s, nb) for e, s in b], ["front"] + bl)
    scenario("front_and_one_back", [(f[0], f[1], nf), (b[0][0], b[0][1], nb)], ["front", "back0"])
    if len(b) > 1:
        scenario("front_and_other_back", [(f[0], f[1], nf), (b[1][0], b[1][1], nb)], ["front", "back1"])
print(json.dumps(res))
