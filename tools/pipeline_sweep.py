"""Sweep the partition plan of pipeline.ClipPipeline on the headline workload (AudioLDM2, T=200, tstart=100): seconds per
clip for a list of (edit_cus, edit_lanes), one model build.

    PYTHONPATH=. python tools/pipeline_sweep.py [K] [cfg ...]   cfg = edit_cus:edit_lanes   -> gpurun_out/pipeline_sweep.json"""
import gc
import json
import os
import sys
import time

import torch

from audioeditingcode_amd import models
from audioeditingcode_amd.pipeline import ClipPipeline
from audioeditingcode_amd.utils import prepare_waveform, synthetic_clip

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfgs = [tuple(int(x) for x in c.split(":")) for c in sys.argv[2:]] or [(128, 1), (112, 2), (120, 2), (104, 2)]
dev = torch.device("cuda:0")
T, tstart, G = 200, 100, 100
m = models.load_model("cvssp/audioldm2", dev, T, allow_synthetic=True)
args = (["a recording of a piano melody"], ["a recording of an electric guitar melody"], [""], [3.0], [12.0], T, tstart)


def clip_wave(i):
    w = prepare_waveform(synthetic_clip(10.0, seed=1234 + i), 1024 * 160)
    return torch.clip(torch.from_numpy(w)[None], -1, 1).to(dev)


def wave_to_mel(view, wave):
    mel, _, _ = view.get_fn_STFT().mel_spectrogram(wave)
    return mel[0].T[:1024][None, None].contiguous()


out = []
waves = [clip_wave(5000 + i) for i in range(K)]
for edit_cus, edit_lanes in cfgs:
    pipe = ClipPipeline(m, plan="partition", edit_cus=edit_cus, edit_lanes=edit_lanes, timestep_group=G)
    pipe.warm_up(clip_wave(99), *args, prepare=wave_to_mel, seeds=[999])
    pipe.edit_clips([clip_wave(i) for i in range(1 + edit_lanes)], *args, prepare=wave_to_mel, seeds=[1000 + i for i in range(1 + edit_lanes)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = pipe.edit_clips(waves, *args, prepare=wave_to_mel, seeds=[2000 + i for i in range(K)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(torch.isfinite(r[2]).all() for r in res)
    rep = pipe.report()
    row = dict(edit_cus=edit_cus, edit_lanes=edit_lanes, clips=K, s_per_clip=dt / K, clips_per_s=K / dt,
               steady_state_estimate_s=(dt - 1.6) / max(1, K - 1), device_ms=rep["device_ms"],
               clip_latency_ms_avg=rep["clip_latency_ms_avg"])
    out.append(row)
    print(json.dumps(row), flush=True)
    pipe.close()
    del pipe, res
    gc.collect()
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/pipeline_sweep.json", "w"), indent=1)
