"""EXPERIMENTAL leg: the benchmark clip (AudioLDM2, T=200, tstart=100, timestep-batched inversion: U-Net batch 200) with the
LDS-staged GEMMs of the BATCHED engines on split-bf16 MFMAs (`model.arith = "bf16x6"`, csrc/conv_gemm_x6.hip) next to the
product's fp32-MFMA arithmetic, same weights, same clip, same seed, one clip at a time on the whole chip.

Prints ONE JSON line: the batch-200 U-Net forward in both arithmetics (ms, rel L2 of eps between them), seconds per clip in
both, and the rel L2 between the two edited latents and the two waveforms.  Not the headline: bench.py runs this as a bounded
sub-process and records it under `extras.x6_inversion`; nothing here feeds `value`.

    PYTHONPATH=. python tools/bench_x6_inversion.py [--T 200] [--tstart 100] [--group 100] [--clips 2]"""
import argparse
import json
import sys
import time

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=200)
ap.add_argument("--tstart", type=int, default=100)
ap.add_argument("--group", type=int, default=100)
ap.add_argument("--clips", type=int, default=2)
ap.add_argument("--model_id", default="cvssp/audioldm2")
ap.add_argument("--seconds", type=float, default=10.0)       # 1024 mel frames -> 256x16 latent: the benched shapes (tile tables)
a = ap.parse_args()

from audioeditingcode_amd import _lib as L, models                                        # noqa: E402
from audioeditingcode_amd.main_run import edit_clip                                      # noqa: E402
from audioeditingcode_amd.utils import load_audio, synthetic_clip                        # noqa: E402

dev = "cuda:0"
t0 = time.time()
m = models.load_model(a.model_id, dev, a.T, allow_synthetic=True)
print(f"weights ({m.weights_source}) ready in {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
x0, _, _ = load_audio((synthetic_clip(a.seconds, seed=1234), m.get_sr()), m.get_fn_STFT(), device=dev, stft=True,
                      model_sr=m.get_sr())
src, tgt, neg = ["a recording of a piano melody"], ["a recording of a violin melody"], [""]
SEED, oracle_w = 5, None
# With the committed CPU-oracle run of the benched schedule at hand (tests/golden/bench_parity_T200.npz: bench.py's clip, prompts,
# seed; oracle/make_bench_parity_golden.py) edit THAT clip: both arithmetics are then also compared with the oracle.
import os                                                                                # noqa: E402
import numpy as np                                                                       # noqa: E402
_fx = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bench_parity_T200.npz")
if os.path.exists(_fx) and a.T == 200 and a.tstart == 100:
    fx = np.load(_fx)
    x0 = torch.from_numpy(fx["x0"]).to(dev)
    src, tgt, neg = ([str(p)] for p in fx["prompts"])
    SEED, oracle_w = int(fx["seed"]), torch.from_numpy(fx["w_edit"])


def rel(x, y):
    return float((x.double() - y.double()).norm() / y.double().norm())


def run_clips(arith):
    m.arith = arith
    outs, times = [], []
    for k in range(a.clips + 1):                 # first one builds engines / graphs
        torch.manual_seed(SEED)
        torch.cuda.synchronize()
        t = time.perf_counter()
        audio, _, w_edit = edit_clip(m, x0, src, tgt, neg, [3.0], [12.0], a.T, a.tstart, schedule="batched",
                                     timestep_group=a.group)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t)
        outs = (audio.float().cpu(), w_edit.float().cpu())
    return outs, min(times[1:])


res = dict(workload=f"AudioLDM2 T={a.T} tstart={a.tstart}, inversion batched {a.group} timesteps per U-Net call, one clip at "
                    f"a time; arith of the batched engines' LDS-staged GEMMs: f32 MFMA vs split-bf16 (bf16x6)")
def partial(tag):
    """Every finished leg is printed at once (stderr): a later failure must not lose it (round 3 lost both clip legs that way)."""
    print(f"[x6_inversion partial:{tag}] " + json.dumps(res), file=sys.stderr, flush=True)


(audio_f, w_f), s_f = run_clips("f32")
res.update(clip_s_f32=round(s_f, 4), clips_per_s_f32=round(1 / s_f, 4))
partial("f32 clips")
(audio_x, w_x), s_x = run_clips("bf16x6")
res.update(clip_s_f32=round(s_f, 4), clip_s_bf16x6=round(s_x, 4), clips_per_s_f32=round(1 / s_f, 4),
           clips_per_s_bf16x6=round(1 / s_x, 4), edited_latent_rel_l2=rel(w_x, w_f), waveform_rel_l2=rel(audio_x, audio_f))
if oracle_w is not None:        # the oracle ran the reference step order; the batched schedule differs from it by ~2e-6 (bench.py)
    res.update(latent_rel_l2_vs_cpu_oracle_f32=rel(w_f, oracle_w), latent_rel_l2_vs_cpu_oracle_bf16x6=rel(w_x, oracle_w))
partial("bf16x6 clips")

# the batched forward alone, both arithmetics, same inputs
with torch.inference_mode():
    w0 = m.vae_encode(x0)
ed = m.editor(w0.shape[2], w0.shape[3])
fwd = {}
eps = {}
_inference = torch.inference_mode()          # engine buffers are inference tensors: in-place writes need the mode (round-3 crash)
_inference.__enter__()
for arith in ("f32", "bf16x6"):
    ed.arith = arith
    key = [k for k in ed._unets if k[0] == 2 * a.group and (len(k) == 4) == (arith != "f32")]
    if not key:
        continue
    eng = ed._unets[key[0]]
    g = torch.Generator(device=dev).manual_seed(11)
    eng.x_in.copy_(torch.randn(eng.x_in.shape, device=dev, generator=g))
    st = torch.cuda.current_stream()
    for _ in range(2):
        eng.tape.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5):
        eng.tape.run()
    e1.record(st)
    e1.synchronize()
    fwd[arith] = e0.elapsed_time(e1) / 5
    eps[arith] = eng.eps.float().cpu().clone()
    ops = eng.tape.ops
    res[f"conv_gemm_ops_{arith}"] = dict(total=sum(o.code == L.OP_CONV_GEMM for o in ops),
                                         split_bf16=sum(o.code == L.OP_CONV_GEMM and bool(o.flags & 4) for o in ops))
_inference.__exit__(None, None, None)
if len(fwd) == 2:
    res.update(forward_ms_f32=round(fwd["f32"], 2), forward_ms_bf16x6=round(fwd["bf16x6"], 2),
               forward_speedup=round(fwd["f32"] / fwd["bf16x6"], 3), eps_rel_l2=rel(eps["bf16x6"], eps["f32"]),
               unet_batch=2 * a.group)
res["seconds"] = round(time.time() - t0, 1)
res["checks"] = dict(finite=bool(torch.isfinite(audio_x).all() and torch.isfinite(w_x).all()),
                     latent_within_5e_3=res["edited_latent_rel_l2"] < 5e-3)
print(json.dumps(res))
