"""Diagnostic 2 (round 5): which loop of the full-size partition pipeline is perturbed with CFG row sharing, under which
variation.  Repeats the 4-clip run R times and compares every repeat with the first: the inversion's noise maps (front
payload) and the edited latent."""
import sys

import torch

sys.path.insert(0, ".")
from audioeditingcode_amd import editing, models, tape as tape_mod          # noqa: E402
from audioeditingcode_amd.pipeline import ClipPipeline                     # noqa: E402
from audioeditingcode_amd.utils import load_audio, synthetic_clip          # noqa: E402

DEV = "cuda:0"
ARGS = (["a dog barking"], ["a cat meowing"], [""], [3.0], [12.0])
T, tstart, G = 8, 4, 4
SW = dict(inv=True, edit=True)
orig_unet = editing.EditEngine.unet


def unet(self, B, L0=0, L1=0, share=1):
    if (B == 2 and not SW["edit"]) or (B > 2 and not SW["inv"]):
        share = 1
    return orig_unet(self, B, L0, L1, share)


editing.EditEngine.unet = unet
orig_front = ClipPipeline._front
STASH = {}


def front(self, w, st, job, i):
    f = orig_front(self, w, st, job, i)
    STASH.setdefault(i, []).append((f["zs"].clone(), f["wts"].clone()))
    return f


ClipPipeline._front = front
orig_pick = ClipPipeline._pick_lane


def run(label, inv=True, edit=True, wide=1, lane_only=False, R=5, **kw):
    SW.update(inv=inv, edit=edit)
    tape_mod.WIDE_CHUNKS = wide
    ClipPipeline._pick_lane = (lambda self, w, job, s: w.lane) if lane_only else orig_pick
    m = models.load_model("cvssp/audioldm2", DEV, T, allow_synthetic=True)
    mels = [load_audio((synthetic_clip(seconds=10.0, seed=3 + i), 16000), m.get_fn_STFT(), device=DEV, stft=True)[0]
            for i in range(4)]
    seeds = [7, 8, 9, 10]
    pipe = ClipPipeline(m, plan="partition", edit_cus=128, timestep_group=G, **kw)
    pipe.warm_up(mels[0], *ARGS, T, tstart)
    STASH.clear()
    runs = [[r[2] for r in pipe.edit_clips(mels, *ARGS, T, tstart, seeds=seeds)] for _ in range(R)]
    torch.cuda.synchronize()
    mx = lambda a, b: float((a - b).abs().max())                           # noqa: E731
    bad = []
    for r in range(1, R):
        for i in range(4):
            dz, dx, dw = mx(STASH[i][r][0], STASH[i][0][0]), mx(STASH[i][r][1], STASH[i][0][1]), mx(runs[r][i], runs[0][i])
            if dz or dx or dw:
                # first perturbed timestep of the inversion (zs rows are in loop order)
                zr = (STASH[i][r][0] - STASH[i][0][0]).flatten(1).abs().amax(1)
                bad.append((r, i, f"zs {dz:.3g} xts {dx:.3g} w {dw:.3g}", [round(float(v), 4) for v in zr]))
    print(label, "->", bad if bad else "all repeats identical", flush=True)
    pipe.close()
    del pipe, m
    torch.cuda.empty_cache()


if __name__ == "__main__":
    with torch.inference_mode():
        run("share inv+edit")
        run("share inv only", edit=False)
        run("share edit only", inv=False)
        run("share inv+edit, narrow chunks", wide=0)
        run("share inv+edit, front always on its lane", lane_only=True)
        run("no share", inv=False, edit=False)
