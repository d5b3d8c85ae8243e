"""Diagnostic (round 5): the full-size partition pipeline with and without CFG row sharing -- is the pipelined result
deterministic, does it equal the clip alone, how far are both from the one-clip-at-a-time path."""
import sys

import torch

sys.path.insert(0, ".")
from audioeditingcode_amd import editing, models                          # noqa: E402
from audioeditingcode_amd.main_run import edit_clip                        # noqa: E402
from audioeditingcode_amd.pipeline import ClipPipeline                     # noqa: E402
from audioeditingcode_amd.utils import load_audio, synthetic_clip          # noqa: E402

DEV = "cuda:0"
ARGS = (["a dog barking"], ["a cat meowing"], [""], [3.0], [12.0])
T, tstart, G = 8, 4, 4


def run(share, **kw):
    editing.EditEngine.SHARE_CFG_ROWS = share
    m = models.load_model("cvssp/audioldm2", DEV, T, allow_synthetic=True)
    mels = [load_audio((synthetic_clip(seconds=10.0, seed=3 + i), 16000), m.get_fn_STFT(), device=DEV, stft=True)[0]
            for i in range(4)]
    seeds = [7, 8, 9, 10]
    ref = []
    for x0, s in zip(mels, seeds):
        torch.manual_seed(s)
        ref.append(edit_clip(m, x0, *ARGS, T, tstart, schedule="batched", timestep_group=G)[2])
    pipe = ClipPipeline(m, plan="partition", edit_cus=128, timestep_group=G, **kw)
    pipe.warm_up(mels[0], *ARGS, T, tstart)
    A = [r[2] for r in pipe.edit_clips(mels, *ARGS, T, tstart, seeds=seeds)]
    B = [r[2] for r in pipe.edit_clips(mels, *ARGS, T, tstart, seeds=seeds)]
    al = [pipe.edit_clips([x0], *ARGS, T, tstart, seeds=[s])[0][2] for x0, s in zip(mels, seeds)]
    al2 = [pipe.edit_clips([x0], *ARGS, T, tstart, seeds=[s])[0][2] for x0, s in zip(mels, seeds)]
    mx = lambda a, b: float((a - b).abs().max())                           # noqa: E731
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())   # noqa: E731
    print(dict(share=share, **kw), flush=True)
    for i in range(4):
        print(f"  clip {i}: |A-B| {mx(A[i], B[i]):.3g}  |A-alone| {mx(A[i], al[i]):.3g}  |alone-alone2| {mx(al[i], al2[i]):.3g}  "
              f"rel(A,ref) {rel(A[i], ref[i]):.3g}  rel(alone,ref) {rel(al[i], ref[i]):.3g}", flush=True)
    pipe.close()
    del pipe, m
    torch.cuda.empty_cache()


if __name__ == "__main__":
    with torch.inference_mode():
        run(True)
        run(True, overlap_prep=False)
        run(True, widen_on_drain=False)
        run(False)
