"""Diagnostic 3 (round 5): the batch-2 U-Net engine with CFG row sharing (prefix at batch 1) on a 128-CU lane while VAE encodes
run on an unmasked queue: which buffer of the forward differs from the solo run first."""
import sys

import torch

sys.path.insert(0, ".")
from audioeditingcode_amd import models, tape as tape_mod                  # noqa: E402
from audioeditingcode_amd.pipeline import ClipPipeline                     # noqa: E402
from audioeditingcode_amd.utils import load_audio, synthetic_clip          # noqa: E402

DEV = "cuda:0"
ARGS = (["a dog barking"], ["a cat meowing"], [""], [3.0], [12.0])
T, tstart, G = 8, 4, 4


def main(share_key, N=30):
    m = models.load_model("cvssp/audioldm2", DEV, T, allow_synthetic=True)
    mel = load_audio((synthetic_clip(seconds=10.0, seed=3), 16000), m.get_fn_STFT(), device=DEV, stft=True)[0]
    pipe = ClipPipeline(m, plan="partition", edit_cus=128, timestep_group=G)
    pipe.warm_up(mel, *ARGS, T, tstart)
    back, front = pipe.workers[1], pipe.workers[0]
    ed = back.view.editor(256, 16)
    eng = [e for k, e in ed._unets.items() if e.B == 2 and (("share2" in k) == share_key)][0]
    print("engine", [k for k, e in ed._unets.items() if e is eng], "S", eng.S, flush=True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 256, 16, 8, generator=g).to(DEV)
    eng.x_in.copy_(x.expand(2, -1, -1, -1))
    eng.set_timestep(501)
    bufs = [t for t in eng.tape.keep if torch.is_tensor(t) and t.is_floating_point() and t.numel() > 0]
    bufs += [t for t in eng._tmp.values()]
    ptr_ops = {}
    for idx, op in enumerate(eng.tape.ops):
        for s in range(10):
            if op.p[s]:
                ptr_ops.setdefault(int(op.p[s]), []).append((idx, s, eng.tape.meta[idx]["name"]))
    lane, side = back.lane.stream, front.prep.stream if front.prep is not None else None
    print("side stream", side, flush=True)

    def fwd():
        with torch.cuda.stream(lane):
            eng.forward()
    fwd()
    torch.cuda.synchronize()
    ref = [b.clone() for b in bufs]
    fwd()
    torch.cuda.synchronize()
    print("solo repeat identical:", all(torch.equal(a, b) for a, b in zip(ref, bufs)), flush=True)
    fv = front.view
    nbad = 0
    for it in range(N):
        with torch.cuda.stream(side):
            for _ in range(3):
                fv.vae_encode(mel)
        for _ in range(4):
            fwd()
        torch.cuda.synchronize()
        diff = [(j, float((a - b).abs().max())) for j, (a, b) in enumerate(zip(ref, bufs)) if not torch.equal(a, b)]
        if diff:
            nbad += 1
            rows = []
            for j, d in diff:
                writers = [w for w in ptr_ops.get(bufs[j].data_ptr(), [])]
                first = min((w[0] for w in writers), default=-1)
                rows.append((first, j, tuple(bufs[j].shape), d, [w[2] for w in writers][:3]))
            rows.sort()
            print(f"iter {it}: {len(diff)} buffers differ; earliest:", rows[:6], flush=True)
    print("share" if share_key else "no share", "bad iterations", nbad, "of", N, flush=True)
    pipe.close()


if __name__ == "__main__":
    with torch.inference_mode():
        main(True)
