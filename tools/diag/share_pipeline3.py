"""Diagnostic 4 (round 5): which part of the next clips' set-up, running on the unmasked side stream while the first clip's edit
loop runs, perturbs the batch-2 edit engine with CFG row sharing (EditEngine.SHARE_IN_EDIT_LOOP).  Variants of ClipPipeline._front:
only the VAE encode on the side stream / only the x_t draws (noise upload + sample kernel + text conditioning) on it / everything
on it but no helper thread for the noise."""
import sys

import torch

sys.path.insert(0, ".")
from audioeditingcode_amd import editing, models                          # noqa: E402
from audioeditingcode_amd.ddm_inversion.inversion_utils import prepare_forward, run_forward    # noqa: E402
from audioeditingcode_amd.pipeline import ClipPipeline                     # noqa: E402
from audioeditingcode_amd.utils import load_audio, synthetic_clip          # noqa: E402

DEV = "cuda:0"
ARGS = (["a dog barking"], ["a cat meowing"], [""], [3.0], [12.0])
T, tstart, G = 8, 4, 4
MODE = {"m": "all"}
orig_front = ClipPipeline._front


def front(self, w, st, job, i):
    mode = MODE["m"]
    if mode == "all" or w.prep is None:
        return orig_front(self, w, st, job, i)
    v, a = w.view, job["a"]
    x0 = job["items"][i]
    ps = w.prep.stream
    ev = self.event_type
    if mode == "vae_only":
        with self._stream_ctx(ps):
            w0 = v.vae_encode(x0)
        ready = ev()
        ready.record(ps)
        st.wait_event(ready)
        w0.record_stream(st)
        prepared = prepare_forward(v, w0, a["src"], a["cfg_src"], a["T"])
    else:                                   # "xts_only"
        w0 = v.vae_encode(x0)
        got = ev()
        got.record(st)
        ps.wait_event(got)
        w0.record_stream(ps)
        with self._stream_ctx(ps):
            prepared = prepare_forward(v, w0, a["src"], a["cfg_src"], a["T"])
        ready = ev()
        ready.record(ps)
        st.wait_event(ready)
        conds = [getattr(c, n) for c in (prepared["cond_src"], prepared["cond_unc"]) if c is not None
                 for n in ("ehs0", "ehs1", "mask0", "mask1", "class_labels")]
        for t in (prepared["xts0"], *conds):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(st)
    _, zs, wts, _ = run_forward(v, w0, prepared, a["eta"], a["cfg_src"], True, a["schedule"], a["group"])
    done = ev()
    done.record(st)
    return dict(x0=x0, zs=zs, wts=wts, done=done)


ClipPipeline._front = front
orig_job = ClipPipeline._job


def run(label, mode="all", prefetch=True, R=6, masked_prep=False, **kw):
    MODE["m"] = mode
    editing.EditEngine.SHARE_IN_EDIT_LOOP = True

    def job(self, items, seeds, prepare, a):
        j = orig_job(self, items, seeds, prepare, a)
        if not prefetch:
            j["uniform"] = False
        return j
    ClipPipeline._job = job
    m = models.load_model("cvssp/audioldm2", DEV, T, allow_synthetic=True)
    mels = [load_audio((synthetic_clip(seconds=10.0, seed=3 + i), 16000), m.get_fn_STFT(), device=DEV, stft=True)[0]
            for i in range(4)]
    seeds = [7, 8, 9, 10]
    pipe = ClipPipeline(m, plan="partition", edit_cus=128, timestep_group=G, **kw)
    if masked_prep:                     # the side stream confined to the inversion partition's CUs
        from audioeditingcode_amd.streams import PartitionStream
        pipe.workers[0].prep = PartitionStream.acquire(torch.device(DEV), cus=range(128, 256), total=256, index=18)
    pipe.warm_up(mels[0], *ARGS, T, tstart)
    runs = [[r[2] for r in pipe.edit_clips(mels, *ARGS, T, tstart, seeds=seeds)] for _ in range(R)]
    torch.cuda.synchronize()
    bad = [(r, i, round(float((runs[r][i] - runs[0][i]).abs().max()), 3)) for r in range(1, R) for i in range(4)
           if not torch.equal(runs[r][i], runs[0][i])]
    print(label, "->", bad if bad else "all repeats identical", flush=True)
    pipe.close()
    del pipe, m
    torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "first"
    with torch.inference_mode():
        if which == "first":
            run("side stream: VAE encode only", mode="vae_only")
            run("side stream: x_t draws + text only", mode="xts_only")
            run("side stream: everything, no noise helper thread", prefetch=False)
        else:       # the reliable reproducer (VAE encode only on the side stream) under three changes
            run("VAE-only side stream MASKED to the inversion partition", mode="vae_only", masked_prep=True, R=4)
            run("VAE-only side stream, edit lanes launched eagerly (no hipGraph)", mode="vae_only", launch="eager", R=4)
            run("VAE-only side stream, no noise helper thread", mode="vae_only", prefetch=False, R=4)
