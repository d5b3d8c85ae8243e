"""Diagnostic 6 (round 5): name the first perturbed buffer of the batch-2 edit engine with CFG row sharing.  The edit loop of ONE
clip (hipGraph replays, back lane of the partition pipeline) is repeated from identical inputs while VAE encodes run on the
unmasked side stream and a helper thread draws noise on the CPU; after a perturbed repeat every buffer of the engine is compared
with the solo run's.  If the engine's input of the LAST step (x_in) still equals the solo run's, the differing buffers belong to
the last step alone and the earliest writer among them is the culprit node."""
import sys
import threading
import traceback

import torch

sys.path.insert(0, ".")
from audioeditingcode_amd import editing, models, tape as tape_mod          # noqa: E402
from audioeditingcode_amd.pipeline import ClipPipeline                     # noqa: E402
from audioeditingcode_amd.utils import load_audio, synthetic_clip          # noqa: E402

DEV = "cuda:0"
ARGS = (["a dog barking"], ["a cat meowing"], [""], [3.0], [12.0])
T, tstart, G = 8, 4, 4


def main(share=True, modes=(("vae + helper", True, True),), N=8, detail=True):
    editing.EditEngine.SHARE_IN_EDIT_LOOP = share
    m = models.load_model("cvssp/audioldm2", DEV, T, allow_synthetic=True)
    mel = load_audio((synthetic_clip(seconds=10.0, seed=3), 16000), m.get_fn_STFT(), device=DEV, stft=True)[0]
    pipe = ClipPipeline(m, plan="partition", edit_cus=128, timestep_group=G)
    pipe.warm_up(mel, *ARGS, T, tstart)
    fw, bw = pipe.workers[0], pipe.workers[1]
    stash = {}
    orig = pipe._front

    def front(w, st, job, i):
        f = orig(w, st, job, i)
        stash["f"] = dict(x0=f["x0"], zs=f["zs"].clone(), wts=f["wts"].clone())
        return f
    pipe._front = front
    pipe.edit_clips([mel], *ARGS, T, tstart, seeds=[7])
    pipe._front = orig
    torch.cuda.synchronize()
    f0 = stash["f"]
    ed = bw.view.editor(256, 16)
    eng = [e for e in ed._unets.values() if e.B == 2 and e.S == (2 if share else 1)][0]
    plan = [p for k, p in ed._plans.items() if k[0] == "edit"][0]
    bufs = [("keep", j, t) for j, t in enumerate(eng.tape.keep) if torch.is_tensor(t) and t.is_floating_point() and t.numel()]
    bufs += [("tmp", k, t) for k, t in eng._tmp.items()]
    bufs += [("plan", k, plan[k]) for k in ("cur", "zs", "coef") if torch.is_tensor(plan.get(k))]
    writers = {}
    for idx, op in enumerate(eng.tape.ops):
        for s in range(10):
            if op.p[s]:
                writers.setdefault(int(op.p[s]), []).append((idx, s, eng.tape.meta[idx]["name"]))
    job = pipe._job([mel], [7], None, pipe._args(*ARGS, T, tstart, 1.0))

    def edit_once():
        with tape_mod.tile_regime(bw.regime), pipe._on(bw, bw.lane) as st:
            done = pipe.event_type()
            done.record(st)
            out = pipe._back(bw, st, job, dict(x0=f0["x0"], zs=f0["zs"], wts=f0["wts"], done=done), with_codec=False)
        return out["w_edit"]
    w_ref = edit_once()
    torch.cuda.synchronize()
    ref = [t.clone() for _, _, t in bufs]
    xin_ref = eng.x_in.clone()
    w2 = edit_once()
    torch.cuda.synchronize()
    print("solo repeat identical:", torch.equal(w_ref, w2), all(torch.equal(a, b[2]) for a, b in zip(ref, bufs)), flush=True)
    side = fw.prep.stream
    for label, use_vae, use_helper in modes:
        nbad = 0
        for it in range(N):
            stop = threading.Event()

            def helper():
                while not stop.is_set():
                    torch.stack([torch.randn(1, 8, 256, 16) for _ in range(T)]).pin_memory()
            th = threading.Thread(target=helper, daemon=True)
            if use_helper:
                th.start()
            if use_vae:
                with torch.cuda.stream(side):
                    for _ in range(2 + it % 3):
                        fw.view.vae_encode(mel)
            w = edit_once()
            torch.cuda.synchronize()
            stop.set()
            if use_helper:
                th.join()
            if torch.equal(w, w_ref):
                continue
            nbad += 1
            if not detail:
                continue
            last_step_only = torch.equal(eng.x_in, xin_ref)
            rows = []
            for (kind, key, t), r in zip(bufs, ref):
                if not torch.equal(t, r):
                    ws = writers.get(t.data_ptr(), [])
                    rows.append((min((w_[0] for w_ in ws), default=-1), kind, str(key)[:40], tuple(t.shape),
                                 round(float((t - r).abs().max()), 4), int((t != r).sum()), [w_[2] for w_ in ws][:2]))
            rows.sort()
            print(f"iter {it}: |w - ref| {float((w - w_ref).abs().max()):.3g}; x_in of the last step equals the solo run's: "
                  f"{last_step_only}; {len(rows)} of {len(bufs)} buffers differ; earliest writers:", flush=True)
            for r_ in rows[:8]:
                print("   ", r_, flush=True)
        print(f"share={share}, stress = {label}: perturbed repeats {nbad} of {N}", flush=True)
    pipe.close()


if __name__ == "__main__":
    try:
        with torch.inference_mode():
            if len(sys.argv) > 1 and sys.argv[1] == "attribute":
                main(True, (("VAE encodes only", True, False), ("CPU helper thread only", False, True), ("VAE + helper", True, True)),
                     N=8, detail=False)
                torch.cuda.empty_cache()
                main(False, (("VAE + helper", True, True),), N=8, detail=False)
            else:
                main()
    except BaseException:                       # noqa: BLE001
        traceback.print_exc()
