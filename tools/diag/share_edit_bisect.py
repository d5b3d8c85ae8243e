"""Diagnostic 7 (round 6): NAME the first perturbed node of the batch-2 edit engine with CFG row sharing.

One clip's edit loop on the 128-CU back lane, replayed from identical inputs ONE STEP GRAPH AT A TIME: after every replay the lane
stream is synchronised and every buffer of the engine is compared with the same step of the solo run, so the first buffer that
differs is the output of the first perturbed node (not of a later step that consumed it).  Stress = VAE encodes on the unmasked
side stream, enqueued right before every replay.

    python tools/diag/share_edit_bisect.py [N=8] [enc=3] [launch=graph|eager] [head=lin|x6] [side=unmasked|masked] [share=1|0]

`head=x6` forces the shared head's gather-mode lin_gemm convolutions (tiles 11 / 17 at M = 1024) onto conv_gemm_x6 tiles (kernel
vs runtime); `side=masked` confines the stressor to the inversion partition's CUs; the runtime's graph path is switched by the
environment (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0) from the lease script."""
import os
import sys
import traceback

import torch

sys.path.insert(0, ".")
from audioeditingcode_amd import editing, models, tape as tape_mod          # noqa: E402
from audioeditingcode_amd import _lib as L                                  # noqa: E402
from audioeditingcode_amd.pipeline import ClipPipeline                     # noqa: E402
from audioeditingcode_amd.tape import Tape                                 # noqa: E402
from audioeditingcode_amd.utils import load_audio, synthetic_clip          # noqa: E402

DEV = "cuda:0"
ARGS = (["a dog barking"], ["a cat meowing"], [""], [3.0], [12.0])
T, tstart, G = 8, 4, 4


def opts():
    o = dict(N="8", enc="3", launch="graph", head="lin", side="unmasked", share="1")
    for a in sys.argv[1:]:
        k, _, v = a.partition("=")
        o[k] = v
    return o


def describe(t, r):
    """Where two versions of one buffer differ: element count, flat range, rows / columns of the [rows, C] view."""
    d = (t != r)
    nz = d.reshape(-1).nonzero().reshape(-1)
    C = t.shape[-1] if t.dim() > 1 else 1
    rows, cols = nz // C, nz % C
    return dict(n=int(nz.numel()), of=int(t.numel()), max=float((t - r).abs().max()),
                rows=(int(rows.min()), int(rows.max()), int(rows.unique().numel())),
                cols=(int(cols.min()), int(cols.max()), int(cols.unique().numel())))


def main():
    o = opts()
    N, n_enc, share = int(o["N"]), int(o["enc"]), o["share"] == "1"
    editing.EditEngine.SHARE_IN_EDIT_LOOP = share
    if o["head"] == "x6":
        conv0 = Tape.conv

        def conv(self, x, w, bias, out, **kw):
            M = kw["B"] * kw["OH"] * kw["OW"]
            if M == 1024 and kw.get("KH", 1) * kw.get("KW", 1) > 1 and not kw.get("tile"):
                kw["tile"] = 4          # LDS-staged 64x64 block tile -> conv_gemm_x6 under the bf16x6 arithmetic
            return conv0(self, x, w, bias, out, **kw)
        Tape.conv = conv
    m = models.load_model("cvssp/audioldm2", DEV, T, allow_synthetic=True)
    mel = load_audio((synthetic_clip(seconds=10.0, seed=3), 16000), m.get_fn_STFT(), device=DEV, stft=True)[0]
    pipe = ClipPipeline(m, plan="partition", edit_cus=128, timestep_group=G)
    if o["side"] == "masked":
        from audioeditingcode_amd.streams import PartitionStream
        pipe.workers[0].prep = PartitionStream.acquire(torch.device(DEV), cus=range(128, 256), total=256, index=18)
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
    bufs += [("plan", k, plan[k]) for k in ("cur",) if torch.is_tensor(plan.get(k))]
    seen, uniq = set(), []
    for b in bufs:                                  # one entry per storage
        if b[2].data_ptr() not in seen:
            seen.add(b[2].data_ptr())
            uniq.append(b)
    bufs = uniq
    # output slot of a conv_gemm record is p[3]; for the other opcodes every pointer slot is listed and the LAST writer wins below
    touch = {}
    for idx, op in enumerate(eng.tape.ops):
        for s in range(10):
            if op.p[s]:
                touch.setdefault(int(op.p[s]), []).append((idx, s))
    names = [mm["name"] for mm in eng.tape.meta]
    print(f"engine: B={eng.B} share={eng.S} ops={len(eng.tape.ops)} buffers={len(bufs)} "
          f"lin-tile ops at M=1024: {sum(1 for op in eng.tape.ops if op.code == L.OP_CONV_GEMM and op.i[0] == 1024 and op.i[29] >= 10)}",
          flush=True)
    job = pipe._job([mel], [7], None, pipe._args(*ARGS, T, tstart, 1.0))
    side = fw.prep.stream
    state = dict(hook=None)

    def stepwise(body, steps, use_graph=True, plan=None):
        cur = torch.cuda.current_stream(ed.device)
        stream = ed.loop_stream()
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            g = plan.get("graph")
            if g is None:
                g = plan["graph"] = Tape.graph_capture(body)
            for k in range(steps):
                if state["stress"]:
                    with torch.cuda.stream(side):
                        for _ in range(n_enc):
                            fw.view.vae_encode(mel)
                if o["launch"] == "graph":
                    Tape.graph_replay(g)
                else:
                    body()
                stream.synchronize()
                if state["hook"](k):
                    break
        torch.cuda.synchronize()
        cur.wait_stream(stream)
    ed._run_graph = stepwise

    def edit_once():
        with tape_mod.tile_regime(bw.regime), pipe._on(bw, bw.lane) as st:
            done = pipe.event_type()
            done.record(st)
            pipe._back(bw, st, job, dict(x0=f0["x0"], zs=f0["zs"], wts=f0["wts"], done=done), with_codec=False)

    ref = []

    def record(k):
        ref.append([t.clone() for _, _, t in bufs])
        return False
    state.update(hook=record, stress=False)
    edit_once()
    # solo repeat must reproduce every step
    miss = []

    def check_solo(k):
        miss.extend((k, j) for j, (b, r) in enumerate(zip(bufs, ref[k])) if not torch.equal(b[2], r))
        return False
    state.update(hook=check_solo, stress=False)
    edit_once()
    print("solo repeat identical at every step:", not miss, flush=True)

    first_nodes = {}
    nbad = 0
    for it in range(N):
        found = {}

        def compare(k):
            rows = []
            for (kind, key, t), r in zip(bufs, ref[k]):
                if not torch.equal(t, r):
                    ws = touch.get(t.data_ptr(), [])
                    rows.append((min((w_[0] for w_ in ws), default=-1), kind, str(key)[:32], tuple(t.shape), describe(t, r),
                                 [(w_[0], w_[1], names[w_[0]]) for w_ in ws][:4]))
            if rows:
                rows.sort(key=lambda r_: r_[0])
                found.update(step=k, rows=rows)
                return True
            return False
        state.update(hook=compare, stress=True)
        edit_once()
        if found:
            nbad += 1
            r0 = found["rows"][0]
            first_nodes[(r0[0], r0[5][0][2] if r0[5] else "?")] = first_nodes.get((r0[0], r0[5][0][2] if r0[5] else "?"), 0) + 1
            print(f"repeat {it}: first perturbed step {found['step']}; {len(found['rows'])} of {len(bufs)} buffers differ; earliest:",
                  flush=True)
            for r_ in found["rows"][:6]:
                print("    ", r_, flush=True)
            idx = r0[0]
            if idx >= 0:
                op = eng.tape.ops[idx]
                print(f"     op {idx} '{names[idx]}' code={op.code} flags={op.flags} i[:40]={[int(v) for v in op.i[:40]]}", flush=True)
                if idx > 0:
                    opp = eng.tape.ops[idx - 1]
                    print(f"     op {idx - 1} '{names[idx - 1]}' code={opp.code} flags={opp.flags} i[:40]={[int(v) for v in opp.i[:40]]}",
                          flush=True)
    env = {k: v for k, v in os.environ.items() if k.startswith(("DEBUG_", "HIP_", "ROC_", "GPU_", "AMD_"))}
    print(f"RESULT {o} env={env}: perturbed repeats {nbad} of {N}; first perturbed nodes {first_nodes}", flush=True)
    pipe.close()


if __name__ == "__main__":
    try:
        with torch.inference_mode():
            main()
    except BaseException:                       # noqa: BLE001
        traceback.print_exc()
