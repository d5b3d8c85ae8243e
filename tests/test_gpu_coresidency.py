"""Kernels of one queue must not change their results when workgroups of ANOTHER queue share their compute units.

Round 6 named the node behind round 5's run-to-run differences (CFG row sharing in the edit loop): the gather loader of the
latency-regime GEMM (csrc/lin_gemm.hip, tiles 11 / 15 -- the 72 KB-LDS configurations that fit beside a split-bf16 GEMM
workgroup on a CU) produced wrong values in lanes 48-63 of single loader rows whenever split-bf16 workgroups were co-resident
(profiles/r06_lin_gather_hazard.md).  These tests keep the two reproducers in the suite: the standalone one (one op record on a
CU-masked stream, R launches under a stressor on an unmasked stream) and the engine-level one (the CFG-shared batch-2 edit
engine, one step graph at a time, every buffer compared with the solo run)."""
import sys

import pytest
import torch

sys.path.insert(0, ".")
pytestmark = pytest.mark.gpu

from audioeditingcode_amd import editing, models, tape as tape_mod          # noqa: E402
from audioeditingcode_amd.pipeline import ClipPipeline                     # noqa: E402
from audioeditingcode_amd.streams import PartitionStream                   # noqa: E402
from audioeditingcode_amd.tape import Tape                                 # noqa: E402
from audioeditingcode_amd.utils import load_audio, synthetic_clip          # noqa: E402

DEV = torch.device("cuda:0")


def _stressor(gen):
    """A VAE-encoder-like 3x3 convolution (128 -> 128 channels on a 1024 x 64 map) in the split-bf16 arithmetic, four launches."""
    tp = Tape(DEV)
    x = torch.randn(1, 1024, 64, 128, generator=gen, device=DEV)
    w = torch.randn(128, 9 * 128, generator=gen, device=DEV) / (9 * 128) ** 0.5
    out = torch.empty(1, 1024, 64, 128, device=DEV)
    with tape_mod.arith_mode("bf16x6"):
        for _ in range(4):
            tp.conv(x, w, None, out, B=1, IH=1024, IW=64, Cin=128, OH=1024, OW=64, N=128, KH=3, KW=3, pad_h=1, pad_w=1)
    tp.finalize()
    tp.keep += [x, w, out]
    return tp


# (tile, B, IH, IW, Cin, N, k, stride, loader activation): the shared head's own shapes, the two tiles that were perturbed, two that
# were not, the uniform loader, both loader activations
CASES = [(11, 1, 256, 16, 128, 128, 3, 2, 0), (11, 1, 128, 8, 128, 256, 3, 1, 0), (15, 1, 128, 8, 256, 256, 3, 1, 0),
         (11, 2, 64, 4, 384, 384, 3, 1, 0), (15, 2, 64, 4, 384, 384, 3, 1, 0), (10, 1, 128, 8, 256, 256, 3, 1, 0),
         (17, 1, 128, 8, 256, 256, 3, 1, 0), (11, 1, 128, 8, 256, 256, 1, 1, 0), (11, 1, 128, 8, 256, 256, 3, 1, 1),
         (15, 1, 128, 8, 256, 256, 3, 1, 2)]


@pytest.mark.parametrize("tile,B,IH,IW,Cin,N,k,stride,act", CASES)
def test_lin_gemm_launches_are_identical_under_a_co_resident_split_bf16_stressor(tile, B, IH, IW, Cin, N, k, stride, act):
    """R = 200 launches of one lin_gemm record on a 128-CU masked stream while the stressor runs on an unmasked stream: every
    output equals the solo launch bit for bit (round-5 kernel: ~50 % of the launches of tiles 11 / 15 differed), and the solo
    launch matches torch's convolution."""
    R = 200
    gen = torch.Generator(device=DEV)
    gen.manual_seed(tile * 1000 + k)
    lane = PartitionStream.acquire(DEV, cus=range(0, 128), total=256, index=0)
    side = PartitionStream.acquire(DEV, index=17)
    stress = _stressor(gen)
    pad = k // 2
    OH, OW = (IH + 2 * pad - k) // stride + 1, (IW + 2 * pad - k) // stride + 1
    x = torch.randn(B, IH, IW, Cin, generator=gen, device=DEV)
    w = torch.randn(N, k * k * Cin, generator=gen, device=DEV) / (k * k * Cin) ** 0.5
    b = torch.randn(N, generator=gen, device=DEV)
    outs = torch.zeros(R + 1, B, OH, OW, N, device=DEV)
    tp = Tape(DEV)
    with tape_mod.arith_mode("f32"):
        for r in range(R + 1):
            tp.conv(x, w, b, outs[r], B=B, IH=IH, IW=IW, Cin=Cin, OH=OH, OW=OW, N=N, KH=k, KW=k, stride=stride, pad_h=pad,
                    pad_w=pad, tile=tile, in_act=act, in_slope=0.1)
    tp.finalize()
    assert all(op.i[29] == tile for op in tp.ops)
    with torch.cuda.stream(lane.stream):
        tp.run(0, 1)
    torch.cuda.synchronize()
    with torch.cuda.stream(side.stream):
        for _ in range(R // 8):
            stress.run()
    with torch.cuda.stream(lane.stream):
        tp.run(1, R + 1)
    torch.cuda.synchronize()
    bad = [r for r in range(1, R + 1) if not torch.equal(outs[r], outs[0])]
    assert not bad, (len(bad), R, float((outs[bad[0]] - outs[0]).abs().max()))
    xa = x if act == 0 else (torch.nn.functional.silu(x) if act == 1 else torch.nn.functional.leaky_relu(x, 0.1))
    ref = torch.nn.functional.conv2d(xa.permute(0, 3, 1, 2).double(), w.view(N, k, k, Cin).permute(0, 3, 1, 2).double(), b.double(),
                                     stride=stride, padding=pad).permute(0, 2, 3, 1)
    rel = float((outs[0].double() - ref).norm() / ref.norm())
    assert rel < 2e-6, rel


def test_cfg_shared_edit_engine_steps_are_identical_under_co_resident_vae_encodes():
    """The batch-2 edit engine with the CFG-shared head (full-size AudioLDM2, T = 8, tstart = 4) on the 128-CU back lane, ONE
    step graph at a time: after every replay all ~1340 buffers of the engine equal the same step of the solo run although three
    VAE encodes are enqueued on an unmasked stream before every replay (round-5 kernel: 14 of 16 repeats perturbed, first
    buffer = the output of `down_blocks.0.downsamplers.0.conv`)."""
    T, tstart, G, N = 8, 4, 4, 6
    args = (["a dog barking"], ["a cat meowing"], [""], [3.0], [12.0])
    assert editing.EditEngine.SHARE_IN_EDIT_LOOP and editing.EditEngine.SHARE_CFG_ROWS
    with torch.inference_mode():
        m = models.load_model("cvssp/audioldm2", "cuda:0", T, allow_synthetic=True)
        mel = load_audio((synthetic_clip(seconds=10.0, seed=3), 16000), m.get_fn_STFT(), device="cuda:0", stft=True)[0]
        pipe = ClipPipeline(m, plan="partition", edit_cus=128, timestep_group=G, mask_prep=False)
        pipe.warm_up(mel, *args, T, tstart)
        fw, bw = pipe.workers[0], pipe.workers[1]
        stash = {}
        orig = pipe._front

        def front(w, st, job, i):
            f = orig(w, st, job, i)
            stash["f"] = dict(x0=f["x0"], zs=f["zs"].clone(), wts=f["wts"].clone())
            return f
        pipe._front = front
        pipe.edit_clips([mel], *args, T, tstart, seeds=[7])
        pipe._front = orig
        torch.cuda.synchronize()
        f0 = stash["f"]
        ed = bw.view.editor(256, 16)
        eng = [e for e in ed._unets.values() if e.B == 2][0]
        assert eng.S == 2
        plan = [p for k, p in ed._plans.items() if k[0] == "edit"][0]
        seen, bufs = set(), []
        for t in [t for t in eng.tape.keep if torch.is_tensor(t) and t.is_floating_point() and t.numel()] + \
                list(eng._tmp.values()) + [plan["cur"]]:
            if t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                bufs.append(t)
        assert len(bufs) > 1000
        job = pipe._job([mel], [7], None, pipe._args(*args, T, tstart, 1.0))
        side = fw.prep.stream
        state = {}

        def stepwise(body, steps, use_graph=True, plan=None):
            cur = torch.cuda.current_stream(ed.device)
            stream = ed.loop_stream()
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                g = plan["graph"]
                for k in range(steps):
                    if state["stress"]:
                        with torch.cuda.stream(side):
                            for _ in range(3):
                                fw.view.vae_encode(mel)
                    Tape.graph_replay(g)
                    stream.synchronize()
                    state["hook"](k)
            torch.cuda.synchronize()
            cur.wait_stream(stream)
        ed._run_graph = stepwise

        def edit_once():
            with tape_mod.tile_regime(bw.regime), pipe._on(bw, bw.lane) as st:
                done = pipe.event_type()
                done.record(st)
                pipe._back(bw, st, job, dict(x0=f0["x0"], zs=f0["zs"], wts=f0["wts"], done=done), with_codec=False)
        ref = []
        state.update(stress=False, hook=lambda k: ref.append([t.clone() for t in bufs]))
        edit_once()
        assert len(ref) == tstart
        differing = []
        state.update(stress=True, hook=lambda k: differing.extend((k, j) for j, (t, r) in enumerate(zip(bufs, ref[k]))
                                                                   if not torch.equal(t, r)))
        for _ in range(N):
            edit_once()
        assert not differing, (len(differing), differing[:4])
        del ed._run_graph
        pipe.close()
