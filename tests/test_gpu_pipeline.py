"""GPU tests of the clip pipeline (pipeline.ClipPipeline: L clips in flight, each in the reference's step order on its own
HIP stream) and of the CU-masked streams of the C ABI (aed_stream_create_cu_mask / aed_cu_census)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from audioeditingcode_amd import models                                   # noqa: E402
from audioeditingcode_amd.main_run import edit_clip                        # noqa: E402
from audioeditingcode_amd.pipeline import ClipPipeline                     # noqa: E402
from audioeditingcode_amd.streams import PartitionStream                   # noqa: E402
from audioeditingcode_amd.utils import load_audio, synthetic_clip          # noqa: E402

DEV = "cuda:0"
ARGS = (["a dog barking"], ["a cat meowing"], [""], [3.0], [12.0])


def _serial_b(m, mels, T, tstart, seeds, group):
    out = []
    for x0, s in zip(mels, seeds):
        torch.manual_seed(s)
        out.append(edit_clip(m, x0, *ARGS, T, tstart, schedule="batched", timestep_group=group))
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("edit_lanes,launch,overlap", [(1, "graph", True), (2, "graph", True), (1, "eager", True),
                                                        (1, "graph", False)])
def test_partition_pipeline_is_bit_identical_to_one_clip_at_a_time_tiny(edit_lanes, launch, overlap):
    """5 clips through the two-stage partition pipeline (inversion of clip i+1 on CUs [96,256) beside the edit loop of clip
    i on CUs [0,96)) == the same clips one at a time with the same (batched) inversion schedule and per-clip seeds:
    latents and waveforms bit for bit; the waveform -> mel step runs on the front lane's own STFT engine; with and without
    the next clip's preparation on a side stream."""
    T, tstart, G = 10, 6, 5
    m = models.load_model("tiny/audioldm2", DEV, T, seed=0)
    wavs = [synthetic_clip(seconds=1.25, seed=7 + i) for i in range(5)]
    to_mel = lambda view, wav: load_audio((wav, 16000), view.get_fn_STFT(), device=DEV, stft=True)[0]     # noqa: E731
    mels = [to_mel(m, w) for w in wavs]
    seeds = [40 + i for i in range(5)]
    ref = _serial_b(m, mels, T, tstart, seeds, G)
    pipe = ClipPipeline(m, plan="partition", edit_cus=96, edit_lanes=edit_lanes, launch=launch, timestep_group=G,
                        overlap_prep=overlap)
    assert (pipe.workers[0].prep is not None) == (overlap and edit_lanes == 1)       # several lanes: set-up stays on the front lane
    pipe.warm_up(wavs[0], *ARGS, T, tstart, prepare=to_mel)
    got = pipe.edit_clips(wavs, *ARGS, T, tstart, seeds=seeds, prepare=to_mel)
    for i, ((a, o, w), (a2, o2, w2)) in enumerate(zip(got, ref)):
        assert torch.equal(w, w2), (i, float((w - w2).abs().max()))
        assert torch.equal(a, a2) and torch.equal(o, o2), i
    rep = pipe.report()
    # two edit lanes: + the codec stage's worker (VAE decode + vocoder on the inversion partition's queue)
    assert rep["plan"] == "partition" and rep["edit_cus"] == 96
    assert rep["clips_in_flight"] == 1 + edit_lanes + (1 if edit_lanes > 1 else 0)
    if edit_lanes > 1:
        assert rep["device_ms"]["codec_lane"]["n"] == 5 and pipe.workers[-1].lane is pipe.workers[0].lane
    assert rep["device_ms"]["front_chip"]["n"] >= 1               # the first inversion ran on the whole chip (fill)
    assert sum(v["n"] for k, v in rep["device_ms"].items() if k.startswith("back")) == 5
    pipe.close()


def test_edit_lanes_widen_on_drain_and_stay_bit_identical_tiny(monkeypatch):
    """Two edit lanes on disjoint 64-CU slices beside a 128-CU inversion partition: once the front stage has finished its
    last inversion, the remaining edit loops continue on their widened queues (own CUs + a slice of the inversion partition;
    editing.LoopPlumbing._replay_in_chunks).  Same engines and graphs on another stream: results stay bit-identical."""
    from audioeditingcode_amd.editing import EditEngine
    monkeypatch.setattr(EditEngine, "LANE_CHUNK", 1)             # a lane decision per step (6 edit steps per clip here)
    T, tstart, G = 10, 6, 5
    m = models.load_model("tiny/audioldm2", DEV, T, seed=0)
    wavs = [synthetic_clip(seconds=1.25, seed=7 + i) for i in range(5)]
    to_mel = lambda view, wav: load_audio((wav, 16000), view.get_fn_STFT(), device=DEV, stft=True)[0]     # noqa: E731
    mels = [to_mel(m, w) for w in wavs]
    seeds = [40 + i for i in range(5)]
    ref = _serial_b(m, mels, T, tstart, seeds, G)
    pipe = ClipPipeline(m, plan="partition", edit_cus=128, edit_lanes=2, timestep_group=G)
    back = [w for w in pipe.workers if w.stage == "back"]
    assert all(w.wide is not None and len(w.wide.cus) == 128 and set(w.lane.cus) < set(w.wide.cus) for w in back)
    assert not set(back[0].wide.cus) & set(back[1].wide.cus)
    pipe.warm_up(wavs[0], *ARGS, T, tstart, prepare=to_mel)
    got = pipe.edit_clips(wavs, *ARGS, T, tstart, seeds=seeds, prepare=to_mel)
    for i, ((a, o, w), (a2, o2, w2)) in enumerate(zip(got, ref)):
        assert torch.equal(w, w2), (i, float((w - w2).abs().max()))
        assert torch.equal(a, a2) and torch.equal(o, o2), i
    rep = pipe.report()
    # the last clip's loop moved to its widened queue -- or, if the front stage had already drained when that job STARTED (a race
    # of microseconds on this tiny model), ran on the whole chip from its first step: either way nothing idles in the drain
    assert rep["widened_on_drain"] or rep["device_ms"].get("back_chip"), (rep["widened_on_drain"], list(rep["device_ms"]))
    pipe.close()


def test_partition_pipeline_full_size_audioldm2_bit_identical_and_finite():
    """BASELINE config 2's model (346.9 M-parameter U-Net, latent 8x256x16) at a short schedule, 4 clips through the
    partition pipeline (128 | 128 CUs).  The edit partition's engines take their tiles from the 128-CU sweep
    (tile_table_cus128.py), so: (1) pipelined == the same clips pushed through the same engines one per call (nothing
    concurrent, whole chip), bit for bit; (2) against plain main_run.edit_clip the difference is summation order only
    (loop tolerance of the parity tests)."""
    T, tstart, G = 8, 4, 4
    m = models.load_model("cvssp/audioldm2", DEV, T, allow_synthetic=True)
    mels = [load_audio((synthetic_clip(seconds=10.0, seed=3 + i), 16000), m.get_fn_STFT(), device=DEV, stft=True)[0]
            for i in range(4)]
    seeds = [7, 8, 9, 10]
    ref = _serial_b(m, mels, T, tstart, seeds, G)
    pipe = ClipPipeline(m, plan="partition", edit_cus=128, timestep_group=G)
    assert [w.regime for w in pipe.workers] == ["cus128", "cus128"]      # both stages take the tables swept on 128-CU streams
    pipe.warm_up(mels[0], *ARGS, T, tstart)
    got = pipe.edit_clips(mels, *ARGS, T, tstart, seeds=seeds)
    alone = [pipe.edit_clips([x0], *ARGS, T, tstart, seeds=[s])[0] for x0, s in zip(mels, seeds)]
    for i, ((a, o, w), (a1, o1, w1), (a2, o2, w2)) in enumerate(zip(got, alone, ref)):
        assert torch.isfinite(w).all() and torch.isfinite(a).all()
        assert torch.equal(w, w1) and torch.equal(a, a1) and torch.equal(o, o1), (i, float((w - w1).abs().max()))
        err = ((w - w2).norm() / w2.norm()).item()
        assert err < 5e-3, (i, err)
        assert torch.equal(o, o2), i                       # the original's vocoder pass does not touch the U-Net
    # the regime really changed tile choices of the edit engine
    ed_back = pipe.workers[1].view.editor(256, 16)
    ed_plain = m.editor(256, 16)
    tiles = lambda ed: [op.i[29] for eng in ed._unets.values() if eng.B == 2 for op in eng.tape.ops if op.code == 1]   # noqa: E731
    assert tiles(ed_back) != tiles(ed_plain)
    pipe.close()


@pytest.mark.parametrize("lanes,codec_queue,R", [(1, "front", 6), (2, "lane", 20)])
def test_partition_pipeline_full_size_repeated_runs_are_bit_identical(lanes, codec_queue, R):
    """The same 4 clips through the full-size partition pipeline R times WHILE A STRESSOR THREAD keeps split-bf16 VAE-encode
    kernels co-resident on every CU (an unmasked queue of its own): every repeat returns the first run's bits, and the
    inversion's noise maps handed from the front stage to the edit lanes are the same every time.  CFG row sharing is on in BOTH
    loops (round 6): this is the configuration that came out different from run to run in round 5 (first clip of a run, 6 of 14
    runs) until the gather loader of csrc/lin_gemm.hip was fixed (profiles/r06_lin_gather_hazard.md).  20 repeats in the bench's
    layout (two 64-CU lanes, codec on the lanes), 6 in the one-lane layout."""
    import threading
    from audioeditingcode_amd.streams import PartitionStream
    T, tstart, G = 8, 4, 4
    m = models.load_model("cvssp/audioldm2", DEV, T, allow_synthetic=True)
    mels = [load_audio((synthetic_clip(seconds=10.0, seed=3 + i), 16000), m.get_fn_STFT(), device=DEV, stft=True)[0]
            for i in range(4)]
    seeds = [7, 8, 9, 10]
    pipe = ClipPipeline(m, plan="partition", edit_cus=128, edit_lanes=lanes, codec_queue=codec_queue, timestep_group=G)
    pipe.warm_up(mels[0], *ARGS, T, tstart)
    ed_front = pipe.workers[0].view.editor(256, 16)
    assert any(e.S == 2 for e in ed_front._unets.values())                # the inversion's engine shares the context-free head
    for w in pipe.workers:
        if w.stage == "back":                                             # ... and so does every edit lane's batch-2 engine
            assert [e.S for e in w.view.editor(256, 16)._unets.values() if e.B == 2] == [2]
    stash = {}
    orig = pipe._front

    def front(w, st, job, i):
        f = orig(w, st, job, i)
        stash.setdefault(i, []).append((f["zs"].clone(), f["wts"].clone()))
        return f
    pipe._front = front
    first = pipe.edit_clips(mels, *ARGS, T, tstart, seeds=seeds)         # reference run: nothing else on the chip
    torch.cuda.synchronize()
    sv, side, stop, launched = m.lane_view(), PartitionStream.acquire(torch.device(DEV), index=77), threading.Event(), [0]
    with torch.cuda.stream(side.stream):
        sv.vae_encode(mels[0])                                            # builds the stressor's own encoder engine
    torch.cuda.synchronize()

    def stress():
        with torch.inference_mode(), torch.cuda.stream(side.stream):
            while not stop.is_set():
                for _ in range(4):
                    sv.vae_encode(mels[launched[0] % 4])
                    launched[0] += 1
                side.stream.synchronize()
    th = threading.Thread(target=stress, daemon=True)
    th.start()
    try:
        runs = [pipe.edit_clips(mels, *ARGS, T, tstart, seeds=seeds) for _ in range(R)]
    finally:
        stop.set()
        th.join(timeout=60)
    torch.cuda.synchronize()
    assert launched[0] >= 4 * R, launched                                 # the stressor really ran beside every repeat
    for r in range(R):
        for i in range(4):
            assert torch.equal(stash[i][r + 1][0], stash[i][0][0]) and torch.equal(stash[i][r + 1][1], stash[i][0][1]), (r, i)
            for a, b in zip(runs[r][i], first[i]):
                assert torch.equal(a, b), (r, i, float((a - b).abs().max()))
    pipe.close()


def test_codec_on_the_edit_lanes_is_bit_identical_to_one_clip_at_a_time_tiny():
    """codec_queue="lane" (round 5, the bench's default): no codec stage -- the lane that edited a clip runs its VAE decode +
    vocoder on its own 64-CU stream, the next clip's set-up runs on the front stage's side stream: latents and waveforms
    equal the same clips one at a time, bit for bit."""
    T, tstart, G = 10, 6, 5
    m = models.load_model("tiny/audioldm2", DEV, T, seed=0)
    wavs = [synthetic_clip(seconds=1.25, seed=7 + i) for i in range(5)]
    to_mel = lambda view, wav: load_audio((wav, 16000), view.get_fn_STFT(), device=DEV, stft=True)[0]     # noqa: E731
    mels = [to_mel(m, w) for w in wavs]
    seeds = [40 + i for i in range(5)]
    ref = _serial_b(m, mels, T, tstart, seeds, G)
    pipe = ClipPipeline(m, plan="partition", edit_cus=128, edit_lanes=2, timestep_group=G, codec_queue="lane")
    assert [w.stage for w in pipe.workers] == ["front", "back", "back"] and pipe.workers[0].prep is not None
    pipe.warm_up(wavs[0], *ARGS, T, tstart, prepare=to_mel)
    got = pipe.edit_clips(wavs, *ARGS, T, tstart, seeds=seeds, prepare=to_mel)
    for i, ((a, o, w), (a2, o2, w2)) in enumerate(zip(got, ref)):
        assert torch.equal(w, w2), (i, float((w - w2).abs().max()))
        assert torch.equal(a, a2) and torch.equal(o, o2), i
    rep = pipe.report()
    assert rep["clips_in_flight"] == 3 and "codec_lane" not in rep["device_ms"]
    pipe.close()


def test_cu_masked_streams_census_and_results():
    """aed_stream_create_cu_mask: a contiguous range of 8k mask bits is k CUs on each of the 8 XCDs (aed_cu_census reads
    the hardware's XCC / SE / CU ids); a hipGraph replayed on a masked stream gives the same values as on a plain one."""
    from audioeditingcode_amd import configs, weights
    from audioeditingcode_amd.unet import UNetEngine
    full = PartitionStream.acquire(DEV)
    assert PartitionStream.acquire(DEV) is full                   # process-lifetime streams come from a cache
    total = full.total
    assert len(full.census()) == total
    low = PartitionStream.acquire(DEV, cus=range(64))
    cs = low.census()
    per_xcc = {}
    for x, se, sh, cu in cs:
        per_xcc[x] = per_xcc.get(x, 0) + 1
    assert len(cs) == 64 and sorted(per_xcc) == list(range(8)) and set(per_xcc.values()) == {8}, per_xcc
    rest = PartitionStream.acquire(DEV, cus=range(64, total))
    assert len(rest.census()) == total - 64 and not set(rest.census()) & set(cs)
    fam = configs.tiny_family("audioldm2")
    sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
    eng = UNetEngine(fam["unet"], sd, DEV, 2, 32, 16, ctx_len0=8, ctx_len1=8)
    g = torch.Generator().manual_seed(1)
    eng.set_conditioning(ehs0=torch.randn(2, 8, 48, generator=g), ehs1=torch.randn(2, 8, 64, generator=g),
                         bias1=torch.zeros(2, 8))
    eng.x_in.copy_(torch.randn(2, 32, 16, 8, generator=g))
    eng.set_timestep(500)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eng.forward()
        st.synchronize()
        ref = eng.eps.clone()
        eng.tape.capture()
    for ps in (low, rest):
        eng.eps.zero_()
        with torch.cuda.stream(ps.stream):
            eng.tape.replay()
        ps.stream.synchronize()
        assert torch.equal(eng.eps, ref)
    with pytest.raises(ValueError):
        PartitionStream(DEV, cus=[total])
    for ps in (full, low, rest):
        ps.close()
