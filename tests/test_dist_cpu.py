"""world_size=2 gloo tests of the clip-sharding path (CPU, multi-process)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from audioeditingcode_amd import configs, dist as adist, weights
    r, w, _ = adist.init_distributed(backend="gloo")
    fam = configs.tiny_family("audioldm2")
    shapes = weights.unet_param_shapes(fam["unet"])
    sd = weights.random_state_dict(shapes, seed=3) if r == 0 else None
    # tiny bucket size forces several arenas
    got = adist.broadcast_state_dict(sd, shapes, "cpu", src=0, max_bucket=200_000)
    ck = adist.state_checksum(got)
    # the no-host-bounce form (views of the received arenas) must carry the same weights
    got_dev = adist.broadcast_state_dict(sd, shapes, "cpu", src=0, max_bucket=200_000, on_device=True)
    assert adist.state_checksum(got_dev) == ck and all(got_dev[k].shape == got[k].shape for k in got)
    # the Stable Audio components travel the same way (three state dicts: DiT, Oobleck, projection model)
    sa = configs.get_family("tiny/stable-audio-open-1.0")
    sa_shapes = dict(transformer=weights.dit_param_shapes(sa["dit"]), vae=weights.oobleck_param_shapes(sa["oobleck"]),
                     projection_model=weights.projection_param_shapes(sa["projection"]))
    sa_sd = {k: weights.random_state_dict(v, seed=7 + i) for i, (k, v) in enumerate(sa_shapes.items())} if r == 0 else None
    sa_got = {k: adist.broadcast_state_dict(None if sa_sd is None else sa_sd[k], sa_shapes[k], "cpu", max_bucket=300_000,
                                            on_device=True) for k in sa_shapes}
    sa_ck = sum(adist.state_checksum(v) for v in sa_got.values())
    mine = adist.shard_clips(7, r, w)
    local = torch.full((len(adist.shard_clips(8, r, w)), 8, 4, 4), float(r))
    gathered = adist.gather_to_rank0(local)
    mx = adist.max_over_ranks(1.0 + r, "cpu")
    adist.barrier()
    q.put((r, ck, mine, None if gathered is None else [float(g.mean()) for g in gathered], mx, list(got)[:3], sa_ck))


def test_broadcast_shard_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ck0, m0, g0, mx0, k0, sa0), (r1, ck1, m1, g1, mx1, k1, sa1) = res
    assert ck0 == ck1 and k0 == k1 and sa0 == sa1          # identical weights on both ranks after the broadcast
    assert m0 == [0, 2, 4, 6] and m1 == [1, 3, 5]           # clip i -> rank i mod W; every clip exactly once
    assert g0 == [0.0, 1.0] and g1 is None                  # latents gathered on rank 0 in rank order
    assert mx0 == mx1 == 2.0
    sys.path.insert(0, ROOT)
    from audioeditingcode_amd import configs, weights
    ref = weights.random_state_dict(weights.unet_param_shapes(configs.tiny_family("audioldm2")["unet"]), seed=3)
    from audioeditingcode_amd.dist import state_checksum
    assert ck0 == state_checksum(ref)


# ------------------------------------------------------------------------------------------------------------------------
# bench-level world-size invariance: what `bench.py --gpus N` does per rank -- receive the weights from
# rank 0, build the wrapper from them, edit the rank's share of the clips (clip i -> rank i mod W, per-clip seeds), gather the
# edited latents on rank 0 -- on the CPU stack (tapes executed by oracle/tape_interp.py), world 2 against world 1.
CLIPS, T_, TSTART_ = 4, 3, 2


def _edit_share(rank, world, state_dicts):
    """Edit this rank's clips with the product's wrapper (CPU test subclass) and return their latents in clip order."""
    from _pytest.monkeypatch import MonkeyPatch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import install_cpu_stack
    from audioeditingcode_amd import dist as adist, models
    from audioeditingcode_amd.main_run import edit_clip
    from audioeditingcode_amd.utils import load_audio, synthetic_clip
    mpatch = MonkeyPatch()
    install_cpu_stack(mpatch)

    class CpuAudioLDM2(models.AudioLDM2Wrapper):
        def _require_device(self):           # test instrumentation only: the product class refuses a CPU device
            pass
    torch.set_num_threads(2)
    m = CpuAudioLDM2(model_id="tiny/audioldm2", device="cpu", state_dicts=state_dicts)
    m.load_scheduler()
    m.model.scheduler.set_timesteps(T_, device=None)
    out = []
    for i in adist.shard_clips(CLIPS, rank, world):
        x0 = load_audio((synthetic_clip(seconds=0.32, seed=7 + i), 16000), m.get_fn_STFT(), device="cpu", stft=True)[0]
        torch.manual_seed(40 + i)
        out.append(edit_clip(m, x0, ["a dog barking"], ["a cat meowing"], [""], [3.0], [12.0], T_, TSTART_,
                             schedule="batched", timestep_group=3)[2])
    mpatch.undo()
    return torch.cat(out, 0)


def _family_state_dicts(seed_base=0):
    from audioeditingcode_amd import configs, weights
    fam = configs.get_family("tiny/audioldm2")
    shapes = dict(unet=weights.unet_param_shapes(fam["unet"]), vae=weights.vae_param_shapes(fam["vae"]),
                  vocoder=weights.vocoder_param_shapes(fam["vocoder"]))
    return shapes, {k: weights.random_state_dict(shapes[k], seed=seed_base + i) for i, k in enumerate(shapes)}


def _bench_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from audioeditingcode_amd import dist as adist
    r, w, local = adist.init_distributed(backend="gloo")
    adist.pin_rank_resources(local, w, threads=2)                     # per-rank thread cap + CPU affinity (bench.py does the same)
    import time
    shapes, sds = _family_state_dicts()
    sds, t_bcast, nbytes = adist.broadcast_family(sds if r == 0 else None, shapes, "cpu", on_device=False)     # bench.py's call
    t0 = time.perf_counter()
    lat = _edit_share(r, w, sds)
    dt_local = time.perf_counter() - t0
    gathered = adist.gather_to_rank0(lat)
    rates = adist.per_rank_rates(len(adist.shard_clips(CLIPS, r, w)), dt_local, "cpu")                      # bench.py's call
    fields = adist.distributed_fields(w, t_bcast, nbytes, rates, dict(torch_threads=2))
    adist.barrier()
    q.put((r, None if gathered is None else [g.numpy() for g in gathered], fields))


def test_bench_level_world_size_invariance_world2_vs_world1():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    res = {r: g for r, g, _ in got}
    fields = {r: f for r, _, f in got}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None and len(res[0]) == 2
    # the N-rank fields of the bench line (bench.py builds them with the same three calls): every rank saw both rates, the
    # broadcast moved the whole family, nothing is None that the first real 8-rank run needs
    for r in (0, 1):
        f = fields[r]
        assert len(f["per_rank_clips_per_s"]) == 2 and min(f["per_rank_clips_per_s"]) > 0
        assert f["per_rank_clips_per_s"] == fields[0]["per_rank_clips_per_s"]
        assert f["weights_broadcast_s"] > 0 and f["weights_broadcast_GB"] > 0 and f["weights_broadcast_GBps"] > 0
        assert 0 < f["slowest_rank_over_fastest"] <= 1
    sys.path.insert(0, ROOT)
    _, sds = _family_state_dicts()
    one = _edit_share(0, 1, sds)                                      # world 1: the same four clips in this process
    two = torch.empty_like(one)
    two[0::2], two[1::2] = torch.from_numpy(res[0][0]), torch.from_numpy(res[0][1])          # clip i ran on rank i mod 2
    assert torch.isfinite(two).all() and torch.equal(one, two)       # every clip's result is independent of the world size
