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
