"""Clip-level data parallelism over the GPUs of one node (not in the reference, which is
single-GPU: main_run.py:72-73).  One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI
on ROCm; "gloo" for the CPU tests).

Clips are independent (no cross-clip term anywhere in inversion_utils.py), so the data path has NO
collective: clip i -> rank i mod W.  Collectives exist only at the edges:
  * one broadcast of the frozen weights from rank 0 (only rank 0 touches disk): the state dicts are
    flattened into a few large contiguous fp32 arenas -- xGMI is point-to-point and ring broadcasts are
    per-link bound, so few large messages beat ~1700 small ones;
  * one gather of the edited latents ([n_local, 8, 256, 16] fp32 = 128 KiB per clip) to rank 0.
"""
import os
from typing import Dict, List

import torch
import torch.distributed as dist


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def init_distributed(backend=None, force=False):
    """Initialise from the torchrun environment; returns (rank, world, local_rank).  At world size 1 nothing is initialised
    unless `force` (a one-rank RCCL group: lets a single MI355X execute the same broadcast / all_gather / all_reduce calls
    the 8-rank run makes, tests/test_gpu_dist.py and `torchrun --nproc-per-node 1 bench.py`).  A launcher always provides
    MASTER_PORT; without one (world 1 only) a free port is taken -- two jobs on one box must not meet on a fixed port."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or force or "TORCHELASTIC_RUN_ID" in os.environ) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise RuntimeError("WORLD_SIZE > 1 without MASTER_PORT: launch the ranks with torch.distributed.run "
                                   "(bench.py --gpus N does) or export one port for all of them")
            os.environ["MASTER_PORT"] = str(_free_port())
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def local_world_size(world: int) -> int:
    """Ranks on THIS node: torchrun's LOCAL_WORLD_SIZE, else the global world size (one node)."""
    try:
        n = int(os.environ.get("LOCAL_WORLD_SIZE", "") or world)
    except ValueError:
        n = world
    return max(1, min(n, max(1, world)))


def pin_rank_resources(local_rank: int, world: int, threads: int = None, affinity: bool = True):
    """Host-side budget of one rank on a node that runs `world` of them -- `world` here is the number of ranks PER NODE
    (`local_world_size`): with the global world size a two-node job would leave half of every node's cores unused.  Per rank the clip pipeline
    keeps <= 5 host threads busy (front / back / codec workers, the noise-prefetch thread, the caller) and <= 5 HIP queues
    (inversion lane, edit lane(s), whole-chip fill / drain queue, side stream); torch's intra-op CPU pool (the per-clip RNG
    draws, the host-side scheduler tables) would otherwise default to ALL cores in every rank.  Caps torch's intra-op threads
    at cores / world (at least 1, at most 16; `threads` overrides) and, where the OS allows it, pins the process to the
    rank's contiguous slice of the visible cores.  Returns what it did (bench.py prints it in `config.rank_resources`)."""
    cores = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    per = max(1, len(cores) // max(1, world))
    n = int(threads) if threads else max(1, min(16, per))
    torch.set_num_threads(n)
    pinned = None
    if affinity and world > 1 and hasattr(os, "sched_setaffinity") and len(cores) >= world:
        mine = cores[local_rank * per:(local_rank + 1) * per]
        try:
            os.sched_setaffinity(0, mine)
            pinned = [mine[0], mine[-1]]
        except OSError:
            pinned = None
    return dict(torch_threads=n, cores_visible=len(cores), cores_per_rank=per, affinity=pinned)


def broadcast_family(sds, shapes: Dict[str, Dict[str, tuple]], device, on_device: bool = True):
    """What every rank of `bench.py --gpus N` does before building its wrapper: receive the frozen component weights (U-Net, VAE,
    vocoder) from rank 0 -- one `broadcast_state_dict` per component, in the order of `shapes` on every rank.  `sds` is rank 0's
    {component: state dict} (None elsewhere).  Returns ({component: state dict on `device`}, seconds, bytes received)."""
    import time
    t0 = time.time()
    out = {k: broadcast_state_dict(None if sds is None else sds[k], shapes[k], device, on_device=on_device) for k in shapes}
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    nbytes = 4 * sum(_arena_shapes(shapes[k])[1] for k in shapes)
    return out, time.time() - t0, nbytes


def per_rank_rates(units: float, seconds: float, device) -> List[float]:
    """all_gather of every rank's own rate (units / seconds BEFORE the closing barrier): shows whether one rank lags; the
    whole-job value is computed from the max-over-ranks time by the caller, not from these."""
    mine = torch.tensor([units / max(seconds, 1e-9)], dtype=torch.float64, device=device)
    allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allr, mine)
    return [float(t.item()) for t in allr]


def distributed_fields(world: int, t_bcast: float, nbytes: int, per_rank, rank_resources) -> Dict:
    """The `config` fields of the bench line that describe the N-rank run; asserted by tests/test_dist_cpu.py so that the first
    real 8-rank run prints a usable line."""
    assert per_rank is None or (len(per_rank) == world and all(r > 0 for r in per_rank)), per_rank
    return dict(weights_broadcast_s=t_bcast, weights_broadcast_GB=nbytes / 1e9,
                weights_broadcast_GBps=(nbytes / 1e9 / t_bcast) if t_bcast > 0 else None,
                per_rank_clips_per_s=per_rank, rank_resources=rank_resources,
                slowest_rank_over_fastest=(min(per_rank) / max(per_rank)) if per_rank else None)


def shard_clips(n_clips: int, rank: int, world: int) -> List[int]:
    """Round-robin clip -> rank map (clip i runs on rank i mod W)."""
    return list(range(rank, n_clips, world))


def _arena_shapes(shapes: Dict[str, tuple]):
    offs, total = {}, 0
    for k, shp in shapes.items():
        n = 1
        for d in shp:
            n *= d
        offs[k] = (total, n, tuple(shp))
        total += n
    return offs, total


def broadcast_state_dict(sd, shapes: Dict[str, tuple], device, src: int = 0, max_bucket: int = 1 << 28,
                         on_device: bool = False):
    """Broadcast one component's weights as contiguous fp32 arenas of <= max_bucket elements (1 GiB).
    `sd` is the real state dict on `src` and may be None elsewhere; `shapes` (name -> shape, identical and
    identically ordered on every rank) comes from weights.*_param_shapes.  Returns the state dict.
    on_device=True: the returned tensors are views of the received arenas on `device` (no host bounce: the engines'
    weight packers consume device tensors directly); False keeps the round-1 behaviour (CPU copies)."""
    if not (dist.is_available() and dist.is_initialized()):
        return sd
    rank = dist.get_rank()
    offs, total = _arena_shapes(shapes)
    out = {}
    names = list(shapes)
    i = 0
    while i < len(names):
        j, n = i, 0
        while j < len(names) and (n == 0 or n + offs[names[j]][1] <= max_bucket):
            n += offs[names[j]][1]
            j += 1
        arena = torch.empty(n, dtype=torch.float32, device=device)
        if rank == src:
            p = 0
            for k in names[i:j]:
                cnt = offs[k][1]
                arena[p:p + cnt].copy_(sd[k].reshape(-1))
                p += cnt
        dist.broadcast(arena, src=src)
        p = 0
        flat = arena if on_device else arena.cpu()
        for k in names[i:j]:
            cnt, shp = offs[k][1], offs[k][2]
            out[k] = flat[p:p + cnt].view(shp) if on_device else flat[p:p + cnt].view(shp).clone()
            p += cnt
        i = j
    return out


def state_checksum(sd) -> float:
    """Order-independent fp64 checksum used to verify the broadcast (world-size invariance tests)."""
    return float(sum(v.double().sum().item() for v in sd.values()))


def gather_to_rank0(local: torch.Tensor, dst: int = 0):
    """Gather equally-shaped per-rank tensors on `dst`; returns the list there, None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()):
        return [local]
    world = dist.get_world_size()
    if dist.get_backend() == "nccl":
        bufs = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(bufs, local.contiguous())
        return bufs if dist.get_rank() == dst else None
    bufs = [torch.empty_like(local) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(local.contiguous(), bufs, dst=dst)
    return bufs


def max_over_ranks(x: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
