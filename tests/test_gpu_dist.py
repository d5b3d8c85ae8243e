"""RCCL on real hardware: a ONE-rank `nccl` process group on cuda:0 executes exactly the collectives the
8-rank run makes -- the frozen AudioLDM2 weights as device-resident arenas through `broadcast_state_dict(on_device=True)`,
the per-rank gather of edited latents, the max-over-ranks all-reduce and the barrier (audioeditingcode_amd/dist.py; the reference
is single-GPU, main_run.py:72-73).  Runs in a child process: a process group is process-global state."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
import torch
from audioeditingcode_amd import configs, dist as adist, weights
rank, world, local = adist.init_distributed(force=True)
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" and (rank, world) == (0, 1)
dev = torch.device("cuda:0")
fam = configs.get_family("cvssp/audioldm2")
shapes = dict(unet=weights.unet_param_shapes(fam["unet"]), vae=weights.vae_param_shapes(fam["vae"]),
              vocoder=weights.vocoder_param_shapes(fam["vocoder"]))
sds = {k: weights.random_state_dict(shapes[k], seed=i) for i, k in enumerate(shapes)}
want = {k: adist.state_checksum(v) for k, v in sds.items()}
t0 = time.time()
got = {k: adist.broadcast_state_dict(sds[k], shapes[k], dev, on_device=True) for k in shapes}
torch.cuda.synchronize()
t_b = time.time() - t0
assert all(v.is_cuda for sd in got.values() for v in sd.values())
have = {k: adist.state_checksum({n: t.cpu() for n, t in v.items()}) for k, v in got.items()}   # same summation order as `want`
lat = torch.randn(3, 8, 256, 16, device=dev)
g = adist.gather_to_rank0(lat)
mx = adist.max_over_ranks(1.25, dev)
adist.barrier()
torch.cuda.synchronize()
print(json.dumps(dict(want=want, have=have, broadcast_s=t_b, n_params=sum(v.numel() for sd in got.values() for v in sd.values()),
                      gathered=len(g), gather_equal=bool(torch.equal(g[0], lat)), mx=mx,
                      port=os.environ.get("MASTER_PORT"))))
torch.distributed.destroy_process_group()
"""


def test_one_rank_rccl_group_moves_the_full_audioldm2_weights_and_gathers_latents():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print("one-rank RCCL group:", {k: d[k] for k in ("broadcast_s", "n_params", "mx", "port")})
    for k in d["want"]:
        assert d["have"][k] == d["want"][k], (k, d["have"][k], d["want"][k])       # fp64 checksums of identical fp32 values
    assert d["gathered"] == 1 and d["gather_equal"] and d["mx"] == 1.25
    assert d["n_params"] > 4.0e8                                                   # U-Net 346.9 M + VAE + vocoder
    assert d["port"] not in (None, "29511")                                         # no fixed rendezvous port any more
