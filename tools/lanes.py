"""Lanes, measured properly (wall clock, execution counted): host cost of issuing one batch-2 U-Net forward as a
hipGraphLaunch vs launch by launch (aed_tape_run), and the chip time per forward of L concurrent lanes for
{graph, eager} x {one host thread, one host thread per lane}.

    PYTHONPATH=. python tools/lanes.py [Lmax] [n] -> gpurun_out/lanes.json"""
import json
import os
import sys
import threading
import time

import torch

from audioeditingcode_amd import configs, weights
from audioeditingcode_amd.tape import Tape
from audioeditingcode_amd.unet import PackedUNetWeights, UNetEngine

LMAX = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = "cuda:0"
fam = configs.FAMILIES["audioldm2"]
sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
pw = PackedUNetWeights(sd, dev)
gen = torch.Generator().manual_seed(1)


def mk(B):
    eng = UNetEngine(fam["unet"], pw, dev, B, 256, 16, ctx_len0=8, ctx_len1=16)
    eng.set_conditioning(ehs0=torch.randn(B, 8, 768, generator=gen), ehs1=torch.randn(B, 16, 1024, generator=gen),
                         bias1=torch.zeros(B, 16))
    eng.x_in.copy_(torch.randn(B, 256, 16, 8, generator=gen))
    eng.set_timestep(500)
    return eng


engs = [mk(2) for _ in range(LMAX)]
streams = [torch.cuda.Stream() for _ in range(LMAX)]
counters = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(LMAX)]
advs, graphs = [], []
for e, s, c in zip(engs, streams, counters):
    adv = Tape(dev)
    adv.advance(c)
    adv.finalize()
    advs.append(adv)
    with torch.cuda.stream(s):
        e.forward()
        s.synchronize()
        graphs.append(Tape.graph_capture(lambda e=e, adv=adv: (e.tape.run(), adv.run())))
        s.synchronize()
ref = engs[0].eps.clone()
out = {"n_per_lane": N, "algorithmic_gflop_per_forward": engs[0].tape.flops / 1e9,
       "executed_gflop_per_forward": engs[0].tape.exec_flops / 1e9, "launches_per_forward": len(engs[0].tape.ops)}


def issue(k, mode, n):
    with torch.cuda.stream(streams[k]):
        for _ in range(n):
            if mode == "graph":
                Tape.graph_replay(graphs[k])
            else:
                engs[k].tape.run()
                advs[k].run()


# ---- host cost of issuing ONE forward on an idle stream (the call returns when everything is enqueued)
for mode in ("graph", "eager"):
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        issue(0, mode, 1)
        ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    ts.sort()
    out[f"host_ms_per_forward_{mode}"] = 1e3 * ts[len(ts) // 2]
    print(f"host cost of issuing one forward ({mode}): median {1e3 * ts[len(ts) // 2]:.3f} ms, min {1e3 * ts[0]:.3f} ms",
          flush=True)

# ---- L lanes: wall clock from first enqueue to all done; every lane must have executed exactly N forwards
res = {}
for mode in ("graph", "eager"):
    for host in ("1thread", "threads"):
        for L in [l for l in (1, 2, 3, 4, 6, 8) if l <= LMAX]:
            for c in counters:
                c.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if host == "1thread":
                # round-robin so that no lane's backlog is enqueued long before another's
                for _ in range(N):
                    for k in range(L):
                        issue(k, mode, 1)
                t_enq = time.perf_counter() - t0
            else:
                ths = [threading.Thread(target=issue, args=(k, mode, N)) for k in range(L)]
                for th in ths:
                    th.start()
                for th in ths:
                    th.join()
                t_enq = time.perf_counter() - t0
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            counts = [int(c[0]) for c in counters[:L]]
            ok = all(v == N for v in counts) and all(torch.equal(e.eps, ref) for e in engs[:L])
            chip = 1e3 * wall / (L * N)
            res[f"{mode}_{host}_L{L}"] = dict(wall_ms=1e3 * wall, enqueue_ms=1e3 * t_enq, chip_ms_per_forward=chip,
                                             lane_ms_per_forward=1e3 * wall / N, executed_ok=ok)
            print(f"{mode:5s} {host:7s} L={L}: wall {1e3 * wall:8.1f} ms (host enqueue {1e3 * t_enq:8.1f} ms) -> "
                  f"{1e3 * wall / N:7.3f} ms per lane-forward, {chip:6.3f} ms of chip time per forward "
                  f"({out['executed_gflop_per_forward'] / chip:6.1f} TF/s executed)  ok={ok}", flush=True)
out["runs"] = res
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/lanes.json", "w"), indent=1)
