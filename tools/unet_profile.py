"""Per-op profile of one AudioLDM2 U-Net forward at batch B (eager, hipGraph replay, HIP-event pair per op) and a JSON
dump of every op's shape + time.  Usage: PYTHONPATH=. python tools/unet_profile.py <B>"""
import torch, time, collections, sys
from audioeditingcode_amd import configs, weights
from audioeditingcode_amd.unet import UNetEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
fam = configs.FAMILIES["audioldm2"]
sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
eng = UNetEngine(fam["unet"], sd, "cuda:0", B, 256, 16, ctx_len0=8, ctx_len1=16)
g = torch.Generator().manual_seed(1)
eng.set_conditioning(ehs0=torch.randn(B,8,768,generator=g), ehs1=torch.randn(B,16,1024,generator=g), bias1=torch.zeros(B,16))
eng.x_in.copy_(torch.randn(B,256,16,8,generator=g)); eng.set_timestep(500)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(3): eng.forward()
    st.synchronize()
    t0 = time.time()
    for _ in range(10): eng.forward()
    st.synchronize()
    dt = (time.time()-t0)/10
    print(f"B={B} eager forward {dt*1e3:.3f} ms  -> {eng.tape.flops/dt/1e12:.1f} TF/s  ({len(eng.tape.ops)} ops)")
    eng.tape.capture()
    for _ in range(3): eng.tape.replay()
    st.synchronize()
    t0 = time.time()
    for _ in range(20): eng.tape.replay()
    st.synchronize()
    dt = (time.time()-t0)/20
    print(f"B={B} graph  forward {dt*1e3:.3f} ms  -> {eng.tape.flops/dt/1e12:.1f} TF/s")
    ms = eng.tape.profile(); ms = eng.tape.profile()
    agg = collections.defaultdict(lambda: [0,0.0,0])
    for m, t in zip(eng.tape.meta, ms):
        key = m["name"].split(".")[-1] if m["code"] != 1 else "conv_gemm:" + m["name"].split(".")[-1]
        a = agg[key]; a[0]+=1; a[1]+=t; a[2]+=m["flops"]
    print(f"sum of per-op ms {sum(ms):.3f}")
    for k,(n,t,f) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:28]:
        print(f"  {k:28s} n={n:4d} {t:8.3f} ms  {f/max(t,1e-9)/1e9:8.1f} TF/s")
    import json
    rows = []
    for m, t, op in zip(eng.tape.meta, ms, eng.tape.ops):
        rows.append(dict(name=m["name"], code=m["code"], ms=t, flops=m["flops"], i=list(op.i), flags=op.flags))
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open(f"gpurun_out/perop_B{B}.json", "w"))
