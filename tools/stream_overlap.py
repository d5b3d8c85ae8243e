"""Premise test: B=2 forwards (latency-bound) and B=40 forwards (throughput-bound) on two streams at once."""
import torch, time
from audioeditingcode_amd import configs, weights
from audioeditingcode_amd.unet import UNetEngine, PackedUNetWeights
fam = configs.FAMILIES["audioldm2"]
sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
pw = PackedUNetWeights(sd, "cuda:0")
g = torch.Generator().manual_seed(1)
def mk(B):
    eng = UNetEngine(fam["unet"], pw, "cuda:0", B, 256, 16, ctx_len0=8, ctx_len1=16)
    eng.set_conditioning(ehs0=torch.randn(B,8,768,generator=g), ehs1=torch.randn(B,16,1024,generator=g), bias1=torch.zeros(B,16))
    eng.x_in.copy_(torch.randn(B,256,16,8,generator=g)); eng.set_timestep(500)
    return eng
e2, e40 = mk(2), mk(40)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1):
    e2.forward(); s1.synchronize(); e2.tape.capture()
with torch.cuda.stream(s2):
    e40.forward(); s2.synchronize(); e40.tape.capture()
def run(n2, n40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s2):
        for _ in range(n40): e40.tape.replay()
    with torch.cuda.stream(s1):
        for _ in range(n2): e2.tape.replay()
    s1.synchronize(); t1 = time.perf_counter(); s2.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t0) * 1e3
run(5, 1)
print("B=2 x100 alone: %.1f ms" % run(100, 0)[0])
print("B=40 x10 alone: %.1f ms" % run(0, 10)[1])
a, b = run(100, 10); print("together: B=2 stream done %.1f ms, all done %.1f ms" % (a, b))
a, b = run(100, 4); print("together 100+4: B=2 stream done %.1f ms, all done %.1f ms" % (a, b))
