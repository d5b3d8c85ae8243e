"""s_memtime timeline of the wave-split-K kernel on the small GEMM shapes of the batch-2 U-Net (cold and warm caches)."""
import torch, time
from audioeditingcode_amd.tape import Tape
DEV="cuda:0"
def run(M,N,K,tile,ks=1,res=True):
    A = torch.randn(M,K,device=DEV); W = torch.randn(N,K,device=DEV)*0.05; out = torch.empty(M,N,device=DEV)
    bias = torch.randn(N,device=DEV); R = torch.randn(M,N,device=DEV) if res else None
    dbg = torch.zeros(32, dtype=torch.int64, device=DEV)
    tp = Tape(DEV)
    tp.linear(A,W,bias,out,M=M,K=K,N=N,tile=tile,ksplit=ks,res=R)
    tp.ops[0].p[7] = dbg.data_ptr(); tp.ops[0].flags |= 1; tp.finalize()
    for cold in (True, False):
        for _ in range(2):
            if cold: flush = torch.empty(128*1024*1024, device=DEV).fill_(1.0)
            torch.cuda.synchronize(); tp.run(); torch.cuda.synchronize()
        d = dbg.cpu().tolist(); t = [x for x in d[:30] if x]
        rel = [t[i+1]-t[i] for i in range(len(t)-1)]
        print(f"M{M} N{N} K{K} tile{tile} ks{ks} {'cold' if cold else 'warm'}: stamp deltas {rel} total {t[-1]-t[0]}", flush=True)
run(128,640,640,7)
run(2048,256,256,7)
run(512,384,384,7)
run(128,640,2560,7,3)
run(128,640,640,4,4)
