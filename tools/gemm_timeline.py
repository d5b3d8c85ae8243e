"""conv_gemm on large GEMM shapes: TF/s (20 back-to-back launches) and the in-kernel s_memtime timeline of block 0
(op flag 1 + p[7] = int64[32] device buffer).  Usage on the GPU box: PYTHONPATH=. python tools/gemm_timeline.py"""
import torch, time
from audioeditingcode_amd.tape import Tape
DEV="cuda:0"
def run(M,N,K,tile,ks=1):
    A = torch.randn(M,K,device=DEV); W = torch.randn(N,K,device=DEV)*0.05; out = torch.empty(M,N,device=DEV)
    bias = torch.randn(N,device=DEV)
    dbg = torch.zeros(32, dtype=torch.int64, device=DEV)
    tp = Tape(DEV)
    tp.linear(A,W,bias,out,M=M,K=K,N=N,tile=tile,ksplit=ks)
    tp.ops[0].p[7] = dbg.data_ptr(); tp.ops[0].flags |= 1; tp.finalize()
    for _ in range(3):
        torch.cuda.synchronize(); tp.run(); torch.cuda.synchronize()
    d = dbg.cpu().tolist(); t = [x for x in d[:30] if x]
    rel = [t[i+1]-t[i] for i in range(len(t)-1)]
    tp2 = Tape(DEV); 
    for _ in range(20): tp2.linear(A,W,bias,out,M=M,K=K,N=N,tile=tile,ksplit=ks)
    tp2.finalize(); tp2.run(); torch.cuda.synchronize(); t0=time.perf_counter(); tp2.run(); torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
    print(f"M{M} N{N} K{K} tile{tile}: {dt*1e6:.1f} us {2*M*N*K/dt/1e12:.1f} TF/s  stamp deltas {rel}", flush=True)
run(163840,128,1152,1)
run(40960,2048,256,1)
run(8192,8192,1024,1)
run(8192,8192,1024,2)
run(8192,8192,1024,4)
