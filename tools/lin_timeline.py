"""In-kernel s_memtime timelines (shader cycles) of the GEMM kernels on the contraction shapes of the batch-2 U-Net,
cold weights (a 512 MB flush precedes the launch) and warm.  For lin_gemm tiles (>= 10): first and last workgroup,
wave 0: [start, prefetch issued, chunk 1 done, chunk 2 done, ..., partials written, barrier passed, stores issued];
for the tiled kernels: block 0 only.   PYTHONPATH=. python tools/lin_timeline.py"""
import torch
from audioeditingcode_amd.tape import Tape
from audioeditingcode_amd.unet import geglu_pack_index

DEV = "cuda:0"


def run(M, N, K, tile, taps=1, geglu=0, ln=0, res=False, ks=1):
    Cin = K // taps
    if taps == 9:
        H = 16 if M % 16 == 0 else 8
        B_, W_ = 1, M // H
        A = torch.randn(B_, H, W_, Cin, device=DEV)
    else:
        A = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) * 0.05
    n_out = N // 2 if geglu else N
    out = torch.empty(M, n_out, device=DEV)
    bias = torch.randn(N, device=DEV)
    R = torch.randn(M, n_out, device=DEV) if res else None
    rs = W.sum(1).contiguous() if ln else None
    dbg = torch.zeros(32, dtype=torch.int64, device=DEV)
    tp = Tape(DEV)
    if taps == 9:
        tp.conv(A, W, bias, out, B=1, IH=H, IW=W_, Cin=Cin, OH=H, OW=W_, N=N, KH=3, KW=3, pad_h=1, pad_w=1, tile=tile,
                ksplit=ks, res=R)
    else:
        tp.linear(A, W, bias, out, M=M, K=K, N=N, tile=tile, ksplit=ks, res=R, ln_rowsum=rs, geglu=geglu)
    tp.ops[0].p[7] = dbg.data_ptr()
    tp.ops[0].flags |= 1
    tp.finalize()
    for cold in (True, False):
        for _ in range(2):
            if cold:
                flush = torch.empty(128 * 1024 * 1024, device=DEV).fill_(1.0)    # noqa: F841
            dbg.zero_()
            torch.cuda.synchronize()
            tp.run()
            torch.cuda.synchronize()
        d = dbg.cpu().tolist()
        tag = f"M{M} N{N} K{K} t{taps} g{geglu} l{ln} tile{tile}:{ks} {'cold' if cold else 'warm'}"
        if tile >= 10:
            f = [x for x in d[:15] if x]
            l_ = [x for x in d[16:31] if x]
            t0 = f[0]
            print(f"{tag}\n   first WG: {[x - t0 for x in f]}\n   last  WG: {[x - t0 for x in l_]}", flush=True)
        else:
            t = [x for x in d[:30] if x]
            print(f"{tag}\n   block 0: {[x - t[0] for x in t]}", flush=True)


run(128, 5120, 640, 13, geglu=1, ln=1)
run(128, 5120, 640, 15, geglu=1, ln=1)
run(2048, 2048, 256, 15, geglu=1, ln=1)
run(2048, 2048, 256, 1, geglu=1, ln=1)
run(512, 3072, 384, 13, geglu=1, ln=1)
run(128, 640, 640, 12, res=True)
run(128, 640, 640, 11, res=True)
run(2048, 256, 256, 10, res=True)
run(512, 384, 384, 12, res=True)
run(2048, 768, 256, 10, ln=1)
run(2048, 768, 256, 4, ln=1)
run(128, 640, 2560, 11, res=True)
run(128, 640, 5760, 12, taps=9)
run(2048, 256, 2304, 10, taps=9)
run(8192, 128, 1152, 4, taps=9)
