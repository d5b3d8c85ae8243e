"""Microbench conv_gemm per shape/config with cold weights (pool > MALL)."""
import torch, sys, time, math
from audioeditingcode_amd.tape import Tape
DEV = "cuda:0"
shapes = [(40960,2048,256),(40960,256,256),(10240,3072,384),(40960,768,256),(40960,256,1024),(2560,5120,640),(10240,1152,384),(10240,384,384),(2560,640,640),(2560,1920,640),(2560,640,5760),(2560,640,2560),(10240,384,1536),(10240,384,3456),(40960,256,2304),(163840,128,1152)]
_old = [(16384, 256, 256), (16384, 768, 256), (16384, 2048, 256), (16384, 256, 1024), (4096, 384, 384), (4096, 3072, 384), (1024, 640, 640), (1024, 5120, 640), (65536, 128, 1152)]
tiles = {1: "128x128", 2: "128x64", 4: "64x64", 6: "32x128", 5: "128x32", 7: "32x32"}
st = torch.cuda.Stream()
def bench(M, N, K, tile, ks, reps):
    wbytes = N * K * 4
    pool = max(4, min(reps, int(600e6 // wbytes) + 1))
    Ws = [torch.randn(N, K, device=DEV) * 0.05 for _ in range(pool)]
    A = torch.randn(M, K, device=DEV)
    outs = [torch.empty(M, N, device=DEV) for _ in range(2)]
    tp = Tape(DEV)
    for r in range(reps):
        tp.linear(A, Ws[r % pool], None, outs[r % 2], M=M, K=K, N=N, tile=tile, ksplit=ks)
    tp.finalize()
    with torch.cuda.stream(st):
        tp.run(); st.synchronize()
        g = Tape.graph_capture(tp.run)
        Tape.graph_replay(g); st.synchronize()
        t0 = time.perf_counter()
        Tape.graph_replay(g); st.synchronize()
        dt = (time.perf_counter() - t0) / reps
    return dt * 1e6
for (M, N, K) in shapes:
    ideal = 2 * M * N * K / 157.3e12 * 1e6
    res = []
    for tile in (4, 2, 1, 7):
        bm, bn = map(int, tiles[tile].split("x"))
        if bn > N or (bm > M and tile != 6): continue
        nblk = math.ceil(M / bm) * math.ceil(N / bn)
        nch = K // 32 if tile != 7 else K // 16
        for ks in sorted({1, 2, 4, 8, 16}):
            if ks > nch // 2 or nblk * ks > 4096: continue
            if ks > 1 and nblk >= 512: continue
            if tile == 7 and nblk > 8192: continue
            us = bench(M, N, K, tile, ks, 30)
            res.append((us, tiles[tile], ks, nblk * ks))
    res.sort()
    print(f"M={M} N={N} K={K} ideal {ideal:.1f}us | " + " | ".join(f"{t} ks{k} ({b}blk): {u:.1f}" for u, t, k, b in res[:8]), flush=True)
