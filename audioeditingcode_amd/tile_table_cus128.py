"""Tile choices for AED_OP_CONV_GEMM on a 128-CU partition: (M, N, K, geglu) -> (tile code, ksplit).

The edit loop of the clip pipeline (pipeline.ClipPipeline, plan "partition") runs its batch-2 U-Net forwards on half of the
chip's compute units.  There the step is no longer purely latency-bound: kernels with several hundred workgroups run
more than one round per CU, so larger tiles (less redundant operand traffic per flop) win where the whole-chip table
prefers the smallest ones.  Measured with `AED_SWEEP_CUS=128 python tools/tile_sweep.py 2 60` on the MI355X (every distinct
contraction of the batch-2 forward, cold weights, dependent launches in a hipGraph, on a stream masked to 128 CUs); listed
are the shapes whose best tile beats the whole-chip choice by more than 3 % there (conv_gemm time per forward
10.13 -> 9.42 ms).  Used only by engines built under `tape.tile_regime("cus128")`."""
TILE_TABLE = {
    (128, 640, 64, 0): (12, 1),   # 4.7 us on 128 CUs; whole-chip choice 10:1 -> 5.1 us
    (128, 640, 128, 0): (12, 1),   # 5.1 us on 128 CUs; whole-chip choice 10:1 -> 5.3 us
    (128, 640, 640, 0): (10, 1),   # 8.2 us on 128 CUs; whole-chip choice 11:1 -> 8.4 us
    (128, 640, 3456, 0): (12, 1),   # 26.6 us on 128 CUs; whole-chip choice 4:26 -> 32.1 us
    (128, 1920, 640, 0): (10, 1),   # 14.7 us on 128 CUs; whole-chip choice 11:1 -> 17.6 us
    (128, 5120, 640, 1): (13, 1),   # 32.5 us on 128 CUs; whole-chip choice 15:1 -> 39.7 us
    (512, 64, 384, 0): (11, 1),   # 8.1 us on 128 CUs; whole-chip choice 10:1 -> 9.0 us
    (512, 128, 384, 0): (11, 1),   # 8.6 us on 128 CUs; whole-chip choice 10:1 -> 9.8 us
    (512, 384, 640, 0): (10, 1),   # 12.1 us on 128 CUs; whole-chip choice 11:1 -> 14.3 us
    (512, 384, 768, 0): (10, 1),   # 13.4 us on 128 CUs; whole-chip choice 11:1 -> 15.7 us
    (512, 384, 1024, 0): (10, 1),   # 16.1 us on 128 CUs; whole-chip choice 11:1 -> 18.5 us
    (512, 384, 1920, 0): (10, 1),   # 25.3 us on 128 CUs; whole-chip choice 11:1 -> 27.4 us
    (512, 384, 3456, 0): (4, 11),   # 47.4 us on 128 CUs; whole-chip choice 12:1 -> 49.3 us
    (512, 384, 6912, 0): (4, 11),   # 76.0 us on 128 CUs; whole-chip choice 12:1 -> 90.9 us
    (2048, 64, 256, 0): (11, 1),   # 7.5 us on 128 CUs; whole-chip choice 10:1 -> 8.2 us
    (2048, 128, 1152, 0): (10, 1),   # 20.9 us on 128 CUs; whole-chip choice 11:1 -> 23.4 us
    (2048, 256, 64, 0): (16, 1),   # 8.9 us on 128 CUs; whole-chip choice 10:1 -> 9.9 us
    (2048, 256, 256, 0): (4, 1),   # 11.4 us on 128 CUs; whole-chip choice 10:1 -> 13.7 us
    (2048, 256, 384, 0): (4, 1),   # 14.4 us on 128 CUs; whole-chip choice 10:1 -> 17.4 us
    (2048, 256, 512, 0): (15, 1),   # 17.0 us on 128 CUs; whole-chip choice 10:1 -> 20.1 us
    (2048, 256, 640, 0): (15, 1),   # 19.5 us on 128 CUs; whole-chip choice 10:1 -> 23.0 us
    (2048, 256, 1152, 0): (4, 1),   # 32.8 us on 128 CUs; whole-chip choice 10:1 -> 41.5 us
    (2048, 256, 1280, 0): (15, 1),   # 30.6 us on 128 CUs; whole-chip choice 10:1 -> 34.2 us
    (2048, 256, 2304, 0): (17, 1),   # 56.2 us on 128 CUs; whole-chip choice 13:1 -> 69.6 us
    (2048, 384, 3456, 0): (4, 3),   # 131.1 us on 128 CUs; whole-chip choice 12:1 -> 144.7 us
    (2048, 768, 256, 0): (3, 1),   # 29.0 us on 128 CUs; whole-chip choice 4:1 -> 30.2 us
    (2048, 2048, 256, 1): (1, 1),   # 52.8 us on 128 CUs; whole-chip choice 3:1 -> 56.6 us
    (8192, 8, 1152, 0): (10, 1),   # 20.6 us on 128 CUs; whole-chip choice 11:1 -> 22.8 us
    (8192, 128, 2304, 0): (4, 1),   # 98.9 us on 128 CUs; whole-chip choice 17:1 -> 113.4 us
    (8192, 128, 3456, 0): (4, 1),   # 144.1 us on 128 CUs; whole-chip choice 17:1 -> 158.6 us
    (8192, 256, 2304, 0): (1, 1),   # 178.7 us on 128 CUs; whole-chip choice 4:1 -> 191.3 us
}
