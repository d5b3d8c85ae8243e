"""Measured tile choices for AED_OP_CONV_GEMM: (M, N, K, geglu) -> (tile code, ksplit).

Filled from tools/tile_sweep.py runs on the MI355X (cold weights, warm activations, dependent launches in a hipGraph);
shapes not listed fall back to the rule in Tape.pick_tile.  Tile codes: see Tape.LIN_TILES / conv_gemm.hip."""
TILE_TABLE = {}
