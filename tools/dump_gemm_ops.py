"""Write the AED_OP_CONV_GEMM records of one AudioLDM2 U-Net forward at batch B (default 200: the inversion's batched forward)
as text, for tools/x6_bench.cpp's `replay` mode: the GEMM-only forward in both arithmetics on synthetic operands, no Python
on the GPU box.  The engine is laid out on the CPU (torch.empty does not touch its 24 GB of activation pages); only the
records' integers travel.

    PYTHONPATH=. python tools/dump_gemm_ops.py [B] [regime=cus128] [share=2] > profiles/unet_b200_gemm_ops.txt

One line per record: name | flags under arith_mode("bf16x6") | fp32 tile | which of bias res rowvec A2 exist | i[0..39] | f[0..4]"""
import sys

from audioeditingcode_amd import _lib as L, configs, tape as tape_mod, weights
from audioeditingcode_amd.unet import UNetEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
REGIME = next((a.split("=")[1] for a in sys.argv[2:] if a.startswith("regime=")), None)     # tile tables of a pipeline partition
SHARE = next((int(a.split("=")[1]) for a in sys.argv[2:] if a.startswith("share=")), 1)      # 2: CFG-shared head (the product's loops)
fam = configs.FAMILIES["audioldm2"]
sd = weights.random_state_dict(weights.unet_param_shapes(fam["unet"]), seed=0)
engs = {}
for arith in ("f32", "bf16x6"):
    with tape_mod.arith_mode(arith), tape_mod.tile_regime(REGIME):
        engs[arith] = UNetEngine(fam["unet"], sd, "cpu", B, 256, 16, ctx_len0=8, ctx_len1=16, share=SHARE)
n = 0
for a, b, mt in zip(engs["f32"].tape.ops, engs["bf16x6"].tape.ops, engs["f32"].tape.meta):
    if a.code != L.OP_CONV_GEMM:
        continue
    have = [int(bool(b.p[k])) for k in (2, 4, 5, 8)]
    print("|".join([mt["name"].replace("|", "/").replace(" ", "_"), str(b.flags), str(a.i[29]), " ".join(map(str, have)),
                    " ".join(str(v) for v in b.i), " ".join(repr(float(v)) for v in list(b.f)[:5])]))
    n += 1
print(f"{n} conv_gemm records at batch {B}", file=sys.stderr)
