"""EXPERIMENTAL split-bf16 contraction (csrc/conv_gemm_x6.hip, tape.arith_mode("bf16x6")) through the Python host: the
same records with flag bits 2|3 against the fp32-MFMA kernel and an fp64 CPU convolution, and a full-size AudioLDM2 U-Net
built in that arithmetic against the fp32 engine and the oracle.

The kernel itself was validated on the MI355X through the C ABI (tools/x6_bench.cpp: 20 loader / epilogue modes x 2 variants,
profiles/r03_x6_gemm.md); the Python host path first ran on hardware in the round-3 driver's GPUTEST (passed)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from audioeditingcode_amd import _lib as L, configs, tape as tape_mod, weights          # noqa: E402
from audioeditingcode_amd.tape import Tape                                               # noqa: E402
from audioeditingcode_amd.unet import UNetEngine                                         # noqa: E402
from oracle import unet as ounet                                                         # noqa: E402

DEV = "cuda:0"


def _conv_pair(B, H, W, Cin, N, k, stride, res, act, seed):
    """One conv record built twice (fp32 / bf16x6) over the same device operands."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, Cin, generator=g) * torch.exp(torch.randn(Cin, generator=g))      # mixed channel scales
    w = torch.randn(N, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    OH, OW = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    r = torch.randn(B, OH, OW, N, generator=g) if res else None
    outs = {}
    for arith in ("f32", "bf16x6"):
        tp = Tape(DEV)
        xd, wd, bd = tp.hold(x.to(DEV)), tp.hold(w.reshape(N, -1).contiguous().to(DEV)), tp.hold(b.to(DEV))
        rd = tp.hold(r.to(DEV)) if res else None
        out = tp.alloc(B, OH, OW, N)
        with tape_mod.arith_mode(arith):
            tp.conv(xd, wd, bd, out, B=B, IH=H, IW=W, Cin=Cin, OH=OH, OW=OW, N=N, KH=k, KW=k, stride=stride, pad_h=k // 2,
                    pad_w=k // 2, res=rd, out_act=act, tile=1)
        assert bool(tp.ops[0].flags & 4) == (arith == "bf16x6")
        tp.run()
        torch.cuda.synchronize()
        outs[arith] = out.cpu()
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), stride=stride, padding=k // 2)
    ref = ref.permute(0, 2, 3, 1)
    if res:
        ref = ref + r.double()
    if act == L.ACT_SILU:
        ref = F.silu(ref)
    return outs, ref


@pytest.mark.parametrize("B,H,W,Cin,N,k,stride,res,act", [
    (8, 32, 16, 64, 128, 3, 1, False, 0),
    (8, 32, 16, 128, 128, 3, 2, True, 0),
    (4, 64, 16, 256, 256, 1, 1, True, 1),
    (5, 20, 10, 32, 136, 3, 1, False, 0),           # ragged M and N
])
def test_split_bf16_conv_is_as_close_to_fp64_as_the_fp32_kernel(B, H, W, Cin, N, k, stride, res, act):
    outs, ref = _conv_pair(B, H, W, Cin, N, k, stride, res, act, seed=B + Cin)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())         # noqa: E731
    e32, e6 = rel(outs["f32"], ref), rel(outs["bf16x6"], ref)
    assert rel(outs["bf16x6"], outs["f32"]) < 3e-6
    assert e6 < 1.5 * e32 + 1e-7, (e6, e32)          # six bf16 piece products lose nothing against the fp32 chain


def test_full_audioldm2_unet_in_split_bf16_matches_the_fp32_engine_and_the_oracle():
    """AudioLDM2 U-Net (346.9 M) at batch 8: the engine built under arith_mode("bf16x6") flags its LDS-staged GEMMs only, and
    its eps agrees with the fp32 engine to ~1e-5 (both carry fp32 rounding through ~400 GEMMs) and with the CPU oracle (first two rows) like the fp32 engine does."""
    fam = configs.FAMILIES["audioldm2"]
    cfg = fam["unet"]
    sd = weights.random_state_dict(weights.unet_param_shapes(cfg), seed=0)
    g = torch.Generator().manual_seed(3)
    Bn, H, W, L0, L1 = 8, 256, 16, 8, 16
    x = torch.randn(Bn, cfg["in_channels"], H, W, generator=g)
    e0 = torch.randn(Bn, L0, fam["ctx"]["gpt2_dim"], generator=g)
    e1 = torch.randn(Bn, L1, fam["ctx"]["t5_dim"], generator=g)
    m1 = torch.ones(Bn, L1)
    m1[0, L1 // 2:] = 0
    eps = {}
    for arith in ("f32", "bf16x6"):
        with tape_mod.arith_mode(arith):
            eng = UNetEngine(cfg, sd, DEV, Bn, H, W, ctx_len0=L0, ctx_len1=L1)
        n_split = sum(1 for o in eng.tape.ops if o.code == L.OP_CONV_GEMM and (o.flags & 4))
        assert (n_split > 20) if arith == "bf16x6" else (n_split == 0), (arith, n_split)
        eng.set_conditioning(ehs0=e0, ehs1=e1, bias1=(1 - m1) * -10000.0)
        eng.x_in.copy_(x.permute(0, 2, 3, 1))
        eng.set_timestep(601)
        eng.forward()
        torch.cuda.synchronize()
        eps[arith] = eng.eps.cpu().permute(0, 3, 1, 2).clone()
        del eng
        torch.cuda.empty_cache()
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())         # noqa: E731
    assert rel(eps["bf16x6"], eps["f32"]) < 5e-5, rel(eps["bf16x6"], eps["f32"])      # two fp32-accurate evaluations of a 400-GEMM graph
    ref, _, _ = ounet.unet_forward(cfg, sd, x[:2], torch.tensor(601), encoder_hidden_states=e0[:2],
                                   encoder_hidden_states_1=e1[:2], encoder_attention_mask_1=m1[:2])
    assert rel(eps["bf16x6"][:2], ref) < 1e-4 and rel(eps["f32"][:2], ref) < 1e-4


def test_c_abi_harness_feature_matrix():
    """tools/x6_bench.cpp `cases`: a plain C++ host (no Python, no torch) launches 20 records covering every loader / epilogue
    mode of AED_OP_CONV_GEMM (stride, upsample, two-source A, SiLU / LeakyReLU of A, row vector, residual, activations, split-K,
    accumulate modes, row scatter, ragged edges, Cin = 16, the fp32 fallback) through aed_launch with and without the
    interleave hints and compares the split-bf16 result with the fp32 kernel's on the device (rel L2 < 5e-6 each)."""
    import json
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audioeditingcode_amd", "x6_bench")
    if not os.path.exists(exe):
        pytest.skip("audioeditingcode_amd/x6_bench is built by __graft_entry__.build()")
    r = subprocess.run([exe, "1", "cases"], capture_output=True, text=True, timeout=300)
    rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{"case"')]
    assert len(rows) == 40, (r.returncode, r.stderr[-400:])
    bad = [row for row in rows if not row["pass"]]
    assert r.returncode == 0 and not bad, (bad[:3], r.stderr[-400:])
    assert max(row["rel_l2_vs_fp32_kernel"] for row in rows) < 5e-6
