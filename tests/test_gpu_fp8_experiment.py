"""EXPERIMENT (BASELINE config 5's "fp8 MFMA path"): csrc/conv_gemm_f8.hip, tape.arith_mode("fp8").

Not a parity path -- an MX-FP8 GEMM deviates from fp32 by a few percent (tools/fp8_tolerance_study.py).  What
is tested: (1) the kernel computes EXACTLY the MX-FP8 contraction it claims to (against oracle/mxfp8.py, a CPU emulation of the same
quantisation); (2) the Stable Audio DiT built in that arithmetic stays within the experiment's acceptance bound of the fp32-exact
engine, and says so with numbers; (3) a clip edited in fp8 passes the experiment's acceptance criteria against the fp32-exact edit.

Agreement level of (1): 1.5e-5 ... 4e-5 relative, three orders under the arithmetic's own deviation from fp32 (3e-2).  It is not
1e-6 because ONE v_mfma_scale instruction does not add its 64 products exactly: with a spread of magnitudes inside the 64 products
the small ones lose their low bits against the largest (measured with tools/f8_acc_probe.cpp, profiles/r04_f8_probes.md: exact for
similar magnitudes, 5e-5 rel L2 / 1e-4 of the sum of magnitudes at worst for a wide spread), while the emulation sums in fp64.
The conversions themselves are bit-identical to torch's (RNE incl. subnormals; values above 464 would become NaN -- the kernel
clamps to +-448 first: tools/f8_cvt_probe.cpp)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from audioeditingcode_amd import _lib as L, configs, tape as tape_mod, weights          # noqa: E402
from audioeditingcode_amd.tape import Tape                                               # noqa: E402
from oracle import mxfp8                                                                 # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _run(tp):
    tp.run()
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K,tile", [(300, 200, 512, 1), (300, 200, 512, 2), (300, 200, 512, 3), (300, 200, 512, 4),
                                        (2050, 1536, 1536, 1), (128, 136, 64, 4)])
def test_fp8_linear_equals_the_cpu_emulation_of_the_same_quantisation(M, N, K, tile):
    g = torch.Generator().manual_seed(M + N + tile)
    x = torch.randn(M, K, generator=g) * torch.exp(torch.randn(K, generator=g))          # mixed column scales: blocks differ
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g)
    tp = Tape(DEV)
    out = tp.alloc(M, N)
    with tape_mod.arith_mode("fp8"):
        tp.linear(tp.hold(x.to(DEV)), tp.hold(w.to(DEV)), tp.hold(b.to(DEV)), out, M=M, K=K, N=N, res=tp.hold(r.to(DEV)), tile=tile)
    assert tp.ops[0].flags & 64
    _run(tp)
    emu = mxfp8.mx_linear(x, w) + b.double() + r.double()
    exact = x.double() @ w.double().T + b.double() + r.double()
    e_emu, e_exact = rel(out.cpu(), emu), rel(out.cpu(), exact)
    print(f"\n[fp8 linear] M={M} N={N} K={K} tile={tile}: vs CPU emulation {e_emu:.2e}, vs exact fp64 {e_exact:.2e}")
    assert e_emu < 2e-4, e_emu                     # see the module docstring: the MFMA's internal sum, not the quantisation
    assert 1e-4 < e_exact < 0.1, e_exact           # it IS an fp8 contraction: percent-level, not parity


def test_fp8_conv3x3_with_silu_epilogue_equals_the_cpu_emulation():
    B, H, W, C, N = 3, 16, 12, 128, 192
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, H, W, C, generator=g)
    w = torch.randn(N, 3, 3, C, generator=g) / (9 * C) ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    tp = Tape(DEV)
    out = tp.alloc(B, H, W, N)
    with tape_mod.arith_mode("fp8"):
        tp.conv(tp.hold(x.to(DEV)), tp.hold(w.reshape(N, -1).contiguous().to(DEV)), tp.hold(b.to(DEV)), out, B=B, IH=H, IW=W, Cin=C,
                OH=H, OW=W, N=N, KH=3, KW=3, pad_h=1, pad_w=1, out_act=L.ACT_SILU, tile=1)
    assert tp.ops[0].flags & 64
    _run(tp)
    emu = F.silu(mxfp8.mx_conv2d_nhwc(x, w) + b.double())
    assert rel(out.cpu(), emu) < 2e-4, rel(out.cpu(), emu)


def test_fp8_layernorm_fold_and_swiglu_epilogue_equal_the_cpu_emulation():
    """The DiT's FF1: LayerNorm folded into the GEMM (row statistics from the RAW fp32 rows, gathered in the loader before the
    quantisation) + SwiGLU in the epilogue."""
    M, C, Fd = 520, 256, 512
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, C, generator=g) * 2 + 0.5
    ga, be = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    w = torch.randn(2 * Fd, C, generator=g) / C ** 0.5
    b = torch.randn(2 * Fd, generator=g) * 0.1
    from audioeditingcode_amd.unet import geglu_pack_index
    perm = geglu_pack_index(Fd)
    wf = (w.double() * ga.double()[None, :]).float()[perm]
    t = (w.double() @ be.double() + b.double()).float()[perm]
    rowsum = wf.double().sum(1).float()
    tp = Tape(DEV)
    out = tp.alloc(M, Fd)
    with tape_mod.arith_mode("fp8"):
        tp.linear(tp.hold(x.to(DEV)), tp.hold(wf.to(DEV)), tp.hold(t.to(DEV)), out, M=M, K=C, N=2 * Fd,
                  ln_rowsum=tp.hold(rowsum.to(DEV)), geglu=2, tile=1)
    assert tp.ops[0].flags & 64
    _run(tp)
    mean = x.double().mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(x.double().var(1, unbiased=False, keepdim=True) + 1e-5)
    acc = mxfp8.mx_linear(x, wf)                                          # packed row order
    y = rstd * (acc - mean * rowsum.double()[None, :]) + t.double()[None, :]
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(2 * Fd)
    y = y[:, inv]
    emu = y[:, :Fd] * F.silu(y[:, Fd:])
    assert rel(out.cpu(), emu) < 2e-4, rel(out.cpu(), emu)


def test_shapes_the_fp8_kernel_does_not_take_fall_back_to_split_bf16():
    """Cin = 32 (not a multiple of the 64-wide MX chunk): the flagged record runs the split-bf16 kernel -- fp32-exact."""
    M, N, K = 256, 128, 32
    g = torch.Generator().manual_seed(2)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    tp = Tape(DEV)
    out = tp.alloc(M, N)
    with tape_mod.arith_mode("fp8"):
        tp.linear(tp.hold(x.to(DEV)), tp.hold(w.to(DEV)), None, out, M=M, K=K, N=N, tile=4)
    assert tp.ops[0].flags & 64 and tp.ops[0].flags & 4
    _run(tp)
    assert rel(out.cpu(), x.double() @ w.double().T) < 2e-6


def test_stable_audio_dit_in_fp8_stays_within_the_experiment_acceptance_bound():
    """Full-width Stable Audio DiT (1536 wide, 24 heads, 1025 tokens, 130 context keys; 4 of the 24 layers) built under
    arith_mode("fp8") against the split-bf16 (fp32-exact) engine: deviation reported, bounded, finite; the zero-context row and
    the conditioned row still differ.  Acceptance of the EXPERIMENT, not parity: rel L2 < 0.15 per forward (measured on the CPU
    emulation of tools/fp8_tolerance_study.py: ~8e-2)."""
    from audioeditingcode_amd.scheduler import CosineDPMSolverMultistepScheduler
    from audioeditingcode_amd.stable_audio import DiTEngine
    cfg = dict(configs.FAMILIES["stable_audio"]["dit"])
    cfg["num_layers"] = 4
    S = 130
    sd = weights.random_state_dict(weights.dit_param_shapes(cfg), seed=3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, cfg["in_channels"], cfg["sample_size"], generator=g)
    x[1] = x[0]
    ctx = torch.randn(2, S, cfg["cross_attention_input_dim"], generator=g)
    ctx[0] = 0
    glob = torch.randn(1, cfg["global_states_input_dim"], generator=g).expand(2, -1).contiguous()
    s = CosineDPMSolverMultistepScheduler()
    s.set_timesteps(200)
    t = s.timesteps[90]
    outs = {}
    for arith in ("bf16x6", "fp8"):
        with tape_mod.arith_mode(arith):
            eng = DiTEngine(cfg, sd, DEV, 2, S)
        n8 = sum(1 for op in eng.tape.ops if op.code == 1 and op.flags & 64)
        assert (n8 > 0) == (arith == "fp8")
        eng.set_conditioning(ctx, glob)
        eng.set_timestep(t)
        eng.x_in.copy_(x.transpose(1, 2))
        outs[arith] = eng.forward().transpose(1, 2).cpu().clone()
        torch.cuda.synchronize()
    dev = rel(outs["fp8"], outs["bf16x6"])
    print(f"\n[fp8 DiT] 4 layers at full width: rel L2 of the fp8 forward from the fp32-exact forward = {dev:.3e}")
    assert torch.isfinite(outs["fp8"]).all()
    assert 1e-4 < dev < 0.15, dev
    assert float((outs["fp8"][0] - outs["fp8"][1]).abs().max()) > 1e-3


def _logmel_like(wav, n_fft=512, hop=128, bands=32):
    """A perceptual proxy without external libraries: log of band-averaged STFT magnitudes (mean over channels)."""
    w = torch.hann_window(n_fft)
    mag = torch.stft(wav.float(), n_fft, hop, window=w, return_complex=True).abs()          # [ch, n_fft/2+1, frames]
    edges = torch.logspace(0, torch.log10(torch.tensor(float(n_fft // 2))), bands + 1).long().clamp(1, n_fft // 2)
    bandsum = torch.stack([mag[:, lo:max(hi, lo + 1)].mean(1) for lo, hi in zip(edges[:-1], edges[1:])], 1)
    return torch.log(bandsum.clamp_min(1e-5)).mean(0)


def test_fp8_edit_passes_the_experiment_acceptance_criteria_against_the_fp32_exact_edit(monkeypatch):
    """Acceptance of the fp8 EXPERIMENT at clip level (Stable Audio Open at FULL WIDTH -- 1536-wide DiT, 1025 tokens, full Oobleck
    -- with 4 of the 24 DiT layers and a short schedule; random weights: the criteria are structural, the numbers are reported).
    A clip is inverted and edited with the DiT on MX-FP8 GEMMs and on the fp32-exact split-bf16 GEMMs.  Criteria: the fp8 edit is
    finite and is CLOSER to the fp32-exact edit than that edit is to the un-edited clip, by a factor of two at least, in the latent
    and in a log-band-spectrum distance of the decoded audio: d(fp8, exact) < 0.5 * d(exact, original).  (Measured on the MI355X:
    latent 5.5e-2 against 5.3e-1, spectrum distance 34 against 8.5e3.)  Reported, not asserted: the same comparison for a replay
    with the SOURCE prompt and guidance from x_T.  With random weights the last, noise-free solver step returns the model's own
    denoised estimate, so neither arithmetic reproduces the input latent (rel 0.53 for both): "reconstruction" is not a criterion
    this container can test -- it needs the trained checkpoint."""
    from audioeditingcode_amd.main_run import edit_clip
    from audioeditingcode_amd.models import load_model
    from audioeditingcode_amd.utils import load_audio
    monkeypatch.setitem(configs.FAMILIES["stable_audio"]["dit"], "num_layers", 4)
    T, tstart = 10, 7
    g = torch.Generator().manual_seed(3)
    res = {}
    for arith in ("bf16x6", "fp8"):
        m = load_model("stabilityai/stable-audio-open-1.0", DEV, T, allow_synthetic=True)
        m.arith = arith
        n = m.model.transformer.config.sample_size * m.model.vae.hop_length
        sr = m.get_sr()
        if "wav" not in res:
            tt = torch.arange(n - 16, dtype=torch.float64) / sr
            res["wav"] = torch.stack([0.3 * torch.sin(2 * torch.pi * (220.0 + 5 * c) * tt).float()
                                      + 0.05 * torch.randn(n - 16, generator=g) for c in range(2)]).numpy()
        x0, _, duration = load_audio((res["wav"], sr), None, stft=False, model_sr=sr)
        torch.manual_seed(9)
        audio, orig, w_edit = edit_clip(m, x0, ["a dog barking"], ["a cat meowing"], [""], [2.0], [6.0], T, tstart, duration=duration)
        audio, orig, w_edit = audio.cpu().clone(), orig.cpu().clone(), w_edit.cpu().clone()    # (the latent is a loop-plan buffer)
        torch.manual_seed(9)
        _, _, w_same = edit_clip(m, x0, ["a dog barking"], ["a dog barking"], [""], [2.0], [2.0], T, T, duration=duration)
        w_same = w_same.cpu().clone()
        torch.manual_seed(9)
        w0 = m.vae_encode(x0).cpu().clone()
        res[arith] = dict(audio=audio, orig=orig, w=w_edit, w_same=w_same, w0=w0)
        del m
        torch.cuda.empty_cache()
    a, b = res["bf16x6"], res["fp8"]
    assert torch.isfinite(b["audio"]).all() and torch.isfinite(b["w"]).all()
    replay = rel(b["w_same"], a["w_same"])
    effect = rel(a["w"], a["w_same"])                              # what changing the prompt / guidance / tstart does (exact path)
    d_lat = rel(b["w"], a["w"])
    d_edit = rel(a["w"].reshape(-1), a["w0"].reshape(-1))
    ma, mb, mo = _logmel_like(a["audio"]), _logmel_like(b["audio"]), _logmel_like(a["orig"])
    F_ = min(ma.shape[-1], mb.shape[-1], mo.shape[-1])
    d_mel, d_mel_edit = float((mb[..., :F_] - ma[..., :F_]).norm()), float((ma[..., :F_] - mo[..., :F_]).norm())
    print(f"\n[fp8 acceptance] edited latent fp8 vs exact {d_lat:.3e} (the edit moved the latent by {d_edit:.3e}); log-band-spectrum "
          f"distance fp8 vs exact {d_mel:.3f} (edit vs original {d_mel_edit:.3f}); source-prompt replay fp8 vs exact {replay:.3e}; "
          f"target-prompt edit vs source-prompt replay (exact path) {effect:.3e}")
    assert d_lat > 1e-4                                            # the fp8 kernel really ran (full width: its GEMMs are LDS-staged)
    assert torch.isfinite(b["w_same"]).all()
    assert d_lat < 0.5 * d_edit, (d_lat, d_edit)
    assert d_mel < 0.5 * d_mel_edit, (d_mel, d_mel_edit)


def test_weight_quantiser_is_bit_identical_to_the_cpu_emulation_and_the_prequantised_path_is_used():
    """aed_mx_quantize_rows (what engines built under arith_mode("fp8") call once per weight) against oracle/mxfp8.py -- the bytes
    reinterpreted as e4m3 times the e8m0 scales equal the emulation's dequantised values EXACTLY; records built with it carry
    flag bit 7 and give the same result as in-loader quantisation of the same weights (identical operands, identical MFMAs)."""
    N, K, M = 200, 512, 300
    g = torch.Generator().manual_seed(4)
    w = torch.randn(N, K, generator=g) * torch.exp(torch.randn(N, 1, generator=g))
    w[3, 64:96] = 0                                               # an all-zero block
    x = torch.randn(M, K, generator=g)
    wd = w.to(DEV)
    q, sc = tape_mod.mx_quantized_weights(wd, N, K)
    torch.cuda.synchronize()
    deq = q.view(torch.float8_e4m3fn).float().cpu().reshape(N, K // 32, 32) * \
        torch.ldexp(torch.ones(()), sc.cpu().to(torch.int32) - 127).reshape(N, K // 32, 1)
    assert torch.equal(deq.reshape(N, K), mxfp8.mx_dequantized(w))
    outs = []
    for pre in (1, 0):
        tape_mod.FP8_PREQUANT = pre
        try:
            tp = Tape(DEV)
            out = tp.alloc(M, N)
            with tape_mod.arith_mode("fp8"):
                tp.linear(tp.hold(x.to(DEV)), wd, None, out, M=M, K=K, N=N, tile=1)
            assert bool(tp.ops[0].flags & 128) == bool(pre)
            _run(tp)
            outs.append(out.cpu().clone())
        finally:
            tape_mod.FP8_PREQUANT = 1
    assert torch.equal(outs[0], outs[1])
    assert rel(outs[0], mxfp8.mx_linear(x, w)) < 2e-4
