"""The wrapper API (models.PipelineWrapper subclasses, main_run.edit_clip, the hook arguments of unet_forward,
pc_drift on the real `unet_forward_pair`) executed on CPU: a test-only subclass lifts the CUDA requirement, tapes are
run by oracle/tape_interp.py and the three direct C-ABI calls by its FakeLib.  This is the smoke() flow of
__graft_entry__ and the hook tests of the GPU suite, without a GPU -- host logic only; the kernels are proven on the
GPU."""
import pytest
import torch

from audioeditingcode_amd import _lib as L
from audioeditingcode_amd import models, pc_drift
from audioeditingcode_amd.main_run import edit_clip
from audioeditingcode_amd.tape import Tape
from audioeditingcode_amd.utils import get_text_embeddings, load_audio, synthetic_clip
from oracle import loops as oloops
from oracle import pc as opc
from oracle import unet as ounet
from oracle import vae as ovae
from oracle.scheduler import OracleDDIMScheduler


class _CpuAudioLDM2(models.AudioLDM2Wrapper):
    def _require_device(self):           # test instrumentation only: the product class refuses a CPU device
        pass


def rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def _model(T):
    m = _CpuAudioLDM2(model_id="tiny/audioldm2", device="cpu", seed=0)
    m.load_scheduler()
    m.model.scheduler.set_timesteps(T, device=None)
    return m


def _oracle_wrapper(m, T):
    cfg, sd = m.family["unet"], m.state_dicts["unet"]
    osched = OracleDDIMScheduler()
    osched.set_timesteps(T)

    def unet_fn(x, t, cond):
        hs, cl, mk = (v.expand(x.shape[0], *v.shape[1:]) for v in cond)
        return ounet.unet_forward(cfg, sd, x, t, encoder_hidden_states=hs, encoder_hidden_states_1=cl,
                                  encoder_attention_mask_1=mk)[0]
    return oloops.OracleWrapper(osched, unet_fn)


def test_product_wrapper_refuses_cpu():
    with pytest.raises(L.AedError):
        models.AudioLDM2Wrapper(model_id="tiny/audioldm2", device="cpu")


def test_edit_clip_end_to_end_on_cpu(cpu_stack):
    """wav -> STFT/mel -> VAE encode -> inversion -> edit -> VAE decode -> vocoder through the wrapper API."""
    T, tstart = 4, 3
    m = _model(T)
    x0, _, _ = load_audio((synthetic_clip(seconds=0.64, seed=7), 16000), m.get_fn_STFT(), device="cpu", stft=True)
    torch.manual_seed(5)
    audio, orig, w_edit = edit_clip(m, x0, ["a dog barking"], ["a cat meowing"], [""], [3.0], [12.0], T, tstart)
    ow = _oracle_wrapper(m, T)
    w0 = ovae.vae_encode(m.family["vae"], m.state_dicts["vae"], x0)
    xts0 = ow.sample_xts_from_x0(w0, T, generator=torch.Generator().manual_seed(5))
    _, zs, xts = oloops.invert(ow, w0, m.encode_text(["a dog barking"]), m.encode_text([""]), [3.0], T, xts=xts0)
    w_o = oloops.edit(ow, xts, torch.tensor([tstart]), m.encode_text(["a cat meowing"]), m.encode_text([""]), [12.0],
                      zs[:tstart], eta=1.0)
    assert rel(w_edit, w_o) < 2e-3, rel(w_edit, w_o)
    assert torch.isfinite(audio).all() and torch.isfinite(orig).all() and audio.shape[0] == orig.shape[0] == 1


def test_unet_forward_hooks_on_cpu(cpu_stack):
    """h-space replacement / mid-block residual / zeroed skip connections (models.py:691-899) vs the oracle."""
    m = _model(4)
    cfg, sd = m.family["unet"], m.state_dicts["unet"]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 8, 16, 16, generator=g)
    hs, cl, mk = m.encode_text(["rain"])
    t = torch.tensor(501)
    out, h, skips = m.unet_forward(x, t, encoder_hidden_states=hs, class_labels=cl, encoder_attention_mask=mk)
    ref, ref_h, ref_s = ounet.unet_forward(cfg, sd, x, t, encoder_hidden_states=hs, encoder_hidden_states_1=cl,
                                           encoder_attention_mask_1=mk)
    assert rel(out.sample, ref) < 1e-4 and rel(h, ref_h) < 1e-4
    new_h = torch.randn(h.shape, generator=g)
    add = torch.randn(h.shape, generator=g) * 0.1
    out2, _, _ = m.unet_forward(x, t, encoder_hidden_states=hs, class_labels=cl, encoder_attention_mask=mk,
                                replace_h_space=new_h, mid_block_additional_residual=add, zero_out_resconns=[1])
    ref2, _, _ = ounet.unet_forward(cfg, sd, x, t, encoder_hidden_states=hs, encoder_hidden_states_1=cl,
                                    encoder_attention_mask_1=mk, replace_h_space=new_h,
                                    mid_block_additional_residual=add, zero_out_resconns=[1])
    assert rel(out2.sample, ref2) < 1e-4
    assert rel(out2.sample, ref) > 1e-2                                    # the hooks did change the output


def test_product_unet_forward_matches_the_references_inline_graphs_on_cpu(cpu_stack):
    """The product's `unet_forward` (three wrappers, hook arguments included) on the CPU tape interpreter against the fixture
    written by the reference's own inline forward graphs (tests/unet_graph_cases.py; models.py:160-393, :691-899)."""
    import unet_graph_cases as ugc
    g = ugc.load()
    classes = {"audioldm": _CpuAudioLDM, "tango": _CpuTango, "audioldm2": _CpuAudioLDM2}
    for fam in ugc.FAMILIES:
        m = classes[fam](model_id=f"tiny/{fam}", device="cpu", seed=int(g[f"{fam}.seed"]))
        for fam_, size, hook in ugc.cases():
            if fam_ != fam:
                continue
            x, t, cond = ugc.inputs(g, fam, size)
            out, h, skips = m.unet_forward(x, torch.tensor(t), **cond, **ugc.hook_kwargs(g, fam, hook))
            skips = {i: [s.clone() for s in skips[i]] for i in range(4)}
            ugc.check(fam, size, hook, (out.sample, h, skips), ugc.expected(g, fam, size, hook), 1e-4, rel)


def test_pc_functions_on_the_real_wrapper_on_cpu(cpu_stack):
    """forward_directional / get_eigenvectors through the wrapper's own `unet_forward_pair` (one 2*n_ev batch)."""
    T = 6
    m = _model(T)
    ow = _oracle_wrapper(m, T)
    g = torch.Generator().manual_seed(3)
    xt = torch.randn(1, 8, 16, 16, generator=g) * 0.8
    latent = torch.randn(1, 8, 16, 16, generator=g)
    t = m.model.scheduler.timesteps[2]
    _, e_txt, e_unc = get_text_embeddings(["jazz"], [""], m)
    xtm1, x0p = pc_drift.forward_directional(m, xt, t, latent, e_unc, e_txt, 3.0, eta=1.0)
    xtm1_o, x0p_o = opc.forward_directional(ow, xt, t, latent, m.encode_text([""]), m.encode_text(["jazz"]), 3.0, eta=1.0)
    assert rel(xtm1, xtm1_o) < 1e-4 and rel(x0p, x0p_o) < 1e-4
    mask = torch.ones_like(xt)
    init = torch.randn(2, 8, 16, 16, generator=g)
    ev, val, _, _, _, _ = pc_drift.get_eigenvectors(m, xt, e_txt, e_unc, latent, mask, t, x0p, const=1e-2, cfg_tar=3.0,
                                                    iters=3, n_ev=2, init_eigvecs=init)
    ev_o, val_o, _, _ = opc.get_eigenvectors(ow, xt, m.encode_text(["jazz"]), m.encode_text([""]), latent, mask, t,
                                             x0p_o, init, const=1e-2, cfg_tar=3.0, iters=3, n_ev=2)
    cos = (ev.reshape(2, -1) * ev_o.reshape(2, -1)).sum(1)
    assert (cos > 0.99).all(), cos
    assert rel(val, val_o) < 2e-2


class _CpuAudioLDM(models.AudioLDMWrapper):
    def _require_device(self):
        pass


class _CpuTango(models.TangoWrapper):
    def _require_device(self):
        pass


def _model_of(cls, model_id, T):
    m = cls(model_id=model_id, device="cpu", seed=0)
    m.load_scheduler()
    m.model.scheduler.set_timesteps(T, device=None)
    return m


def _oracle_wrapper_for(m, T):
    cfg, sd = m.family["unet"], m.state_dicts["unet"]
    osched = OracleDDIMScheduler()
    osched.set_timesteps(T)
    kind = m.kind

    def unet_fn(x, t, cond):
        hs, cl, mk = (None if v is None else v.expand(x.shape[0], *v.shape[1:]) for v in cond)
        if kind == "audioldm2":
            kw = dict(encoder_hidden_states=hs, encoder_hidden_states_1=cl, encoder_attention_mask_1=mk)
        elif kind == "audioldm":
            kw = dict(class_labels=cl)
        else:
            kw = dict(encoder_hidden_states=hs, encoder_attention_mask=mk)
        return ounet.unet_forward(cfg, sd, x, t, **kw)[0]
    return oloops.OracleWrapper(osched, unet_fn)


@pytest.mark.parametrize("cls,model_id", [(_CpuAudioLDM, "tiny/audioldm"), (_CpuTango, "tiny/tango")])
def test_other_families_invert_and_edit_on_cpu(cpu_stack, cls, model_id):
    from audioeditingcode_amd.ddm_inversion import inversion_forward_process, inversion_reverse_process
    T, tstart = 4, 3
    m = _model_of(cls, model_id, T)
    ow = _oracle_wrapper_for(m, T)
    w0 = torch.randn(1, 8, 16, 16, generator=torch.Generator().manual_seed(2)) * 0.7
    torch.manual_seed(6)
    _, zs, wts, _ = inversion_forward_process(m, w0, etas=1.0, prompts=["rain"], cfg_scales=[3.0],
                                              num_inference_steps=T, numerical_fix=True)
    w, _ = inversion_reverse_process(m, xT=wts, tstart=torch.tensor([tstart]), etas=1.0, prompts=["jazz"],
                                     neg_prompts=[""], cfg_scales=[9.0], zs=zs[:tstart])
    xts0 = ow.sample_xts_from_x0(w0, T, generator=torch.Generator().manual_seed(6))
    _, zs_o, xts_o = oloops.invert(ow, w0, m.encode_text(["rain"]), m.encode_text([""]), [3.0], T, xts=xts0)
    w_o = oloops.edit(ow, xts_o, torch.tensor([tstart]), m.encode_text(["jazz"]), m.encode_text([""]), [9.0],
                      zs_o[:tstart], eta=1.0)
    assert rel(wts[1:], xts_o[1:]) < 1e-5 and rel(w, w_o) < 2e-3, (rel(wts[1:], xts_o[1:]), rel(w, w_o))


def test_per_step_eta_lists_through_the_reference_signatures_on_cpu(cpu_stack):
    """`etas` given as a list to inversion_forward_process / inversion_reverse_process (inversion_utils.py:59-60, :124,
    :210-214, :302): device loops with a coefficient row per step; a list that turns the noise off at SOME edit steps
    takes the literal step-by-step path (the reference skips the noise term there)."""
    from audioeditingcode_amd.ddm_inversion import inversion_forward_process, inversion_reverse_process
    T, tstart = 4, 3
    m = _model_of(_CpuTango, "tiny/tango", T)
    ow = _oracle_wrapper_for(m, T)
    w0 = torch.randn(1, 8, 16, 16, generator=torch.Generator().manual_seed(2)) * 0.7
    etas = [1.0, 0.7, 0.9, 0.6]
    torch.manual_seed(6)
    _, zs, wts, _ = inversion_forward_process(m, w0, etas=list(etas), prompts=["rain"], cfg_scales=[3.0],
                                              num_inference_steps=T, numerical_fix=True)
    xts0 = ow.sample_xts_from_x0(w0, T, generator=torch.Generator().manual_seed(6))
    _, zs_o, xts_o = oloops.invert(ow, w0, m.encode_text(["rain"]), m.encode_text([""]), [3.0], T, eta=etas, xts=xts0)
    assert rel(wts[1:], xts_o[1:]) < 1e-5 and rel(zs[1:], zs_o[1:]) < 2e-3
    for ed_etas in (etas, [1.0, 0.0, 0.9, 0.6]):                    # second list: no noise at idx 1 -> literal path
        w, _ = inversion_reverse_process(m, xT=wts, tstart=torch.tensor([tstart]), etas=list(ed_etas), prompts=["jazz"],
                                         neg_prompts=[""], cfg_scales=[9.0], zs=zs[:tstart])
        w_o = oloops.edit(ow, xts_o, torch.tensor([tstart]), m.encode_text(["jazz"]), m.encode_text([""]), [9.0],
                          zs_o[:tstart], eta=ed_etas)
        assert torch.isfinite(w).all() and rel(w, w_o) < 3e-3, rel(w, w_o)


def test_ddim_mode_and_sdedit_on_cpu(cpu_stack):
    from audioeditingcode_amd.sdedit import sdedit
    T = 5
    m = _model(T)
    ow = _oracle_wrapper(m, T)
    x0, _, _ = load_audio((synthetic_clip(seconds=0.64, seed=3), 16000), m.get_fn_STFT(), device="cpu", stft=True)
    _, _, w_ddim = edit_clip(m, x0, ["a dog barking"], ["a cat meowing"], [""], [3.0], [7.0], T, 4, mode="ddim")
    w0 = ovae.vae_encode(m.family["vae"], m.state_dicts["vae"], x0)
    wT = oloops.ddim_invert(ow, w0, m.encode_text(["a dog barking"]), m.encode_text([""]), 3.0, T, 1)
    w_o = oloops.ddim_sample(ow, wT, m.encode_text(["a cat meowing"]), m.encode_text([""]), 7.0, skip=1)
    assert rel(w_ddim, w_o) < 2e-3, rel(w_ddim, w_o)
    # SDEdit: default draws in the reference's order (torch global RNG), checked against the oracle loop pinned to
    # the reference's main_run_sdedit.py
    torch.manual_seed(11)
    got = sdedit(m, w0, ["jazz"], [""], 5.0, skip=2, eta=1.0)
    torch.manual_seed(11)
    ref = opc.sdedit_loop(ow, w0, m.encode_text(["jazz"]), m.encode_text([""]), 5.0, skip=2, eta=1.0)
    assert rel(got, ref) < 2e-3, rel(got, ref)


def test_two_prompt_segments_equal_and_unequal_tstart_on_cpu(cpu_stack):
    """Multi-prompt editing: equal tstart runs on the device-resident loop with per-element cfg tensors, unequal tstart
    on the host-driven hook path with the trajectory blend -- both against the oracle."""
    from audioeditingcode_amd.ddm_inversion import inversion_forward_process, inversion_reverse_process
    T = 6
    m = _model(T)
    ow = _oracle_wrapper(m, T)
    w0 = torch.randn(1, 8, 32, 16, generator=torch.Generator().manual_seed(4)) * 0.7
    torch.manual_seed(8)
    _, zs, wts, _ = inversion_forward_process(m, w0, etas=1.0, prompts=["rain"], cfg_scales=[3.0],
                                              num_inference_steps=T, numerical_fix=True)
    xts0 = ow.sample_xts_from_x0(w0, T, generator=torch.Generator().manual_seed(8))
    _, zs_o, xts_o = oloops.invert(ow, w0, m.encode_text(["rain"]), m.encode_text([""]), [3.0], T, xts=xts0)
    tgt = m.encode_text(["jazz", "rock"])
    for tstart in ([4, 4], [4, 3]):
        w, _ = inversion_reverse_process(m, xT=wts, tstart=torch.tensor(tstart), fix_alpha=0.2, etas=1.0,
                                         prompts=["jazz", "rock"], neg_prompts=[""], cfg_scales=[9.0, 6.0],
                                         zs=zs[:max(tstart)], cutoff_points=[0.5])
        w_o = oloops.edit(ow, xts_o, torch.tensor(tstart), tgt, m.encode_text([""]), [9.0, 6.0], zs_o[:max(tstart)],
                          eta=1.0, n_prompts=2, cutoff_points=[0.5], fix_alpha=0.2)
        assert rel(w, w_o) < 2e-3, (tstart, rel(w, w_o))


def test_pc_extract_and_apply_on_the_product_stack_on_cpu(cpu_stack):
    """main_pc_extract_inv.extract_pcs + main_pc_apply_drift.apply_pcs with their DEFAULT functions (product inversion,
    pc_drift on the native wrapper API): runs end to end, window / layout / finiteness; the arithmetic of each piece is
    pinned elsewhere (pc_cli.npz, test_pc_functions_on_the_real_wrapper_on_cpu)."""
    from argparse import Namespace
    from audioeditingcode_amd import main_pc_apply_drift as papply, main_pc_extract_inv as pext
    T = 5
    m = _model(T)
    w0 = torch.randn(1, 8, 16, 16, generator=torch.Generator().manual_seed(5)) * 0.7
    a = pext.finish_args(Namespace(seed=1, cfg_tar=3, model_id="tiny/audioldm2", init_aud=None, num_diffusion_steps=T,
                                   source_prompt=["rain"], target_neg_prompt=[""], corr_to_swap=0.8, drift_start=4,
                                   drift_end=2, results_path="unused", const=1e-2, n_evs=2, patch=None, iters=2, dry=False))
    torch.manual_seed(1)
    ck = pext.extract_pcs(m, w0, a)
    ts = [int(t) for t in m.model.scheduler.timesteps]
    assert sorted(ck["eigdata"].keys(), reverse=True) == ts[1:3]                # its 1, 2 of T = 5
    assert all(torch.isfinite(e["eigvec"]).all() and (e["eigval"] > 0).all() for e in ck["eigdata"].values())
    ap = Namespace(drift_start=4, drift_end=2, amount=1.5, use_specific_ts_pc=None, fix_alpha=None, fade_length=0.0,
                   evs=[1, 2], combine_evs=False, evals_pt=None, rand_v=False, shift_x0_for_np=True, sub_iters=None)
    out = papply.apply_pcs(m, {k: ck[k] for k in ("eigdata", "args", "corrs", "in_corrs", "latents", "in_norms", "xts")},
                           ap, torch.device("cpu"))
    assert out.shape == (2, 8, 16, 16) and torch.isfinite(out).all()
    assert rel(out[0:1], ck["final"]) > 1e-3                                      # the drift moved the sample


def test_product_tape_executor_fails_loudly_without_gpu():
    """Without the test-side interpreter patch, running a tape on a CPU box raises -- there is no CPU execution path in
    the product."""
    tp = Tape("cpu")
    x, y = torch.ones(4, 8), tp.alloc(4, 8)
    tp.axpby(x, y, numel=32, a=2.0, b=0.0)
    with pytest.raises(Exception):
        tp.run()


def test_prefetched_noise_equals_inline_draws(cpu_stack):
    """PipelineWrapper.prefetch_noise: the x_t noise drawn ahead on a host thread is bit-identical to the in-line draws
    (same generator algorithm / seed / order), the global generator ends in the same state, and a caller that seeded
    differently silently gets the in-line path."""
    T = 6
    m = _model(T)
    w0 = torch.randn(1, 8, 16, 16, generator=torch.Generator().manual_seed(2)) * 0.5
    torch.manual_seed(31)
    ref = m.sample_xts_from_x0(w0, T)
    after = torch.randn(3)
    m.prefetch_noise(31, w0.shape, T)
    torch.manual_seed(31)
    got = m.sample_xts_from_x0(w0, T)
    assert torch.equal(got, ref) and torch.equal(torch.randn(3), after)
    m.prefetch_noise(31, w0.shape, T)
    torch.manual_seed(32)                                   # other seed: the prefetched buffer must not be used
    other = m.sample_xts_from_x0(w0, T)
    assert not torch.equal(other, ref)
    torch.manual_seed(32)
    assert torch.equal(other, m.sample_xts_from_x0(w0, T))
    m.next_noise_seed = 31                                  # announced seed -> drawn under the current clip
    torch.manual_seed(5)
    m.sample_xts_from_x0(w0, T)
    assert m._noise_box["seed"] == 31
    torch.manual_seed(31)
    assert torch.equal(m.sample_xts_from_x0(w0, T), ref)
    # a latent whose per-step element count is not a multiple of 16 (one stacked randn would differ from T draws there)
    w_odd = torch.randn(1, 8, 3, 5, generator=torch.Generator().manual_seed(3)) * 0.5
    torch.manual_seed(77)
    noise_inline = torch.stack([torch.randn(w_odd.shape) for _ in range(T)])
    m.prefetch_noise(77, w_odd.shape, T)
    torch.manual_seed(77)
    assert torch.equal(m._take_prefetched_noise(w_odd.shape, T), noise_inline)


def _fake_hip_for_pipeline(monkeypatch, log):
    import contextlib

    from audioeditingcode_amd.pipeline import ClipPipeline

    class FakeStream:
        def __init__(self, name):
            self.name = name

        def wait_event(self, ev):
            log.append(("wait", self.name, ev.tag))

    class FakeLane:
        def __init__(self, device, cus=None, total=None, priority=0):
            self.total = 256 if total is None else total
            self.cus = None if cus is None else list(cus)
            self.stream = FakeStream("chip" if cus is None else f"cus{self.cus[0]}-{self.cus[-1]}")

        def close(self):
            pass

    class FakeEvent:
        n = 0

        def __init__(self, enable_timing=False):
            FakeEvent.n += 1
            self.tag = FakeEvent.n

        def record(self, st):
            log.append(("record", st.name, self.tag))

        def query(self):
            return True

        def elapsed_time(self, other):
            return 0.0
    monkeypatch.setattr(ClipPipeline, "lane_type", FakeLane)
    monkeypatch.setattr(ClipPipeline, "event_type", FakeEvent)
    monkeypatch.setattr(ClipPipeline, "_stream_ctx", staticmethod(lambda st: contextlib.nullcontext()))
    return ClipPipeline


def test_clip_pipeline_equals_serial_edit_clip(cpu_stack, monkeypatch):
    """pipeline.ClipPipeline on the CPU stack with recording stand-ins for the HIP streams / events.
    Partition plan (front stage: inversion on CUs [96,256) | back stage: edit loop on CUs [0,96), 2 edit lanes): every
    clip equals main_run.edit_clip with the batched inversion, bit for bit, with per-clip seeds; fill runs on the whole
    chip, steady state on the partitions; every back half waits for its clip's front-half event; without per-clip seeds the
    clips consume ONE continuous global generator stream in clip order.  Lane views share the frozen weights and own their
    engines; a failing clip surfaces; the next clip's set-up stream is masked to the inversion partition (round 6)."""
    log = []
    ClipPipeline = _fake_hip_for_pipeline(monkeypatch, log)
    threads = torch.get_num_threads()
    torch.set_num_threads(2)                    # lane threads x intra-op threads would oversubscribe the test box
    try:
        T, tstart = 3, 2
        m = _model(T)
        mels = [load_audio((synthetic_clip(seconds=0.32, seed=7 + i), 16000), m.get_fn_STFT(), device="cpu", stft=True)[0]
                for i in range(3)]
        args = (["a dog barking"], ["a cat meowing"], [""], [3.0], [12.0], T, tstart)
        serial_b = []
        for i, x0 in enumerate(mels):
            torch.manual_seed(40 + i)
            serial_b.append(edit_clip(m, x0, *args, schedule="batched", timestep_group=3))
        # ---- partition plan
        seen = []
        ed_cls = type(m.editor(mels[0].shape[-2] // 4, mels[0].shape[-1] // 4))
        orig_invert, orig_edit = ed_cls.invert, ed_cls.edit
        monkeypatch.setattr(ed_cls, "invert", lambda self, *a, **k: (seen.append(("invert", self.loop_stream().name)),
                                                                     orig_invert(self, *a, **k))[1])
        monkeypatch.setattr(ed_cls, "edit", lambda self, *a, **k: (seen.append(("edit", self.loop_stream().name)),
                                                                   orig_edit(self, *a, **k))[1])
        pipe = ClipPipeline(m, plan="partition", edit_cus=96, edit_lanes=2, timestep_group=3)
        # two edit lanes -> a third stage decodes (VAE decode + vocoder) on a queue over the inversion partition's CUs
        assert [w.stage for w in pipe.workers] == ["front", "back", "back", "codec"] and pipe.clips_in_flight == 4
        assert pipe.workers[3].lane is pipe.workers[0].lane and pipe.codec_stage           # decodes on the inversion partition's queue
        assert not ClipPipeline(m, plan="partition", edit_cus=96, timestep_group=3).codec_stage           # one lane: as in round 3
        assert pipe.workers[0].lane.cus == list(range(96, 256)) and pipe.workers[1].lane.cus == list(range(96))
        assert pipe.workers[2].lane.cus == list(range(96))              # 48 CUs per lane is not a legal mask: the lanes share
        split = ClipPipeline(m, plan="partition", edit_cus=128, edit_lanes=2, timestep_group=3)
        assert [w.lane.cus for w in split.workers] == [list(range(128, 256)), list(range(64)), list(range(64, 128)),
                                                       list(range(128, 256))]
        # drain widening (round 4): each disjoint edit lane owns a second queue over its CUs + its half of the inversion partition;
        # lanes that share the edit partition (no legal split) have none; the switch is optional
        back = [w for w in split.workers if w.stage == "back"]
        assert [w.wide.cus for w in back] == [list(range(64)) + list(range(128, 192)), list(range(64, 128)) + list(range(192, 256))]
        assert all(w.wide is None for w in pipe.workers)
        assert all(w.wide is None for w in ClipPipeline(m, plan="partition", edit_cus=128, edit_lanes=2, timestep_group=3,
                                                        widen_on_drain=False).workers)
        # 64-CU lanes and the 128-CU inversion partition take the tile tables swept on streams of that size (round 4)
        assert split.edit_lane_cus == 64 and split.workers[1].regime == "cus64" and split.workers[0].regime == "cus128"
        # the next clip's set-up stream is masked to the inversion partition (round 6); mask_prep=False: the unmasked queue of rounds 3-5
        lane_q = ClipPipeline(m, plan="partition", edit_cus=128, edit_lanes=2, timestep_group=3, codec_queue="lane")
        assert lane_q.workers[0].prep.cus == list(range(128, 256)) and lane_q.report()["setup_stream_masked_to_inversion_partition"]
        assert ClipPipeline(m, plan="partition", edit_cus=128, edit_lanes=2, timestep_group=3, codec_queue="lane",
                            mask_prep=False).workers[0].prep.cus is None
        with pytest.raises(ValueError):
            ClipPipeline(m, plan="lanes")
        v0, v1 = pipe.workers[0].view, pipe.workers[1].view
        assert v0.unet_weights is m.unet_weights and v0.state_dicts is m.state_dicts and v0.model is m.model
        assert v0._engines is not m._engines and v0._editors is not v1._editors
        pipe.warm_up(mels[0], *args)
        assert all(w.warm for w in pipe.workers)
        seen.clear()
        log.clear()
        got = pipe.edit_clips(mels, *args, seeds=[40, 41, 42])
        for (a, o, w), (a2, o2, w2) in zip(got, serial_b):
            assert torch.equal(a, a2) and torch.equal(o, o2) and torch.equal(w, w2)
        inv = [s for k, s in seen if k == "invert"]
        assert inv[0] == "chip" and set(inv[1:]) <= {"cus96-255", "chip"} and len(inv) == 3      # fill on the whole chip
        assert sum(1 for k, s in seen if k == "edit") == 3
        assert sum(1 for e in log if e[0] == "wait") >= 3                  # each back half waited for its front half
        rep = pipe.report()
        assert rep["plan"] == "partition" and rep["edit_cus"] == 96 and rep["inversion_cus"] == 160
        assert m.__dict__.get("_lane_stream") is None and m.editor(8, 8).lane_stream is None      # outside the pipeline
        monkeypatch.setattr(ed_cls, "invert", orig_invert)
        monkeypatch.setattr(ed_cls, "edit", orig_edit)
        with pytest.raises(RuntimeError, match="clip 0 failed"):
            pipe.edit_clips([torch.zeros(1, 1, 3, 5), mels[1]], *args)
        with pytest.raises(ValueError):
            ClipPipeline(m, plan="partition", edit_cus=256)
        with pytest.raises(ValueError):
            ClipPipeline(m, plan="partition", timestep_group=1)
        # ---- no per-clip seeds: one global generator stream, consumed in clip order
        torch.manual_seed(99)
        ref = [edit_clip(m, x0, *args, schedule="batched", timestep_group=3) for x0 in mels[:2]]
        after = torch.randn(2)
        torch.manual_seed(99)
        got = pipe.edit_clips(mels[:2], *args)
        for (a, o, w), (a2, o2, w2) in zip(got, ref):
            assert torch.equal(a, a2) and torch.equal(w, w2)
        assert torch.equal(torch.randn(2), after)           # and the generator ends where the serial loop leaves it
        assert "sample_xts_from_x0" not in pipe.workers[0].view.__dict__      # the gated draw hook is removed again
    finally:
        torch.set_num_threads(threads)


def test_clip_pipeline_codec_on_the_edit_lanes_on_cpu(cpu_stack, monkeypatch):
    """codec_queue="lane" (round 5): no codec stage -- the lane that edited a clip decodes it (VAE decode + vocoder on the lane's
    own stream), the next clip's set-up runs on the front stage's side stream; every clip equals the serial edit bit for bit."""
    log = []
    ClipPipeline = _fake_hip_for_pipeline(monkeypatch, log)
    threads = torch.get_num_threads()
    torch.set_num_threads(2)
    try:
        T, tstart = 3, 2
        m = _model(T)
        mels = [load_audio((synthetic_clip(seconds=0.32, seed=7 + i), 16000), m.get_fn_STFT(), device="cpu", stft=True)[0]
                for i in range(4)]
        args = (["a dog barking"], ["a cat meowing"], [""], [3.0], [12.0], T, tstart)
        serial = []
        for i, x0 in enumerate(mels):
            torch.manual_seed(40 + i)
            serial.append(edit_clip(m, x0, *args, schedule="batched", timestep_group=3))
        pipe = ClipPipeline(m, plan="partition", edit_cus=128, edit_lanes=2, timestep_group=3, codec_queue="lane")
        assert [w.stage for w in pipe.workers] == ["front", "back", "back"] and not pipe.codec_stage
        assert pipe.workers[0].prep is not None                      # the set-up of the next clip overlaps the running inversion
        with pytest.raises(ValueError):
            ClipPipeline(m, plan="partition", edit_cus=96, timestep_group=3, codec_queue="somewhere")
        pipe.warm_up(mels[0], *args)
        log.clear()
        got = pipe.edit_clips(mels, *args, seeds=[40 + i for i in range(4)])
        for (a, o, w), (a2, o2, w2) in zip(got, serial):
            assert torch.equal(a, a2) and torch.equal(o, o2) and torch.equal(w, w2)
        lanes = {w.lane.stream.name for w in pipe.workers if w.stage == "back"}
        assert lanes == {"cus0-63", "cus64-127"}
    finally:
        torch.set_num_threads(threads)


def test_noise_maps_of_a_clip_are_drawn_once_whoever_asks():
    """ClipPipeline._claim_noise: several lanes (and the helper thread running one clip ahead) may ask for the same clip's noise
    maps at the same moment; exactly one of them draws, at the clip's turn in the global draw order, and every asker sees that
    draw.  (Two draws of one clip would both pass the order gate and interleave on the global generator.)"""
    import threading
    from audioeditingcode_amd.pipeline import ClipPipeline, _DrawGate
    K, T, shape = 6, 5, (1, 2, 8, 4)
    seeds = [90 + i for i in range(K)]
    want = []
    for s in seeds:
        torch.manual_seed(s)
        want.append(torch.stack([torch.randn(shape) for _ in range(T)]))
    job = dict(prefetch={}, lock=threading.Lock(), gate=_DrawGate(), seeds=seeds)
    pipe = ClipPipeline.__new__(ClipPipeline)
    got, errs = {}, []

    def asker(i, helper):
        try:
            box = pipe._claim_noise(job, i, shape, T, helper=helper)
            box["ready"].wait(timeout=30)
            got.setdefault(i, []).append(box)
        except BaseException as e:               # noqa: BLE001
            errs.append(e)
    threads = [threading.Thread(target=asker, args=(i, h)) for i in reversed(range(K)) for h in (False, True, False)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
    assert not errs and all(not t.is_alive() for t in threads)
    for i in range(K):
        assert len(got[i]) == 3 and all(b is got[i][0] for b in got[i])              # one claim per clip
        assert "error" not in got[i][0] and torch.equal(got[i][0]["noise"], want[i]), i
    assert job["gate"].next == K


def test_export_model_images_on_cpu(monkeypatch, tmp_path):
    """image.export_model_images: the five engines of a wrapper (STFT, VAE encode, U-Net, VAE decode, vocoder) become five
    tape images whose named buffers have the engines' shapes; loaded into host memory, the STFT image's window / basis and
    the U-Net image's timestep table arrive intact."""
    from audioeditingcode_amd import _lib as L
    from audioeditingcode_amd.image import Image, export_model_images
    from conftest import install_cpu_stack
    real_lib = L.lib
    with monkeypatch.context() as mp:                   # engines are built on CPU under the test stack ...
        install_cpu_stack(mp)
        m = _model(4)
        out = export_model_images(m, str(tmp_path), n_samples=5120, unet_batch=2, ctx_len1=8)
    assert L.lib is real_lib                            # ... the images are loaded by the real libaed.so (host-memory mode)
    assert sorted(out) == ["stft.aedimg", "unet_b2.aedimg", "vae_decode.aedimg", "vae_encode.aedimg", "vocoder.aedimg"]
    assert out["unet_b2.aedimg"]["programs"] == ["context", "forward"]
    assert {"x_in", "eps", "ehs0", "ehs1", "bias1", "timesteps", "state"} <= set(out["unet_b2.aedimg"]["names"])
    im = Image(str(tmp_path / "unet_b2.aedimg"), host=True)
    ed = next(iter(m._editors.values()))
    _, nb = im.buffer("x_in")
    eng = next(iter(ed._unets.values()))
    assert nb == eng.x_in.numel() * 4
    assert torch.equal(im.copy_out("timesteps", torch.empty_like(ed.ts_dev)), ed.ts_dev)
    ops, n = im.program("forward")
    assert n == len(eng.tape.ops)
    im.close()
    voc = Image(str(tmp_path / "vocoder.aedimg"), host=True)
    assert voc.buffer("wav")[1] > 0 and voc.buffer("mel_in")[1] > 0
    voc.close()


def test_bench_parity_fixture_leg_plumbing_on_cpu(cpu_stack, monkeypatch, tmp_path):
    """bench.parity_fixture_leg (the reported-only T=200 parity of the bench line) on a stand-in fixture of the tiny model:
    keys, shapes, seeding and the skip rules of the function -- with a fixture made from the same model the three distances
    are zero; a fixture for another schedule or other prompts is skipped, not compared."""
    import importlib.util
    import os
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    T, tstart = 4, 3
    m = _model(T)
    src, tgt, neg = ["a dog barking"], ["a cat meowing"], [""]
    x0, _, _ = load_audio((synthetic_clip(seconds=0.64, seed=7), 16000), m.get_fn_STFT(), device="cpu", stft=True)
    torch.manual_seed(77)
    audio, _, w_edit = edit_clip(m, x0, src, tgt, neg, [3.0], [12.0], T, tstart)
    mel = m.vae_decode(w_edit)
    fx = dict(x0=x0.numpy(), w_edit=w_edit.numpy(), mel=mel.numpy(), wav=audio.numpy(), T=np.array(T), tstart=np.array(tstart),
              seed=np.array(77), clip=np.array(1), prompts=np.array([src[0], tgt[0], neg[0]]))
    golden = tmp_path / "tests" / "golden"
    golden.mkdir(parents=True)
    np.savez(golden / "bench_parity_T200.npz", **fx)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    out = bench.parity_fixture_leg(m, src, tgt, neg, T, tstart)
    assert out["latent_rel_l2"] == 0.0 and out["mel_rel_l2"] == 0.0 and out["waveform_rel_l2"] == 0.0, out
    assert "skipped" in bench.parity_fixture_leg(m, src, tgt, neg, T + 4, tstart)
    assert "skipped" in bench.parity_fixture_leg(m, ["another prompt"], tgt, neg, T, tstart)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path / "nowhere"))
    assert "skipped" in bench.parity_fixture_leg(m, src, tgt, neg, T, tstart)
