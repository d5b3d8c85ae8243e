"""The wrapper API (models.PipelineWrapper subclasses, main_run.edit_clip, the hook arguments of unet_forward,
pc_drift on the real `unet_forward_pair`) executed on CPU: a test-only subclass lifts the CUDA requirement, tapes are
run by oracle/tape_interp.py and the three direct C-ABI calls by its FakeLib.  This is the smoke() flow of
__graft_entry__ and the hook tests of the GPU suite, without a GPU -- host logic only; the kernels are proven on the
GPU."""
import pytest
import torch

from audioeditingcode_amd import _lib as L
from audioeditingcode_amd import models, pc_drift
from audioeditingcode_amd.editing import EditEngine
from audioeditingcode_amd.main_run import edit_clip
from audioeditingcode_amd.tape import Tape
from audioeditingcode_amd.utils import get_text_embeddings, load_audio, synthetic_clip
from oracle import loops as oloops
from oracle import pc as opc
from oracle import tape_interp
from oracle import unet as ounet
from oracle import vae as ovae
from oracle.scheduler import OracleDDIMScheduler


class _CpuAudioLDM2(models.AudioLDM2Wrapper):
    def _require_device(self):           # test instrumentation only: the product class refuses a CPU device
        pass


def rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


@pytest.fixture
def cpu_stack(monkeypatch):
    fake = tape_interp.FakeLib()
    monkeypatch.setattr(Tape, "run", tape_interp.run_tape)
    monkeypatch.setattr(L, "lib", lambda: fake)
    monkeypatch.setattr(L, "current_stream_ptr", lambda: None)

    def run_graph(self, body, steps, use_graph=True, plan=None):
        for _ in range(steps):
            body()

    def sample_xts(self, x0, noise=None, generator=None):
        s = self.sched
        T = s.num_inference_steps
        x0 = x0.float()
        if noise is None:
            noise = torch.stack([torch.randn(x0.shape, generator=generator, dtype=torch.float32) for _ in range(T)])
        ts, abar = s.timesteps.cpu(), s.alphas_cumprod
        t_rows = torch.stack([ts[T - (r + 1)] for r in range(T)])
        shape = (T, *[1] * x0.dim())
        return torch.cat([x0[None], x0[None] * (abar[t_rows] ** 0.5).reshape(shape)
                          + noise * ((1 - abar) ** 0.5)[t_rows].reshape(shape)])
    monkeypatch.setattr(EditEngine, "_run_graph", run_graph)
    monkeypatch.setattr(EditEngine, "sample_xts", sample_xts)


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
