"""Cases of tests/golden/unet_forward_graph.npz: the reference's OWN inline U-Net forward graphs
(/root/reference/code/models.py:160-393 PipelineWrapper.unet_forward, :691-899 AudioLDM2Wrapper.unet_forward) executed by
oracle/make_golden.py `unet_graph` on stand-in diffusers blocks.  Shared by the oracle pin (CPU), the product's wrapper on the
CPU tape interpreter and the GPU parity test."""
import os

import numpy as np
import torch

from audioeditingcode_amd import configs, weights

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "unet_forward_graph.npz")
FAMILIES = ("audioldm", "tango", "audioldm2")
HOOKS = ("plain", "replace_h_space", "mid_add", "replace_skips", "zero_int", "zero_list", "combined")


def load():
    return np.load(PATH)


def family(g, fam):
    """(tiny family dict, U-Net state dict regenerated from the fixture's seed)."""
    f = configs.tiny_family(fam)
    return f, weights.random_state_dict(weights.unet_param_shapes(f["unet"]), seed=int(g[f"{fam}.seed"]))


def inputs(g, fam, size):
    """(x, t, wrapper-level conditioning kwargs exactly as the reference was called)."""
    key = f"{fam}.{size}"
    cond = {k[len(key) + 6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(key + ".cond.")}
    cond.setdefault("encoder_hidden_states", None)
    return torch.from_numpy(g[f"{key}.x"]), int(g[f"{key}.t"]), cond


def hook_kwargs(g, fam, hook):
    """The hook arguments of one case (reference keyword names)."""
    key = f"{fam}.even.hook"
    hs, add = torch.from_numpy(g[f"{key}.h_space_new"]), torch.from_numpy(g[f"{key}.mid_add"])
    rep = {1: [torch.from_numpy(g[f"{key}.replace.1.{j}"]) for j in range(3)]}
    return dict(plain={}, replace_h_space=dict(replace_h_space=hs), mid_add=dict(mid_block_additional_residual=add),
                replace_skips=dict(replace_skip_conns=rep), zero_int=dict(zero_out_resconns=3),
                zero_list=dict(zero_out_resconns=[0, 2]),
                combined=dict(replace_h_space=hs, mid_block_additional_residual=add, zero_out_resconns=[1]))[hook]


def oracle_kwargs(fam, cond):
    """Wrapper-level arguments -> oracle.unet.unet_forward's (AudioLDM2Wrapper's translation, models.py:706-710)."""
    if fam == "audioldm2":
        return dict(encoder_hidden_states=cond["encoder_hidden_states"], encoder_hidden_states_1=cond["class_labels"],
                    encoder_attention_mask_1=cond["encoder_attention_mask"])
    return {k: v for k, v in cond.items()}


def expected(g, fam, size, hook):
    key = f"{fam}.{size}.{hook}"
    out = dict(eps=torch.from_numpy(g[key + ".eps"]), h_space=torch.from_numpy(g[key + ".h_space"]))
    if hook == "plain":
        out["skips"] = {i: [torch.from_numpy(g[f"{key}.skip.{i}.{j}"]) for j in range(3)] for i in range(4)}
    else:
        out["skip_abs_sums"] = torch.from_numpy(g[key + ".skip_abs_sums"])
    return out


def cases():
    for fam in FAMILIES:
        for hook in HOOKS:
            yield fam, "even", hook
        yield fam, "odd", "plain"


def check(fam, size, hook, got, want, tol, rel):
    """got = (eps, h_space, skips dict of lists) in the reference's return convention."""
    eps, h_space, skips = got
    what = f"{fam}.{size}.{hook}"
    assert rel(eps, want["eps"]) < tol, (what, "eps", rel(eps, want["eps"]))
    assert rel(h_space, want["h_space"]) < tol, (what, "h_space", rel(h_space, want["h_space"]))
    assert sorted(skips.keys()) == [0, 1, 2, 3], what
    if "skips" in want:
        for i in range(4):
            assert len(skips[i]) == 3, what
            for a, b in zip(skips[i], want["skips"][i]):
                assert a.shape == b.shape and rel(a, b) < tol, (what, "skip", i)
    else:
        sums = torch.tensor([[float(s.abs().sum()) for s in skips[i]] for i in range(4)], dtype=torch.float64)
        assert torch.allclose(sums, want["skip_abs_sums"].double(), rtol=max(tol, 1e-4), atol=1e-3), (what, sums)
