"""Text conditioning adapter: the transformers-backed encoders behind `wrapper.encode_text`.

The reference keeps text encoding on Hugging Face modules under PyTorch (it runs three times per clip, outside the
diffusion loops): AudioLDM-1 = CLAP text tower (models.py:511-537); AudioLDM2 = CLAP text features + T5 encoder ->
projection model -> GPT-2 autoregression of 8 continuous states (models.py:599-677); TANGO = T5 encoder
(models.py:462-467).  This module states those three call patterns over plain `transformers` modules (CLAP / T5 /
GPT-2 classes are installed; `diffusers` is not, so its two small pieces -- AudioLDM2ProjectionModel.forward and
AudioLDM2Pipeline.generate_language_model -- are restated here from their published semantics).  The conditioning
tensors it returns are INPUTS of the native path; nothing here runs inside the timed loops.

    enc = TextEncoders.from_pretrained(snapshot_dir, kind="audioldm2")      # needs the checkpoint on disk
    wrapper.text_encoders = enc                                             # load_model does this when it finds one
"""
import json
import os
from typing import List, Optional

import torch
import torch.nn.functional as F


def _pads_to_max_length(tokenizer) -> bool:
    """models.py:607: Roberta (CLAP) tokenizers pad to max_length, every other tokenizer pads to the longest prompt."""
    flag = getattr(tokenizer, "pad_to_max_length", None)
    if flag is not None:
        return bool(flag)
    try:
        from transformers import RobertaTokenizer, RobertaTokenizerFast
        return isinstance(tokenizer, (RobertaTokenizer, RobertaTokenizerFast))
    except ImportError:          # pragma: no cover
        return False


class ProjectionModel(torch.nn.Module):
    """diffusers' AudioLDM2ProjectionModel: two Linear maps into the language-model width, each sequence wrapped in
    learned SOS/EOS embeddings, then concatenated along the sequence (masks likewise, special tokens always attended)."""

    def __init__(self, text_encoder_dim=512, text_encoder_1_dim=1024, langauge_model_dim=768):
        super().__init__()
        self.projection = torch.nn.Linear(text_encoder_dim, langauge_model_dim)
        self.projection_1 = torch.nn.Linear(text_encoder_1_dim, langauge_model_dim)
        self.sos_embed = torch.nn.Parameter(torch.ones(langauge_model_dim))
        self.eos_embed = torch.nn.Parameter(torch.ones(langauge_model_dim))
        self.sos_embed_1 = torch.nn.Parameter(torch.ones(langauge_model_dim))
        self.eos_embed_1 = torch.nn.Parameter(torch.ones(langauge_model_dim))

    @staticmethod
    def _add_special(h, mask, sos, eos):
        b = h.shape[0]
        h = torch.cat([sos.expand(b, 1, -1), h, eos.expand(b, 1, -1)], dim=1)
        if mask is not None:
            one = mask.new_ones((b, 1))
            mask = torch.cat([one, mask, one], dim=-1)
        return h, mask

    def forward(self, hidden_states, hidden_states_1, attention_mask=None, attention_mask_1=None):
        h, m = self._add_special(self.projection(hidden_states), attention_mask, self.sos_embed, self.eos_embed)
        h1, m1 = self._add_special(self.projection_1(hidden_states_1), attention_mask_1, self.sos_embed_1,
                                   self.eos_embed_1)
        h = torch.cat([h, h1], dim=1)
        if m is None and m1 is not None:
            m = m1.new_ones(h.shape[0], h.shape[1] - m1.shape[1])
        if m1 is None and m is not None:
            m1 = m.new_ones(h.shape[0], h.shape[1] - m.shape[1])
        mask = None if m is None else torch.cat([m, m1], dim=-1)
        return h, mask


class StableAudioProjection(torch.nn.Module):
    """diffusers' StableAudioProjectionModel restated (un-vendored): the text states pass through `text_projection`
    (identity when the widths agree) and the start / end seconds through two number conditioners -- clamp to
    [min_value, max_value], normalise to [0, 1], learned Fourier features (plus the raw value), one Linear.  Parameter
    names follow the checkpoint (`{start,end}_number_conditioner.time_positional_embedding.{0.weights,1.weight,1.bias}`)."""

    def __init__(self, text_encoder_dim=768, conditioning_dim=768, min_value=0, max_value=512,
                 number_embedding_internal_dim=256, **_ignored):
        super().__init__()
        self.min_value, self.max_value = float(min_value), float(max_value)
        self.text_projection = (torch.nn.Identity() if text_encoder_dim == conditioning_dim
                                else torch.nn.Linear(text_encoder_dim, conditioning_dim))
        half = number_embedding_internal_dim // 2
        for n in ("start", "end"):
            self.register_parameter(f"{n}_weights", torch.nn.Parameter(torch.randn(half)))
            setattr(self, f"{n}_linear", torch.nn.Linear(number_embedding_internal_dim + 1, conditioning_dim))

    def load_diffusers_state_dict(self, sd):
        own = {}
        for n in ("start", "end"):
            pre = f"{n}_number_conditioner.time_positional_embedding."
            own[f"{n}_weights"] = sd[pre + "0.weights"]
            own[f"{n}_linear.weight"] = sd[pre + "1.weight"]
            own[f"{n}_linear.bias"] = sd[pre + "1.bias"]
        if "text_projection.weight" in sd:
            own["text_projection.weight"], own["text_projection.bias"] = sd["text_projection.weight"], sd["text_projection.bias"]
        self.load_state_dict(own)

    def _number(self, which, seconds):
        w, lin = getattr(self, f"{which}_weights"), getattr(self, f"{which}_linear")
        x = torch.as_tensor(seconds, dtype=torch.float32, device=w.device).reshape(-1).clamp(self.min_value, self.max_value)
        x = (x - self.min_value) / (self.max_value - self.min_value)
        t = x[..., None]
        fr = t * w[None] * 2 * torch.pi
        e = lin(torch.cat([t, fr.sin(), fr.cos()], dim=-1))
        return e.view(-1, 1, e.shape[-1])

    def forward(self, text_hidden_states=None, start_seconds=None, end_seconds=None):
        from types import SimpleNamespace
        return SimpleNamespace(
            text_hidden_states=None if text_hidden_states is None else self.text_projection(text_hidden_states),
            seconds_start_hidden_states=None if start_seconds is None else self._number("start", start_seconds),
            seconds_end_hidden_states=None if end_seconds is None else self._number("end", end_seconds))


class TextEncoders:
    """Holds the tokenizers / encoders of one model family and produces the `(hidden_states, class_labels, mask)`
    triple of the reference's `encode_text` for it."""

    def __init__(self, kind: str, tokenizer=None, text_encoder=None, tokenizer_2=None, text_encoder_2=None,
                 language_model=None, projection_model: Optional[ProjectionModel] = None, max_new_tokens: int = 8,
                 source: str = "caller-provided modules"):
        assert kind in ("audioldm", "audioldm2", "tango", "stable_audio")
        self.kind, self.source = kind, source
        self.tokenizer, self.text_encoder = tokenizer, text_encoder
        self.tokenizer_2, self.text_encoder_2 = tokenizer_2, text_encoder_2
        self.language_model, self.projection_model = language_model, projection_model
        self.max_new_tokens = max_new_tokens
        for mod in (text_encoder, text_encoder_2, language_model, projection_model):
            if mod is not None:
                mod.eval()

    # ------------------------------------------------------------------ loading
    @classmethod
    def from_pretrained(cls, root: str, kind: str, device="cpu"):
        """Load from a diffusers-style snapshot directory (tokenizer/, text_encoder/, tokenizer_2/, text_encoder_2/,
        language_model/, projection_model/).  Raises if a required sub-directory is missing."""
        from transformers import AutoTokenizer

        def need(sub):
            p = os.path.join(root, sub)
            if not os.path.isdir(p):
                raise FileNotFoundError(f"{root}: text-conditioning component '{sub}' is missing")
            return p
        if kind == "audioldm":
            from transformers import ClapTextModelWithProjection
            return cls(kind, AutoTokenizer.from_pretrained(need("tokenizer")),
                       ClapTextModelWithProjection.from_pretrained(need("text_encoder")).to(device), source=root)
        if kind == "tango":
            from transformers import T5EncoderModel
            return cls(kind, AutoTokenizer.from_pretrained(need("tokenizer")),
                       T5EncoderModel.from_pretrained(need("text_encoder")).to(device), source=root)
        if kind == "stable_audio":
            from transformers import T5EncoderModel
            proj_dir = need("projection_model")
            with open(os.path.join(proj_dir, "config.json")) as f:
                pc = json.load(f)
            proj = StableAudioProjection(**{k: v for k, v in pc.items() if not k.startswith("_")})
            wfile = os.path.join(proj_dir, "diffusion_pytorch_model.safetensors")
            if os.path.exists(wfile):
                from safetensors.torch import load_file
                proj.load_diffusers_state_dict(load_file(wfile))
            else:
                proj.load_diffusers_state_dict(torch.load(os.path.join(proj_dir, "diffusion_pytorch_model.bin"),
                                                          map_location="cpu", weights_only=True))
            return cls(kind, AutoTokenizer.from_pretrained(need("tokenizer")),
                       T5EncoderModel.from_pretrained(need("text_encoder")).to(device),
                       projection_model=proj.to(device), source=root)
        from transformers import ClapModel, GPT2Model, T5EncoderModel
        proj_dir = need("projection_model")
        with open(os.path.join(proj_dir, "config.json")) as f:
            pc = json.load(f)
        proj = ProjectionModel(pc.get("text_encoder_dim", 512), pc.get("text_encoder_1_dim", 1024),
                               pc.get("langauge_model_dim", 768))
        wfile = os.path.join(proj_dir, "diffusion_pytorch_model.safetensors")
        if os.path.exists(wfile):
            from safetensors.torch import load_file
            proj.load_state_dict(load_file(wfile))
        else:
            proj.load_state_dict(torch.load(os.path.join(proj_dir, "diffusion_pytorch_model.bin"), map_location="cpu",
                                            weights_only=True))
        lm = GPT2Model.from_pretrained(need("language_model")).to(device)
        return cls(kind, AutoTokenizer.from_pretrained(need("tokenizer")),
                   ClapModel.from_pretrained(need("text_encoder")).to(device),
                   AutoTokenizer.from_pretrained(need("tokenizer_2")),
                   T5EncoderModel.from_pretrained(need("text_encoder_2")).to(device), lm, proj.to(device),
                   max_new_tokens=getattr(lm.config, "max_new_tokens", None) or 8, source=root)

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _tokenize(tokenizer, prompts, padding):
        """The tokenizer call of models.py:512-529 / :606-625 incl. the truncation notice."""
        ti = tokenizer(prompts, padding=padding, max_length=tokenizer.model_max_length, truncation=True,
                       return_tensors="pt")
        untruncated = tokenizer(prompts, padding="longest", return_tensors="pt").input_ids
        if untruncated.shape[-1] >= ti.input_ids.shape[-1] and not torch.equal(ti.input_ids, untruncated):
            removed = tokenizer.batch_decode(untruncated[:, tokenizer.model_max_length - 1: -1])
            print(f"The following part of your input was truncated because the text encoder can only handle sequences "
                  f"up to {tokenizer.model_max_length} tokens: {removed}")
        return ti.input_ids, ti.attention_mask

    @staticmethod
    def _dev(module):
        return next(module.parameters()).device

    @torch.no_grad()
    def generate_language_model(self, inputs_embeds, attention_mask=None, max_new_tokens=None):
        """AudioLDM2Pipeline.generate_language_model: continuous autoregression -- the last hidden state of each pass
        is appended to the input embeddings (mask extended by ones); returns the `max_new_tokens` generated states."""
        n = max_new_tokens or self.max_new_tokens
        for _ in range(n):
            out = self.language_model(inputs_embeds=inputs_embeds, attention_mask=attention_mask, return_dict=True)
            inputs_embeds = torch.cat([inputs_embeds, out.last_hidden_state[:, -1:, :]], dim=1)
            if attention_mask is not None:
                attention_mask = torch.cat([attention_mask, attention_mask.new_ones((attention_mask.shape[0], 1))], -1)
        return inputs_embeds[:, -n:, :]

    # ------------------------------------------------------------------ the three families
    @torch.no_grad()
    def encode_audioldm(self, prompts: List[str], device, **kwargs):
        """models.py:511-537 -> (None, L2-normalised CLAP text embedding [P, 512], None)."""
        ids, mask = self._tokenize(self.tokenizer, prompts, "max_length")
        d = self._dev(self.text_encoder)
        emb = self.text_encoder(ids.to(d), attention_mask=mask.to(d))[0]
        return None, F.normalize(emb, dim=-1).to(device=device, dtype=torch.float32), None

    @torch.no_grad()
    def encode_tango(self, prompts: List[str], device, **kwargs):
        """models.py:462-467 (Tango.encode_text) -> (T5 states [P, L, 1024], None, boolean mask [P, L])."""
        ids, mask = self._tokenize(self.tokenizer, prompts, True)
        d = self._dev(self.text_encoder)
        emb = self.text_encoder(input_ids=ids.to(d), attention_mask=mask.to(d))[0]
        return emb.to(device=device, dtype=torch.float32), None, (mask == 1).to(device)

    @torch.no_grad()
    def encode_stable_audio(self, prompts: List[str], device, negative: bool = False, **kwargs):
        """StableAudWrapper.encode_text (models.py:1069-1103) -> (T5 states through the projection model [P, 128, 768],
        None, mask [P, 128]): max-length padding; for NEGATIVE prompts the padded positions are zeroed before the
        projection; the mask multiplies the result (twice in the reference -- idempotent for a 0/1 mask); the empty
        prompt returns zeros and no mask, which `unet_forward` reads as "zero the whole context"."""
        ti = self.tokenizer(prompts, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                            return_tensors="pt")
        d = self._dev(self.text_encoder)
        ids, mask = ti.input_ids.to(d), ti.attention_mask.to(d)
        e = self.text_encoder(ids, attention_mask=mask)[0]
        if negative:
            e = torch.where(mask.to(torch.bool).unsqueeze(2), e, 0.0)
        e = self.projection_model(text_hidden_states=e).text_hidden_states
        if prompts == [""]:
            return torch.zeros_like(e).to(device=device, dtype=torch.float32), None, None
        m = mask.unsqueeze(-1).to(e.dtype)
        return (e * m * m).to(device=device, dtype=torch.float32), None, mask.to(device)

    @torch.no_grad()
    def encode_duration(self, audio_start_in_s, audio_end_in_s, device):
        """StableAudioPipeline.encode_duration without classifier-free duplication (models.py:1161-1162):
        (seconds_start_hidden_states, seconds_end_hidden_states), each [1, 1, conditioning_dim]."""
        out = self.projection_model(start_seconds=[float(audio_start_in_s)], end_seconds=[float(audio_end_in_s)])
        return (out.seconds_start_hidden_states.to(device=device, dtype=torch.float32),
                out.seconds_end_hidden_states.to(device=device, dtype=torch.float32))

    @torch.no_grad()
    def encode_audioldm2(self, prompts: List[str], device, **kwargs):
        """models.py:599-677 -> (GPT-2 generated states [P, 8, 768], T5 states [P, L, 1024], T5 mask [P, L]).
        `negative=True` (and any other keyword) is accepted and ignored, as in the reference."""
        embeds, masks = [], []
        for tok, enc in ((self.tokenizer, self.text_encoder), (self.tokenizer_2, self.text_encoder_2)):
            ids, mask = self._tokenize(tok, prompts, "max_length" if _pads_to_max_length(tok) else True)
            d = self._dev(enc)
            ids, mask = ids.to(d), mask.to(d)
            if getattr(enc.config, "model_type", "") == "clap":
                e = enc.get_text_features(ids, attention_mask=mask)
                e = getattr(e, "pooler_output", e)          # tensor in every transformers release the reference supports
                e = e[:, None, :]                                   # (bs, hidden) -> (bs, 1, hidden)
                mask = mask.new_ones((len(prompts), 1))             # attend to this single state
            else:
                e = enc(ids, attention_mask=mask)[0]
            embeds.append(e)
            masks.append(mask)
        pd = self._dev(self.projection_model)
        proj, proj_mask = self.projection_model(embeds[0].to(pd), embeds[1].to(pd), masks[0].to(pd), masks[1].to(pd))
        ld = self._dev(self.language_model)
        generated = self.generate_language_model(proj.to(ld), attention_mask=proj_mask.to(ld))
        t5 = embeds[1].to(device=device, dtype=torch.float32)
        return generated.to(device=device, dtype=torch.float32), t5, masks[1].to(device)
