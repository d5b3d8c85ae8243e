"""TEST INFRASTRUCTURE -- stand-in pipeline members for pinning A15 (`encode_text`) to the reference's own code.

`oracle/make_golden.py text` runs the reference's `AudioLDMWrapper.encode_text` (models.py:511-537),
`AudioLDM2Wrapper.encode_text` (:599-677) and `TangoWrapper.encode_text` (:455-472) from /root/reference with the members
built here in place of the Hugging Face checkpoints (none exist offline): deterministic word tokenizers (the CLAP one IS a
`RobertaTokenizer` instance, because models.py:607 switches its padding rule on that), random-init `ClapModel` /
`ClapTextModelWithProjection` / `T5EncoderModel` / `GPT2Model` of reduced width (real transformers classes), and the three
pieces that live in un-vendored third-party code, restated from their published definitions:
  * diffusers `AudioLDM2ProjectionModel.forward` (SOS/EOS-wrapped projections, concatenated, special tokens attended);
  * diffusers `AudioLDM2Pipeline.generate_language_model` (continuous autoregression of the language model);
  * declare-lab tango `AudioDiffusion.encode_text` (tokenize with padding=True, T5 states, boolean mask).
PARITY NOTE: those three are restated on both sides (diffusers / tango are absent, requirements.txt:1 unpinned); what the
fixture pins is everything the REFERENCE does around them -- tokenizer padding rules, the CLAP branch (features -> one
attended state), which embeddings / masks come back in which slot, normalisation, dtype / device moves.

The same builders are used by the tests to rebuild identical modules (seeded init), so only tensors travel in the fixture.
"""
from types import SimpleNamespace

import torch
from transformers import RobertaTokenizer


def _word_ids(prompt, vocab):
    return [3 + (sum(map(ord, w)) % (vocab - 4)) for w in prompt.split()]


class _TokBase:
    """Whitespace 'tokenizer' with the call signature models.py uses (padding= True | 'longest' | 'max_length')."""
    bos = eos = None
    pad_id = 1

    def _init(self, model_max_length, vocab):
        object.__setattr__(self, "model_max_length", model_max_length)
        object.__setattr__(self, "n_vocab", vocab)

    def _ids(self, p):
        ids = _word_ids(p, self.n_vocab)
        return ([self.bos] if self.bos is not None else []) + ids + [self.eos]

    def __call__(self, prompts, padding=True, max_length=None, truncation=False, return_tensors="pt"):
        seqs = [self._ids(p) for p in prompts]
        if truncation and max_length:
            seqs = [s[:max_length - 1] + [self.eos] if len(s) > max_length else s for s in seqs]
        L = max_length if padding == "max_length" else max(len(s) for s in seqs)
        ids = torch.full((len(seqs), L), self.pad_id, dtype=torch.long)
        mask = torch.zeros(len(seqs), L, dtype=torch.long)
        for i, s in enumerate(seqs):
            ids[i, :len(s)] = torch.tensor(s)
            mask[i, :len(s)] = 1
        return SimpleNamespace(input_ids=ids, attention_mask=mask)

    def batch_decode(self, ids):
        return [" ".join(map(str, r.tolist())) for r in ids]


class ClapWordTokenizer(_TokBase, RobertaTokenizer):
    """A RobertaTokenizer INSTANCE (isinstance check of models.py:607 -> padding='max_length') without vocabulary files:
    <s> words </s>, pad id 1 (Roberta's)."""
    bos, eos, pad_id = 0, 2, 1

    def __init__(self, model_max_length=12, vocab=97):          # deliberately no super().__init__: no files on disk
        self._init(model_max_length, vocab)

    def __getattr__(self, name):            # transformers' tokenizer base would look up missing special-token state
        raise AttributeError(name)


class T5WordTokenizer(_TokBase):
    """Not a Roberta tokenizer -> padding=True (longest): words </s>, pad id 0 (T5's)."""
    bos, eos, pad_id = None, 1, 0

    def __init__(self, model_max_length=20, vocab=97):
        self._init(model_max_length, vocab)


TEXT_CFG = dict(vocab_size=97, hidden_size=32, num_hidden_layers=2, num_attention_heads=4, intermediate_size=64,
                max_position_embeddings=40, projection_dim=16, pad_token_id=1)
AUDIO_CFG = dict(spec_size=64, patch_size=4, patch_stride=[4, 4], num_mel_bins=16, depths=[1, 1],
                 num_attention_heads=[1, 2], patch_embeds_hidden_size=8, hidden_size=16, window_size=4, num_classes=4,
                 enable_fusion=False, projection_dim=16)


def clap_text_with_projection(seed=11):
    """AudioLDM-1's text_encoder class (ClapTextModelWithProjection), reduced width, seeded init."""
    from transformers import ClapTextConfig, ClapTextModelWithProjection
    torch.manual_seed(seed)
    return ClapTextModelWithProjection(ClapTextConfig(**TEXT_CFG)).eval()


def clap_model(seed=12):
    """AudioLDM2's text_encoder class (ClapModel), reduced width, seeded init."""
    from transformers import ClapConfig, ClapModel
    torch.manual_seed(seed)
    return ClapModel(ClapConfig(text_config=dict(TEXT_CFG), audio_config=dict(AUDIO_CFG), projection_dim=16)).eval()


def clap_model_v4_api(seed=12):
    """The same ClapModel behind the transformers-4 signature the reference was written against
    (`get_text_features` returns the normalised, projected feature TENSOR; transformers 5 wraps it in an output object
    whose `pooler_output` holds exactly that tensor)."""
    m = clap_model(seed)
    inner = m.get_text_features

    def get_text_features(input_ids, attention_mask=None, **kw):
        out = inner(input_ids, attention_mask=attention_mask, **kw)
        return getattr(out, "pooler_output", out)
    m.get_text_features = get_text_features
    return m


def t5_encoder(seed=13, d=32):
    from transformers import T5Config, T5EncoderModel
    torch.manual_seed(seed)
    return T5EncoderModel(T5Config(vocab_size=97, d_model=d, d_kv=8, d_ff=64, num_layers=2, num_heads=4)).eval()


def gpt2(seed=14, d=24):
    from transformers import GPT2Config, GPT2Model
    torch.manual_seed(seed)
    return GPT2Model(GPT2Config(vocab_size=50, n_positions=64, n_embd=d, n_layer=2, n_head=4)).eval()


def projection_weights(seed=15, d_clap=16, d_t5=32, d_lm=24):
    """Parameter dict in the checkpoint's naming (`projection`, `projection_1`, `{sos,eos}_embed[_1]`)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g) * 0.3      # noqa: E731
    return {"projection.weight": r(d_lm, d_clap), "projection.bias": r(d_lm), "projection_1.weight": r(d_lm, d_t5),
            "projection_1.bias": r(d_lm), "sos_embed": r(d_lm), "eos_embed": r(d_lm), "sos_embed_1": r(d_lm),
            "eos_embed_1": r(d_lm)}


def audioldm2_projection_forward(w, hidden_states, hidden_states_1, attention_mask=None, attention_mask_1=None):
    """diffusers AudioLDM2ProjectionModel.forward restated over a plain weight dict (independent of the product's
    nn.Module statement of the same definition)."""
    def wrap(h, mask, sos, eos):
        b = h.shape[0]
        h = torch.cat([sos.expand(b, 1, -1), h, eos.expand(b, 1, -1)], 1)
        if mask is not None:
            mask = torch.cat([mask.new_ones(b, 1), mask, mask.new_ones(b, 1)], -1)
        return h, mask
    h = hidden_states @ w["projection.weight"].T + w["projection.bias"]
    h, m = wrap(h, attention_mask, w["sos_embed"], w["eos_embed"])
    h1 = hidden_states_1 @ w["projection_1.weight"].T + w["projection_1.bias"]
    h1, m1 = wrap(h1, attention_mask_1, w["sos_embed_1"], w["eos_embed_1"])
    hs = torch.cat([h, h1], 1)
    if m is None and m1 is not None:
        m = m1.new_ones(h.shape[:2])
    elif m is not None and m1 is None:
        m1 = m.new_ones(h1.shape[:2])
    mask = None if m is None else torch.cat([m, m1], -1)
    return SimpleNamespace(hidden_states=hs, attention_mask=mask)


def generate_language_model(language_model, inputs_embeds, attention_mask=None, max_new_tokens=None):
    """diffusers AudioLDM2Pipeline.generate_language_model restated: each pass appends the last hidden state."""
    n = max_new_tokens if max_new_tokens is not None else 8          # language_model.config.max_new_tokens of the checkpoint
    for _ in range(n):
        out = language_model(inputs_embeds=inputs_embeds, attention_mask=attention_mask, return_dict=True)
        inputs_embeds = torch.cat([inputs_embeds, out.last_hidden_state[:, -1:, :]], 1)
        if attention_mask is not None:
            attention_mask = torch.cat([attention_mask, attention_mask.new_ones(attention_mask.shape[0], 1)], 1)
    return inputs_embeds[:, -n:, :]


def tango_encode_text(tokenizer, text_encoder, prompt):
    """declare-lab tango `AudioDiffusion.encode_text` restated (frozen text encoder branch)."""
    batch = tokenizer(prompt, max_length=tokenizer.model_max_length, padding=True, truncation=True, return_tensors="pt")
    with torch.no_grad():
        states = text_encoder(input_ids=batch.input_ids, attention_mask=batch.attention_mask)[0]
    return states, (batch.attention_mask == 1)


def param_checksum(module):
    """One float64 per module: detects a seeded re-initialisation that did not reproduce the fixture's weights."""
    return float(sum(p.detach().double().abs().sum() for p in module.parameters()))


PROMPT_SETS = [["a dog barking loudly in the rain", "jazz"], [""], ["a recording of a piano melody"]]
