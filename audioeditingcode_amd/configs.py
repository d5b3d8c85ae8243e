"""Architecture configs of the model families behind the reference's wrappers.

The reference reads these from the HF hub at `from_pretrained` time (models.py:478, :556-564);
no checkpoint or config.json exists in this container, so the dictionaries below restate the
PUBLIC configs from memory ([unverified]) and are only the fallback: when a
real model directory is on disk (`weights.find_checkpoint`), its config.json files win.
Structural check: the AudioLDM2 U-Net layout below totals 346.8 M parameters (the paper's 346 M),
AudioLDM-S 184.7 M (in-tree twin: 185.0 M).
"""
import copy

SCHEDULER = dict(num_train_timesteps=1000, beta_start=0.0015, beta_end=0.0195, beta_schedule="scaled_linear",
                 set_alpha_to_one=False, steps_offset=1, prediction_type="epsilon", timestep_spacing="leading",
                 clip_sample=False)

UNET_AUDIOLDM2 = dict(
    in_channels=8, out_channels=8, block_out_channels=[128, 256, 384, 640], layers_per_block=2,
    down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
    up_block_types=["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"],
    cross_attention_dim=[[None, 768, 1024]] * 4, attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5,
    flip_sin_to_cos=True, freq_shift=0, class_embed_type=None, use_linear_projection=False, sample_size=128)

UNET_AUDIOLDM = dict(
    in_channels=8, out_channels=8, block_out_channels=[128, 256, 384, 640], layers_per_block=2,
    down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
    up_block_types=["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"],
    cross_attention_dim=[128, 256, 384, 640], attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5,
    flip_sin_to_cos=True, freq_shift=0, class_embed_type="simple_projection",
    projection_class_embeddings_input_dim=512, class_embeddings_concat=True, use_linear_projection=False,
    sample_size=128)

# TANGO = Stable-Diffusion-2.1 style UNet2DConditionModel, 8 latent channels, T5 (1024-d) cross attention
UNET_TANGO = dict(
    in_channels=8, out_channels=8, block_out_channels=[320, 640, 1280, 1280], layers_per_block=2,
    down_block_types=["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    up_block_types=["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],
    cross_attention_dim=1024, attention_head_dim=[5, 10, 20, 20], norm_num_groups=32, norm_eps=1e-5,
    flip_sin_to_cos=True, freq_shift=0, class_embed_type=None, use_linear_projection=True, sample_size=32)

VAE_AUDIOLDM = dict(in_channels=1, out_channels=1, latent_channels=8, block_out_channels=[128, 256, 512],
                    layers_per_block=2, norm_num_groups=32, scaling_factor=0.9227914214134216)

VOCODER_AUDIOLDM = dict(model_in_dim=64, sampling_rate=16000, upsample_initial_channel=1024,
                        upsample_rates=[5, 4, 2, 2, 2], upsample_kernel_sizes=[16, 16, 8, 4, 4],
                        resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3,
                        leaky_relu_slope=0.1, normalize_before=False)

STFT_AUDIOLDM = dict(filter_length=1024, hop_length=160, win_length=1024, n_mel_channels=64, sampling_rate=16000,
                     mel_fmin=0, mel_fmax=8000)

# ---- Stable Audio Open 1.0 (StableAudWrapper, models.py:1051-1354; BASELINE config 5).  Public configs restated from
# memory ([unverified]); a local checkpoint's config.json files win.
DIT_STABLE_AUDIO = dict(sample_size=1024, in_channels=64, out_channels=64, num_layers=24, attention_head_dim=64,
                        num_attention_heads=24, num_key_value_attention_heads=12, cross_attention_dim=768,
                        time_proj_dim=256, global_states_input_dim=1536, cross_attention_input_dim=768)
OOBLECK_STABLE_AUDIO = dict(encoder_hidden_size=128, downsampling_ratios=[2, 4, 4, 8, 8],
                            channel_multiples=[1, 2, 4, 8, 16], decoder_channels=128, decoder_input_channels=64,
                            audio_channels=2, sampling_rate=44100)
SCHEDULER_COSINE_DPM = dict(sigma_min=0.3, sigma_max=500.0, sigma_data=1.0, sigma_schedule="exponential",
                            num_train_timesteps=1000, solver_order=2, prediction_type="v_prediction", rho=7.0,
                            solver_type="midpoint", lower_order_final=True, euler_at_final=False,
                            final_sigmas_type="zero")
PROJECTION_STABLE_AUDIO = dict(text_encoder_dim=768, conditioning_dim=768, min_value=0, max_value=512,
                               number_embedding_internal_dim=256)
STFT_STABLE_AUDIO = dict(filter_length=1024, hop_length=160, win_length=1024, n_mel_channels=64, sampling_rate=44100,
                         mel_fmin=0, mel_fmax=22050)           # models.py:1105-1115 (spectrogram plots only)

FAMILIES = {
    "audioldm2": dict(unet=UNET_AUDIOLDM2, vae=VAE_AUDIOLDM, vocoder=VOCODER_AUDIOLDM, scheduler=SCHEDULER,
                      stft=STFT_AUDIOLDM, ctx=dict(kind="audioldm2", gpt2_dim=768, gpt2_len=8, t5_dim=1024)),
    "audioldm": dict(unet=UNET_AUDIOLDM, vae=VAE_AUDIOLDM, vocoder=VOCODER_AUDIOLDM, scheduler=SCHEDULER,
                     stft=STFT_AUDIOLDM, ctx=dict(kind="audioldm", clap_dim=512)),
    "tango": dict(unet=UNET_TANGO, vae=VAE_AUDIOLDM, vocoder=VOCODER_AUDIOLDM, scheduler=SCHEDULER,
                  stft=STFT_AUDIOLDM, ctx=dict(kind="tango", t5_dim=1024)),
    "stable_audio": dict(dit=DIT_STABLE_AUDIO, oobleck=OOBLECK_STABLE_AUDIO, scheduler=SCHEDULER_COSINE_DPM,
                         projection=PROJECTION_STABLE_AUDIO, stft=STFT_STABLE_AUDIO,
                         ctx=dict(kind="stable_audio", t5_dim=768, max_length=128)),
}


def family_of(model_id):
    """Substring dispatch of load_model (models.py:1357-1374)."""
    if "tango" in model_id:
        return "tango"
    if "audioldm2" in model_id:
        return "audioldm2"
    if "audioldm" in model_id:
        return "audioldm"
    if "stable-audio" in model_id:
        return "stable_audio"
    raise NotImplementedError(f"{model_id}: only the AudioLDM / AudioLDM2 / TANGO / Stable Audio wrappers are in scope "
                              f"(image models are out of scope)")


def get_family(model_id):
    if model_id.startswith("tiny/"):          # reduced-width twins for tests: "tiny/audioldm2", ...
        return tiny_family(family_of(model_id))
    fam = copy.deepcopy(FAMILIES[family_of(model_id)])
    # the -m- / -l- AudioLDM variants widen the U-Net (audioldm/utils.py:195-200)
    if family_of(model_id) == "audioldm":
        if "-l-" in model_id:
            fam["unet"]["block_out_channels"] = [256, 512, 768, 1280]
        elif "-m-" in model_id:
            fam["unet"]["block_out_channels"] = [192, 384, 576, 960]
        fam["unet"]["cross_attention_dim"] = list(fam["unet"]["block_out_channels"])
    return fam


def tiny_family(kind="audioldm2"):
    """Reduced-width family with the same graph, for parity tests that must run in seconds."""
    fam = copy.deepcopy(FAMILIES[kind])
    if kind == "stable_audio":
        fam["dit"].update(sample_size=32, in_channels=8, out_channels=8, num_layers=2, attention_head_dim=32,
                          num_attention_heads=4, num_key_value_attention_heads=2, cross_attention_dim=64,
                          time_proj_dim=64, global_states_input_dim=128, cross_attention_input_dim=64)
        fam["oobleck"].update(encoder_hidden_size=32, downsampling_ratios=[2, 4], channel_multiples=[1, 2],
                              decoder_channels=32, decoder_input_channels=8, sampling_rate=800)
        fam["projection"].update(text_encoder_dim=64, conditioning_dim=64, max_value=8, number_embedding_internal_dim=16)
        fam["ctx"].update(t5_dim=64, max_length=6)
        return fam
    u = fam["unet"]
    u["block_out_channels"] = [32, 64, 96, 128] if kind != "tango" else [32, 64, 128, 128]
    if kind == "audioldm2":
        u["cross_attention_dim"] = [[None, 48, 64]] * 4
        u["attention_head_dim"] = 2
        fam["ctx"].update(gpt2_dim=48, t5_dim=64)
    elif kind == "audioldm":
        u["cross_attention_dim"] = list(u["block_out_channels"])
        u["attention_head_dim"] = 2
        u["projection_class_embeddings_input_dim"] = 24
        fam["ctx"].update(clap_dim=24)
    else:
        u["cross_attention_dim"] = 64
        u["attention_head_dim"] = [2, 2, 4, 4]
        fam["ctx"].update(t5_dim=64)
    u["norm_num_groups"] = 8
    fam["vae"]["block_out_channels"] = [32, 64, 64]
    fam["vae"]["norm_num_groups"] = 8
    fam["vocoder"]["upsample_initial_channel"] = 64
    return fam
