#!/usr/bin/env python
"""bench.py -- edited-clips/sec of the DDPM-inversion audio-editing hot path on MI355X.

Workload (BASELINE.json configs[1]): AudioLDM2 text-based edit of ONE synthetic 10 s / 16 kHz clip per
step -- STFT/log-mel -> VAE encode -> 200-step edit-friendly DDPM inversion (cfg 3) -> 100-step edit from
tstart=100 (cfg 12) -> VAE decode -> HiFi-GAN vocoder (edited + original, main_run.py:184-185) -- fp32,
seeded-random weights of the real architecture (no checkpoints exist offline), synthetic conditioning.
One "step" = one whole clip.  N GPUs = N independent clips per step (weak scaling, no data-path
collective; weights broadcast once from rank 0 over RCCL, edited latents gathered to rank 0).

Headline schedule (round 3, `--plan partition`): a two-stage clip pipeline (pipeline.ClipPipeline) -- while clip i runs
its latency-bound 100-step edit loop on one CU partition, clip i+1 runs its timestep-batched forward inversion on the
complementary partition (hardware queues with disjoint CU masks); every clip's own launches, order and values are those of
the clip edited alone.  Reported beside it: the same clips one at a time in the reference order (`value_reference_order`),
one at a time with the timestep-batched inversion (`value_single_clip_batched`, the round-1/2 headline), parity against the
CPU oracle (`parity`), and BASELINE configs 3/4/5 as sub-benchmarks.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: fp32-in MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0         # same guide: dense bf16 MFMA peak; six bf16 products per fp32-equivalent product
PEAK_HBM_GBS = 8000.0                  # same guide: HBM3E peak (6.29 TB/s measured by a float4 copy)
ARITH_TEXT = {"f32": "f32 (v_mfma_f32_32x32x2_f32 on fp32 operands)",
              "bf16x6": "bf16x6 (exact 3-way bf16 split of every fp32 operand in the loader, 6 bf16 MFMA piece products, fp32 "
                        "accumulate; operands and results fp32 in HBM; the latency-regime lin_gemm kernels, attention, norms and "
                        "step math stay on fp32 instructions)"}
T_START = time.time()


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def csrc_hash():
    """sha1 over the GEMM-family sources: ties a committed PMC traffic summary to the kernels being benched."""
    h = hashlib.sha1()
    for f in ("conv_gemm.hip", "conv_gemm_x6.hip", "lin_gemm.hip", "cg_params.h", "aed_common.h"):   # the family `traffic` refers to
        h.update(open(os.path.join(ROOT, "audioeditingcode_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


def spawn_ranks(n):
    """`python bench.py --gpus N` without a torchrun environment: launch the N ranks ourselves (one process per GPU)
    and relay rank 0's JSON line.  The driver's own torchrun launch sets WORLD_SIZE and never gets here."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    log(f"--gpus {n} without WORLD_SIZE: spawning {n} ranks through torch.distributed.run (port {port})")
    return subprocess.call(cmd, env=env)


def cpu_baseline(fam, state_dicts, cores, anchor_steps=50):
    """The oracle (CPU restatement pinned to the reference, oracle/) timed on the host cores on a bounded
    sample: 1 warm + 3 timed AudioLDM2 U-Net forwards (B=1), one VAE encode/decode, one vocoder call, one
    STFT; a clip is 600 U-Net forwards + enc + dec + 2 vocoder + STFT."""
    from oracle import audio as oaudio, hifigan as ohifi, unet as ounet, vae as ovae
    from oracle.synth import chirp_waveform
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle on {cores} host threads")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 8, 256, 16, generator=g)
    kw = dict(encoder_hidden_states=torch.randn(1, 8, 768, generator=g),
              encoder_hidden_states_1=torch.randn(1, 16, 1024, generator=g),
              encoder_attention_mask_1=torch.ones(1, 16))
    with torch.no_grad():
        ounet.unet_forward(fam["unet"], state_dicts["unet"], x, torch.tensor(500), **kw)
        t0 = time.time()
        for _ in range(3):
            ounet.unet_forward(fam["unet"], state_dicts["unet"], x, torch.tensor(500), **kw)
        t_unet = (time.time() - t0) / 3
        wav = torch.from_numpy(oaudio.prepare_waveform(chirp_waveform().numpy(), 163840))[None]
        t0 = time.time()
        mel, _, _ = oaudio.mel_spectrogram(wav)
        t_stft = time.time() - t0
        mel4 = mel[:, :, :1024].transpose(1, 2)[:, None]
        t0 = time.time()
        lat = ovae.vae_encode(fam["vae"], state_dicts["vae"], mel4)
        t_enc = time.time() - t0
        t0 = time.time()
        rec = ovae.vae_decode(fam["vae"], state_dicts["vae"], lat)
        t_dec = time.time() - t0
        t0 = time.time()
        ohifi.hifigan_forward(fam["vocoder"], state_dicts["vocoder"], rec[:, 0])
        t_voc = time.time() - t0
    clip_s = 600 * t_unet + t_enc + t_dec + 2 * t_voc + t_stft
    out = dict(value=1.0 / clip_s, unit="edited-clips/sec", cores=cores, kind="port",
               sample=f"oracle (torch CPU fp32): 3 AudioLDM2 U-Net fwd B=1 ({t_unet:.3f} s each), VAE enc "
                      f"{t_enc:.2f} s, dec {t_dec:.2f} s, vocoder {t_voc:.2f} s, STFT {t_stft:.3f} s; "
                      f"clip = 600*unet + enc + dec + 2*voc + stft = {clip_s:.1f} s (extrapolated)")
    if anchor_steps > 0:
        out["config1_anchor"] = cpu_config1_anchor(anchor_steps, cores)
    return out


def cpu_config1_anchor(T, cores):
    """BASELINE configs[0], un-extrapolated: ONE whole clip through the oracle on the host cores -- AudioLDM-S
    (cvssp/audioldm-s-full architecture, seeded-random weights), --mode ddim, T DDIM steps (ddim_inversion.py:44-84:
    T inversion + T sampling steps, two U-Net forwards each), VAE encode/decode, vocoder twice, STFT."""
    from audioeditingcode_amd import configs, weights
    from oracle import audio as oaudio, hifigan as ohifi, loops as oloops, unet as ounet, vae as ovae
    from oracle.scheduler import OracleDDIMScheduler
    from oracle.synth import chirp_waveform
    fam = configs.get_family("cvssp/audioldm-s-full")
    sds = {k: weights.random_state_dict(fn(fam[k]), seed=i) for i, (k, fn) in enumerate(
        (("unet", weights.unet_param_shapes), ("vae", weights.vae_param_shapes), ("vocoder", weights.vocoder_param_shapes)))}
    g = torch.Generator().manual_seed(3)
    clap = torch.nn.functional.normalize(torch.randn(3, 1, 512, generator=g), dim=-1)        # src / tgt / uncond
    sched = OracleDDIMScheduler()
    sched.set_timesteps(T)
    n_fwd = [0]

    def unet_fn(x, t, cond):
        n_fwd[0] += x.shape[0]
        return ounet.unet_forward(fam["unet"], sds["unet"], x, t, class_labels=cond.expand(x.shape[0], -1))[0]
    ow = oloops.OracleWrapper(sched, unet_fn)
    t0 = time.time()
    with torch.no_grad():
        wav = torch.from_numpy(oaudio.prepare_waveform(chirp_waveform().numpy(), 163840))[None]
        mel, _, _ = oaudio.mel_spectrogram(wav)
        mel4 = mel[:, :, :1024].transpose(1, 2)[:, None]
        w0 = ovae.vae_encode(fam["vae"], sds["vae"], mel4)
        wT = oloops.ddim_invert(ow, w0, clap[0], clap[2], 3.0, T, 0)
        w_e = oloops.ddim_sample(ow, wT, clap[1], clap[2], 12.0, 0)
        rec = ovae.vae_decode(fam["vae"], sds["vae"], w_e)
        ohifi.hifigan_forward(fam["vocoder"], sds["vocoder"], rec[:, 0])
        ohifi.hifigan_forward(fam["vocoder"], sds["vocoder"], mel4[:, 0])
    dt = time.time() - t0
    full = 50                                      # BASELINE configs[0]: 50 DDIM steps
    out = dict(workload=f"AudioLDM-S (185 M U-Net, seeded-random weights), --mode ddim, T={T}, one 10 s clip, "
                        f"{n_fwd[0]} U-Net sample-forwards + codec, oracle on {cores} host threads, measured end to end",
               seconds=dt, clips_per_sec=1.0 / dt, finite=bool(torch.isfinite(w_e).all()))
    if T != full:                                  # bounded sample: the 50-step clip costs T-proportionally more U-Net work
        out["seconds_at_50_steps_extrapolated"] = dt * full / T
        out["note"] = (f"bounded CPU sample ({T} of BASELINE configs[0]'s {full} DDIM steps; round 3 measured the full 50-step clip "
                       f"at 83.0 s on this kind of host)")
    return out


def clip_phases(m, fn, wave, src, tgt, neg, args):
    """Wall time of every phase of one clip (device sync after each); the edit loop also as HIP-event time."""
    from audioeditingcode_amd.ddm_inversion.inversion_utils import inversion_forward_process, inversion_reverse_process
    acc = {}

    def tick(label, t_prev):
        torch.cuda.synchronize()
        t = time.perf_counter()
        acc[label] = round(1e3 * (t - t_prev), 2)
        return t
    torch.manual_seed(4242)
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.inference_mode():
        mel, _, _ = fn.mel_spectrogram(wave)
        x0 = mel[0].T[:1024][None, None].contiguous()
        t = tick("stft_mel", t)
        w0 = m.vae_encode(x0)
        t = tick("vae_encode", t)
        _, zs, wts, _ = inversion_forward_process(m, w0, etas=1.0, prompts=src, cfg_scales=[3.0],
                                                  num_inference_steps=args.T, numerical_fix=True,
                                                  schedule=args.schedule, timestep_group=args.group)
        t = tick("inversion (text enc + x_t draws + U-Net loop)", t)
        w_e, _ = inversion_reverse_process(m, xT=wts, tstart=torch.tensor([args.tstart]), etas=1.0, prompts=tgt,
                                           neg_prompts=neg, cfg_scales=[12.0], zs=zs[:args.tstart])
        t = tick("edit (text enc + U-Net loop)", t)
        acc["edit loop on the device (HIP events)"] = round(m.editor(256, 16).last_loop_ms(), 2)
        x0_dec = m.vae_decode(w_e)
        t = tick("vae_decode", t)
        m.decode_to_mel(x0_dec)
        m.decode_to_mel(x0)
        tick("vocoder x2", t)
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model_id", default="cvssp/audioldm2")
    ap.add_argument("--T", type=int, default=200)
    ap.add_argument("--tstart", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the extra timestep-batched-inversion timing")
    ap.add_argument("--group", type=int, default=100,
                    help="timesteps per U-Net call in the batched inversion (measured on the MI355X: U-Net batch 40 / 80 / 200\n"
                         "-> 85 / 92 / 99 TFLOP/s per forward; batch 200 holds ~24 GB of activations)")
    ap.add_argument("--schedule", default="batched", choices=["batched", "sequential"],
                    help="headline schedule of the forward inversion (the other one is timed as an extra)")
    ap.add_argument("--profile-forward", action="store_true", help="only run U-Net forwards (for rocprofv3)")
    ap.add_argument("--clips-per-gpu", type=int, default=1,
                    help="independent clips edited together per step and GPU (BASELINE configs[2]: 8; default: the "
                         "headline configs[1] shape, 1)")
    ap.add_argument("--plan", default="partition", choices=["partition", "serial"],
                    help="how clips share the GPU (pipeline.ClipPipeline): 'partition' = clip i's edit loop and clip i+1's "
                         "inversion on disjoint CU partitions; 'serial' = one clip at a time")
    ap.add_argument("--edit-cus", type=int, default=128, help="partition plan: CUs of the edit-loop partition")
    ap.add_argument("--edit-lanes", type=int, default=2,
                    help="partition plan: concurrent edit loops (disjoint CU slices of the edit partition where they are "
                         "multiples of 32 CUs, shared otherwise)")
    ap.add_argument("--codec-queue", default="lane", choices=["front", "chip", "lane"],
                    help="partition plan: where the edited latent's VAE decode + vocoder run -- 'lane' (default since round 5): on the "
                         "edit lane that edited the clip (the lanes have the slack: 1.3 s of work per 1.66 s); 'front': a codec stage "
                         "on the inversion partition's queue (round 4); 'chip': a codec stage on an unmasked queue of its own")
    ap.add_argument("--arith", default="bf16x6", choices=["f32", "bf16x6"],
                    help="arithmetic of the U-Net engines' LDS-staged GEMMs: bf16x6 (default, the product since round 4) = every "
                         "fp32 operand cut exactly into three bf16 pieces in the loader, six piece products on the bf16 MFMAs, "
                         "fp32 accumulation (csrc/conv_gemm_x6.hip; as close to fp64 as the fp32 chain); f32 = fp32-input MFMAs "
                         "everywhere (the round-1..3 arithmetic, kept for A/B)")
    ap.add_argument("--codec-arith", default=None, choices=["f32", "bf16x6"],
                    help="arithmetic of the codec engines' LDS-staged GEMMs (STFT-as-DFT, VAE, vocoder); default: the wrapper's "
                         "`codec_arith`")
    ap.add_argument("--lane-launch", default="graph", choices=["eager", "graph"],
                    help="how a pipeline worker issues one diffusion step: one hipGraphLaunch (default) or launch by launch")
    ap.add_argument("--no-overlap-prep", action="store_true",
                    help="partition plan: prepare the next clip on the front lane itself instead of a side stream (A/B)")
    ap.add_argument("--serial-clips", type=int, default=3,
                    help="clips timed in each of the two one-clip-at-a-time legs reported beside the headline")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the reported-only legs: parity vs the oracle, BASELINE configs 3 / 4 / 5 sub-benchmarks")
    ap.add_argument("--cpu-anchor-steps", type=int, default=50,
                    help="DDIM steps of the config-1 CPU anchor clip measured end to end through the oracle (0 = skip; BASELINE "
                         "configs[0] is 50 steps: the un-extrapolated anchor SURVEY 8(d) asks for, ~50-110 s of CPU on the box's "
                         "host; round 4 cut it to 12 steps and extrapolated)")
    ap.add_argument("--unmasked-prep", action="store_true",
                    help="A/B: the next clip's set-up on an UNMASKED side stream (rounds 3-5) instead of one masked to the inversion "
                         "partition's CUs (pipeline.ClipPipeline `mask_prep`)")
    ap.add_argument("--no-share-in-edit-loop", action="store_true",
                    help="A/B: CFG row sharing in the inversion only (round 5's shipped configuration; EditEngine.SHARE_IN_EDIT_LOOP)")
    ap.add_argument("--no-share-cfg-rows", action="store_true",
                    help="A/B: every classifier-free-guidance row through the whole U-Net graph (rounds 1-4) instead of computing "
                         "the context-free head once per latent (EditEngine.SHARE_CFG_ROWS)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    from audioeditingcode_amd import configs, dist as adist, editing, models, weights
    editing.EditEngine.SHARE_CFG_ROWS = not args.no_share_cfg_rows
    editing.EditEngine.SHARE_IN_EDIT_LOOP = not args.no_share_in_edit_loop
    from audioeditingcode_amd.main_run import edit_clip
    from audioeditingcode_amd.utils import prepare_waveform, synthetic_clip

    rank, world, local = adist.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but {world} rank(s) are running (n_gpus must equal the ranks that ran)"
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    # N ranks share the host: cap torch's intra-op CPU threads per rank and pin each rank to its slice of the cores
    rank_resources = (adist.pin_rank_resources(local, adist.local_world_size(world)) if world > 1
                      else dict(torch_threads=torch.get_num_threads()))

    # ---- weights: rank 0 materialises them, everyone else receives them over RCCL
    fam = configs.get_family(args.model_id)
    shapes = dict(unet=weights.unet_param_shapes(fam["unet"]), vae=weights.vae_param_shapes(fam["vae"]),
                  vocoder=weights.vocoder_param_shapes(fam["vocoder"]))
    sds = None
    if rank == 0:
        sds = {k: weights.random_state_dict(shapes[k], seed=i) for i, k in enumerate(("unet", "vae", "vocoder"))}
    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()
    t_bcast, bcast_bytes = 0.0, 0
    if grouped:             # world > 1, or a one-rank RCCL group under torchrun (executes the same collectives)
        sds, t_bcast, bcast_bytes = adist.broadcast_family(sds, shapes, dev, on_device=True)
    m = models.load_model(args.model_id, dev, args.T, state_dicts=sds, allow_synthetic=True)   # no checkpoint exists offline
    m.arith = args.arith        # lane views of the pipeline copy it
    if args.codec_arith is not None:
        m.codec_arith = args.codec_arith

    # ---- synthetic inputs, resident in HBM before the timed region
    src, tgt, neg = ["a recording of a piano melody"], ["a recording of an electric guitar melody"], [""]
    fn = m.get_fn_STFT()

    def clip_wave(i):
        w = prepare_waveform(synthetic_clip(10.0, seed=1234 + i), 1024 * 160)
        return torch.clip(torch.from_numpy(w)[None], -1, 1).to(dev)

    def run_clip(wave, schedule):
        mel, _, _ = fn.mel_spectrogram(wave)                              # [1, 64, 1025]
        x0 = mel[0].T[:1024][None, None].contiguous()                    # [1,1,1024,64]
        return edit_clip(m, x0, src, tgt, neg, [3.0], [12.0], args.T, args.tstart, schedule=schedule,
                         timestep_group=args.group)

    NC = max(1, args.clips_per_gpu)

    def run_clips(waves, schedule):
        """NC clips as one U-Net batch per step (EditEngine.edit_latents); codec per clip."""
        from audioeditingcode_amd.ddm_inversion.inversion_utils import conditioning_from_text
        x0s, w0s = [], []
        for wave in waves:
            mel, _, _ = fn.mel_spectrogram(wave)
            x0s.append(mel[0].T[:1024][None, None].contiguous())
            w0s.append(m.vae_encode(x0s[-1]))
        w0 = torch.cat(w0s, 0)
        ed_ = m.editor(w0.shape[-2], w0.shape[-1])
        c = {k: conditioning_from_text(m, m.encode_text(p, negative=neg_flag))
             for k, p, neg_flag in (("src", src, False), ("tgt", tgt, False), ("unc", [""], True), ("neg", neg, True))}
        w = ed_.edit_latents(w0, c["src"], c["unc"], c["tgt"], c["neg"], [3.0], [12.0], args.tstart, schedule=schedule,
                             group=max(1, args.group // NC))
        for i in range(len(waves)):
            x0_dec = m.vae_decode(w[i:i + 1])
            m.decode_to_mel(x0_dec if x0_dec.dim() == 4 else x0_dec[None])
            m.decode_to_mel(x0s[i])
        return w

    if args.profile_forward:
        torch.manual_seed(0)
        for _ in range(max(1, args.warmup)):
            run_clip(clip_wave(0), args.schedule)
        torch.cuda.synchronize()
        return

    def wave_to_mel(view, wave):
        """waveform -> mel on the lane's OWN STFT engine (engines hold their activations: never shared between lanes)"""
        mel, _, _ = view.get_fn_STFT().mel_spectrogram(wave)
        return mel[0].T[:1024][None, None].contiguous()

    dt_local = 0.0

    def finish(t0, lat):
        """Close the timed region: gather, sync, barrier, max over ranks; refuse NaN / all-zero results."""
        local_lat = torch.cat(lat, 0)
        gathered = adist.gather_to_rank0(local_lat)
        torch.cuda.synchronize()
        nonlocal dt_local
        dt_local = time.perf_counter() - t0         # this rank's own time (before the barrier): `per_rank_clips_per_s`
        adist.barrier()
        dt = adist.max_over_ranks(time.perf_counter() - t0, dev)
        if gathered is not None:            # a fast clip of NaNs is not a result
            for r, gl in enumerate(gathered):
                assert torch.isfinite(gl).all(), f"non-finite edited latents from rank {r}"
                assert gl.abs().max() > 0, f"all-zero edited latents from rank {r}"
        return dt, gathered

    def timed(schedule, K, W):
        """K clips (or K groups of NC clips), one after the other on the whole GPU."""
        def step(seed, waves_):
            torch.manual_seed(seed)
            if NC == 1:
                return run_clip(waves_[0], schedule)[2]
            return run_clips(waves_, schedule)
        for i in range(W):
            step(1000 + i, [clip_wave(rank * 100000 + i * NC + j) for j in range(NC)])
        waves = [[clip_wave(rank * 100000 + 5000 + i * NC + j) for j in range(NC)] for i in range(K)]
        lat = []
        torch.cuda.synchronize()
        adist.barrier()
        t0 = time.perf_counter()
        for i in range(K):
            # the x_t noise of clip i+1 (40 ms of single-threaded CPU RNG) is drawn on a host thread while clip i runs on
            # the GPU -- same generator, seed and draw order (PipelineWrapper.prefetch_noise); clip 0 draws in line, so
            # all K draws happen inside the timed region
            if NC == 1 and hasattr(m, "prefetch_noise"):
                m.next_noise_seed = 2000 + i + 1 if i + 1 < K else None
            lat.append(step(2000 + i, waves[i]))
        return finish(t0, lat)

    PLAN = args.plan if NC == 1 else "serial"
    pipe = None
    extra = {}

    def timed_pipeline(K, W):
        """K clips through pipeline.ClipPipeline.  Same waveforms and per-clip seeds as `timed`, so the serial legs below
        edit the same clips."""
        edit_args = (src, tgt, neg, [3.0], [12.0], args.T, args.tstart)
        pipe.warm_up(clip_wave(rank * 100000 + 99), *edit_args, prepare=wave_to_mel, seeds=[999])   # builds the workers
        if W:
            pipe.edit_clips([clip_wave(rank * 100000 + i) for i in range(W)], *edit_args, prepare=wave_to_mel,
                            seeds=[1000 + i for i in range(W)])
        waves = [clip_wave(rank * 100000 + 5000 + i) for i in range(K)]
        torch.cuda.synchronize()
        adist.barrier()
        t0 = time.perf_counter()
        res = pipe.edit_clips(waves, *edit_args, prepare=wave_to_mel, seeds=[2000 + i for i in range(K)])
        return finish(t0, [r[2] for r in res])

    log(f"model ready ({m.weights_source}); timing {args.steps} clip(s) after {args.warmup} warm-up")
    if PLAN != "serial":
        from audioeditingcode_amd.pipeline import ClipPipeline
        try:
            pipe = ClipPipeline(m, plan=PLAN, edit_cus=args.edit_cus, edit_lanes=args.edit_lanes, launch=args.lane_launch,
                                timestep_group=args.group, overlap_prep=not args.no_overlap_prep, codec_queue=args.codec_queue,
                                mask_prep=not args.unmasked_prep)
            dt, gathered = timed_pipeline(args.steps, args.warmup)
        except Exception as e:                                  # noqa: BLE001
            # a driver / container without CU-masked streams (hipExtStreamCreateWithCUMask, HSA_CU_MASK set, <= edit_cus CUs)
            # must still produce a headline: fall back to one clip at a time on the whole GPU
            log(f"clip pipeline unavailable ({e!r}): falling back to --plan serial")
            extra["pipeline_fallback"] = repr(e)
            pipe, PLAN = None, "serial"
    if PLAN != "serial":
        extra["pipeline"] = pipe.report()
        lanes_txt = (f"{pipe.edit_lanes} edit loops on disjoint {pipe.edit_lane_cus}-CU lanes" if pipe.edit_lanes > 1 and
                     pipe.edit_lane_cus != pipe.edit_cus else f"{pipe.edit_lanes} edit loop(s) on {pipe.edit_cus} CUs")
        headline = (f"a STREAM of {args.steps} clips per GPU (throughput of the stream, not the latency of one clip: "
                    f"`value_single_clip_batched` is the clip alone), "
                    f"up to {pipe.clips_in_flight} clips in flight, each alone in its U-Net batches: "
                    + f"forward inversion ({args.group} timesteps per U-Net call) on {pipe.total - pipe.edit_cus} CUs beside {lanes_txt}"
                    + ("; VAE decode + vocoder as a third stage on the inversion partition's queue, between its inversions"
                       if getattr(pipe, "codec_stage", False) else
                       ("; VAE decode + vocoder on the edit lane that edited the clip, the next clip's set-up on a side stream"
                        + (" masked to the inversion partition" if getattr(pipe, "mask_prep", False) else "")
                        if getattr(pipe, "codec_queue", "") == "lane" else "")))
        log(f"{PLAN} pipeline: {dt / args.steps:.3f} s/clip  {json.dumps(extra['pipeline'])}")
    else:
        dt, gathered = timed(args.schedule, args.steps, args.warmup)
        headline = (f"one clip at a time; forward inversion schedule: {args.schedule}"
                    + (f" ({args.group} timesteps per U-Net call)" if args.schedule == "batched" else " (reference order)"))
        log(f"{args.schedule}: {dt / args.steps:.3f} s/clip")
    value = world * NC * args.steps / dt
    # per-rank rates (the driver computes scaling efficiency from `value`; these show whether one rank lags)
    per_rank = adist.per_rank_rates(NC * args.steps, dt_local, dev) if grouped else None

    # ---- the same clips ONE AT A TIME (same waveforms, same seeds): reference order, and the timestep-batched inversion.
    # Every one of the 600 sample-forwards of a clip is computed in all three schedules; batched only regroups the
    # inversion's U-Net calls (the edit-friendly inversion draws all x_t independently from x_0, models.py:67-83).
    if not args.no_batched:
        n_ser = max(1, min(args.serial_clips, args.steps)) if PLAN != "serial" else args.steps
        legs = {}
        for sched_name in ("sequential", "batched"):
            if PLAN == "serial" and sched_name == args.schedule:
                legs[sched_name] = (dt, gathered)
                continue
            legs[sched_name] = timed(sched_name, n_ser, 1)
            log(f"one clip at a time, {sched_name}: {legs[sched_name][0] / n_ser:.3f} s/clip ({n_ser} clips)")
        n_of = {k: (args.steps if (PLAN == "serial" and k == args.schedule) else n_ser) for k in legs}
        extra["value_reference_order"] = world * NC * n_of["sequential"] / legs["sequential"][0]
        extra["ms_per_step_reference_order"] = 1e3 * legs["sequential"][0] / n_of["sequential"]
        extra["value_single_clip_batched"] = world * NC * n_of["batched"] / legs["batched"][0]
        extra["ms_per_step_single_clip_batched"] = 1e3 * legs["batched"][0] / n_of["batched"]
        extra["serial_legs_clips"] = n_ser
        if rank == 0 and legs["sequential"][1] is not None and legs["batched"][1] is not None:
            rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))   # noqa: E731
            n_cmp = min(n_of.values()) * NC
            dev_l2 = max(rel(a[:n_cmp], b[:n_cmp]) for a, b in zip(legs["batched"][1], legs["sequential"][1]))
            extra["schedule_deviation_rel_l2"] = dev_l2
            log(f"edited latents, timestep-batched vs reference order: rel L2 {dev_l2:.2e}")
            # the batched inversion lets x_t enter the U-Net before its ~1-ulp numerical fix: anything beyond the loop
            # tolerance of the GPU parity tests (5e-3) means the regrouping broke the arithmetic
            assert dev_l2 < 5e-3, f"timestep-batched inversion deviates from the reference order by {dev_l2:.3e}"
            if PLAN != "serial":
                # The pipeline only changes WHERE and WHEN a clip's launches run.  (1) Each clip must equal the same clip
                # pushed through the SAME engines alone (one clip per call: fill and drain on the whole chip, nothing
                # concurrent), bit for bit.  (2) Against the plain one-clip-at-a-time leg with the same inversion schedule
                # the only difference is the edit partition's tile choices (tile_table_cus128.py: other summation orders).
                twin = "batched" if PLAN == "partition" else "sequential"
                n_cmp = min(n_of[twin], args.steps)
                edit_args = (src, tgt, neg, [3.0], [12.0], args.T, args.tstart)
                alone = [pipe.edit_clips([clip_wave(rank * 100000 + 5000 + i)], *edit_args, prepare=wave_to_mel,
                                         seeds=[2000 + i])[0][2] for i in range(n_cmp)]
                alone = torch.cat(alone, 0).to(gathered[0].device)
                same = torch.equal(gathered[0][:n_cmp], alone)
                worst = float((gathered[0][:n_cmp] - alone).abs().max())
                vs_plain = rel(gathered[0][:n_cmp], legs[twin][1][0][:n_cmp])
                extra["pipeline_vs_one_clip_at_a_time"] = dict(
                    clips_compared=n_cmp, bit_identical_to_same_engines_alone=same, max_abs_diff=worst,
                    rel_l2_vs_plain_serial_leg=vs_plain, plain_serial_schedule=twin)
                log(f"pipeline vs the same clips alone through the same engines, {n_cmp} clips: bit-identical={same}, "
                    f"max |diff| {worst:.2e}; vs the plain {twin} leg: rel L2 {vs_plain:.2e}")
                assert same, f"clip results changed under the clip pipeline (max |diff| {worst:.3e})"
                assert vs_plain < 5e-3, f"pipeline deviates from the plain serial leg by {vs_plain:.3e}"

    # ---- where one clip's time goes when it has the GPU to itself (single-clip latency mode: batched inversion)
    phases = None
    if rank == 0 and NC == 1 and not args.no_batched:
        try:                    # reported-only leg: the headline line must survive a failure here
            pa = argparse.Namespace(**vars(args))
            pa.schedule = "batched"
            phases = clip_phases(m, fn, clip_wave(777), src, tgt, neg, pa)
            log(f"phases of one clip alone on the GPU [ms]: {phases}")
        except Exception as e:
            log(f"phase timing failed: {e!r}")

    # ---- roofline of the dominant kernel family (conv_gemm / lin_gemm, fp32 MFMA).  Durations are measured live with HIP
    # events on the streams the kernels run on: the captured U-Net forward graph of the headline's batch shape is replayed
    # n times on EVERY lane stream at once between one event pair per lane -> chip time per forward = lane time / L (what
    # the pipeline pays per U-Net call, no per-launch event overhead); one eager pass with an event pair per op gives each
    # op's share of a forward.  `achieved` prices EXECUTED flops (2MNK of every launch; the folded cross-attention executes
    # fewer than the reference formulation it replaces) -- the algorithmic count is used for the path-level figures only.
    roof = None
    if rank == 0:
        roof = roofline_leg(m, pipe, args, NC, dt)
        if roof is not None:
            try:
                check_fractions(roof)
            except AssertionError as e:          # never print a fraction that fails its own sanity check
                log(f"ROOFLINE CHECK FAILED: {e}")
                roof = dict(failed=str(e), bound=roof.get("bound"), peak=roof.get("peak"), unit=roof.get("unit"),
                            achieved=None, frac=None, traffic=None)
    parity = None
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            parity = parity_leg(m, fn, clip_wave(4242), src, tgt, neg)
            log(f"parity vs the CPU oracle: {parity}")
            try:                        # reported only: never fatal
                parity["parity_T200"] = parity_fixture_leg(m, src, tgt, neg, args.T, args.tstart)
                log(f"parity at the benched schedule vs the oracle fixture: {parity['parity_T200']}")
            except Exception as e:      # noqa: BLE001
                parity["parity_T200"] = dict(failed=repr(e))
        except AssertionError:
            raise
        except Exception as e:
            log(f"parity leg failed: {e!r}")
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # many-core hosts (the MI355X box has 256) make torch's CPU kernels slower, not faster, on these
        # small tensors: use at most 32 threads and report that number as `cores`
        try:
            base = cpu_baseline(fam, m.state_dicts, min(os.cpu_count() or 1, 32), args.cpu_anchor_steps)
        except Exception as e:          # the headline line must survive a failure of the reported-only baseline leg
            log(f"cpu_baseline failed: {e!r}")
    subs = None
    if rank == 0 and world == 1 and not args.no_extras:
        # BASELINE configs 3 (per-rank shape) / 4 / 5 as separate processes on this GPU: free this process's engines first
        pipe = None
        m._editors.clear()
        m._engines.clear()
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        subs = sub_benchmarks(time.time() - T_START)

    pipe_info = extra.get("pipeline")
    if rank == 0:
        out = {"metric": "edited-clips/sec (200-step inv+edit, 10 s@16 kHz)", "value": value,
               "unit": "edited-clips/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"AudioLDM2 ({args.model_id}, 346.9M-param U-Net, seeded-random weights) "
                                      f"text-based edit, T={args.T}, tstart={args.tstart}, cfg 3/12, {NC} clip(s) of 10 s "
                                      f"@16 kHz per U-Net batch; {headline}",
                          "clips_per_gpu_per_step": NC,
                          "clips_in_flight_per_gpu": NC if pipe_info is None else pipe_info["clips_in_flight"],
                          "parallelism": f"clip-dp{world}" + ("" if pipe_info is None else f" x {PLAN} pipeline"),
                          "arith": ARITH_TEXT[args.arith], "cfg_row_sharing": not args.no_share_cfg_rows,
                          "cfg_row_sharing_in_edit_loop": not (args.no_share_cfg_rows or args.no_share_in_edit_loop),
                          "codec_arith": getattr(m, "codec_arith", "f32"),
                          **adist.distributed_fields(world, t_bcast if grouped else 0.0, bcast_bytes, per_rank, rank_resources),
                          "process_group": (torch.distributed.get_backend() if grouped else None),
                          "gathered_latents": None if gathered is None else [list(g.shape) for g in gathered][:2]},
               "roofline": roof, "cpu_baseline": base, "parity": parity, "phases_ms_one_clip_alone": phases}
        out.update(extra)
        if subs:
            out.update(subs)
        print(json.dumps(out), flush=True)
    if grouped:
        # leave the process group cleanly (RCCL warns about leaked resources otherwise); never fatal after the line is out
        try:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        except Exception as e:          # noqa: BLE001
            log(f"process group teardown: {e!r}")


# ---------------------------------------------------------------------------------------------------------------------------
# Roofline (round 5: PHYSICAL -- every fraction is executed work of one kernel family over the peak of the instruction that
# family issues, 0 < frac <= 1 with no exception; the fp32-equivalent figures keep their own names).
#
#   family                 instruction                          peak (MI355X_MICROARCH.md)    executed flops per launch
#   gemm_bf16x6            v_mfma_f32_32x32x16_bf16             2500 TFLOP/s dense bf16       6 x 2MNK (six piece products)
#   attention_bf16x6       v_mfma_f32_32x32x16_bf16             2500                          6 x 4 B H Nq Nk D
#   gemm_f32 / attention_f32   v_mfma_f32_32x32x2_f32           157.3 TFLOP/s fp32-in MFMA    2MNK / 4 B H Nq Nk D
#   groupnorm, elementwise     streaming                        8 TB/s HBM (6.3 achievable)   algorithmic bytes of the op record
#
# Durations: HIP events on the stream the kernels run on, one event pair per op record (`aed_tape_profile`), the forward's ops
# launched one at a time -- what `rocprofv3 --kernel-trace --stats` reports per kernel for the same forward (profiles/r05_*).
# A family measured on a CU partition is priced against cus/256 of the chip peak.
def op_family(op, meta):
    """(family, instruction peak in TFLOP/s or None for the streaming ops, executed-flop multiplier) of one op record."""
    code = meta["code"]
    if code == 1:                                   # AED_OP_CONV_GEMM
        if op.flags & 64:
            return "gemm_mxfp8", 5000.0, 1.0
        if (op.flags & 4) and op.i[29] < 10:
            return "gemm_bf16x6", PEAK_BF16_MFMA_TFLOPS, 6.0
        return "gemm_f32", PEAK_FP32_MFMA_TFLOPS, 1.0
    if code == 5:                                   # AED_OP_ATTENTION
        if (op.flags & 4) and op.i[14] == 3:
            return "attention_bf16x6", PEAK_BF16_MFMA_TFLOPS, 6.0
        return "attention_f32", PEAK_FP32_MFMA_TFLOPS, 1.0
    if code in (2, 3, 21, 22):
        return "groupnorm", None, 0.0
    return "elementwise", None, 0.0


def profile_forward(eng, stream, reps=2):
    """Per-op milliseconds of one U-Net forward on `stream` (event pair per op, min over `reps` passes after a warm one)."""
    with torch.cuda.stream(stream):
        eng.tape.profile()
        runs = [eng.tape.profile() for _ in range(reps)]
    return [min(r[k] for r in runs) for k in range(len(runs[0]))]


def family_table(eng, ms, cu_frac=1.0):
    """Per family: launches, total ms, executed flops of the issued instruction, and the fraction of that instruction's peak on
    the CUs the forward ran on."""
    fam = {}
    for op, mt, t in zip(eng.tape.ops, eng.tape.meta, ms):
        name, peak, mult = op_family(op, mt)
        f = fam.setdefault(name, dict(launches=0, ms=0.0, flops_fp32_equiv=0.0, flops_executed=0.0, bytes=0.0, peak=peak))
        f["launches"] += 1
        f["ms"] += t
        f["flops_fp32_equiv"] += mt["exec_flops"]
        f["flops_executed"] += mult * mt["exec_flops"]
        f["bytes"] += mt.get("bytes", 0)
    out = {}
    for name, f in fam.items():
        d = dict(launches=f["launches"], ms=f["ms"], avg_launch_us=1e3 * f["ms"] / max(1, f["launches"]))
        if f["peak"] is not None and f["ms"] > 0:
            ach = f["flops_executed"] / (f["ms"] * 1e-3) / 1e12
            d.update(instruction_peak_tflops=f["peak"] * cu_frac, achieved_tflops=ach, frac=ach / (f["peak"] * cu_frac),
                     matrix_pipe_seconds_at_chip_peak=f["flops_executed"] / (f["peak"] * 1e12),
                     achieved_tflops_fp32_equiv=f["flops_fp32_equiv"] / (f["ms"] * 1e-3) / 1e12)
        elif f["ms"] > 0 and f["bytes"] > 0:
            gbs = f["bytes"] / (f["ms"] * 1e-3) / 1e9
            # HBM is not partitioned by a CU mask: a partition alone may pull the whole chip's bandwidth.  A producer ->
            # consumer pair whose tensor fits the 256 MB Infinity Cache can move its algorithmic bytes faster than HBM could:
            # then there is no HBM fraction to state
            d.update(algorithmic_gb_per_s=gbs, hbm_peak_gb_per_s=PEAK_HBM_GBS)
            if gbs <= PEAK_HBM_GBS:
                d["frac"] = gbs / PEAK_HBM_GBS
            else:
                d["served_from_cache"] = True
        out[name] = d
    return out


def check_fractions(roof):
    """Every `frac` of the roofline object is executed work over the peak of the instruction that did it: it must lie in
    (0, 1].  Raises AssertionError naming the offender (round 3 printed path_frac = 4.23, round 4 let `frac` exceed 1 under
    bf16x6: neither can pass here).  No other key of the object is named like a fraction."""
    def walk(d, where):
        for k, v in d.items():
            if isinstance(v, dict):
                walk(v, f"{where}.{k}")
            elif (k == "frac" or k.endswith("_frac") or k.startswith("frac_")) and v is not None:
                assert 0.0 < v <= 1.0, f"{where}.{k} = {v:.3f} is outside (0, 1]: the accounting is wrong"
    walk(roof, "roofline")


def _find_engine(views, B):
    """The U-Net engine of batch B that this run used, among the given model views (engine keys: (B, L0, L1[, arith]))."""
    for v in views:
        ed = v.editor(256, 16)
        cand = sorted(((len(key), e) for key, e in ed._unets.items() if key[0] == B), key=lambda t: -t[0])
        if cand:
            with torch.inference_mode():
                for pl in ed._plans.values():
                    pl["state"].zero_()        # the time-embedding op indexes the timestep table with the loop counter
            return cand[0][1]
    return None


def graph_ms(eng, stream, n):
    """Milliseconds per forward as a captured hipGraph replayed n times on `stream` (what a loop step pays)."""
    with torch.cuda.stream(stream):
        eng.tape.capture()
        for _ in range(2):
            eng.tape.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            eng.tape.replay()
        e1.record(stream)
        e1.synchronize()
    return e0.elapsed_time(e1) / n


def roofline_leg(m, pipe, args, NC, dt):
    """The `roofline` object of the JSON line.  Dominant kernel: the split-bf16 implicit-GEMM (conv_gemm_x6_kernel) of the
    inversion's timestep-batched U-Net forward -- 400 of a clip's 600 sample-forwards and ~80 % of that forward's time.
    Measured (1) on the whole chip, one launch at a time (the figure a `rocprofv3 --kernel-trace --stats` of
    `bench.py --plan serial` reproduces: profiles/r05_kernel_trace_serial.md) -> `achieved` / `frac`; (2) on the CU partition
    the pipeline's front stage runs it on, against that partition's share of the peak -> `on_partition`; the edit loop's
    forward (U-Net batch 2g on the edit lane) per family -> `edit_step`."""
    dev = m.device
    G = max(1, min(args.group // NC if NC > 1 else args.group, args.T))
    while args.T % G:
        G -= 1
    sequential_headline = pipe is None and args.schedule == "sequential"
    B_inv = 2 * NC if sequential_headline else 2 * G * NC
    views = [m] + ([w.view for w in pipe.workers] if pipe is not None else [])
    whole = torch.cuda.Stream(device=dev)
    out = dict(bound="mfma", unit="TFLOP/s", peak=PEAK_BF16_MFMA_TFLOPS, csrc_hash=csrc_hash(),
               peak_note="dense bf16 MFMA peak of the MI355X (MI355X_MICROARCH.md: ~2.5 PFLOP/s; measured ceiling 2495); the kernel "
                         "issues v_mfma_f32_32x32x16_bf16, six piece products per fp32-equivalent product",
               kernel="conv_gemm_x6_kernel<*> (csrc/conv_gemm_x6.hip): every LDS-staged conv / Linear of the U-Net forward at "
                      f"batch {B_inv}",
               method="executed MFMA flops (6 x 2MNK per launch) / sum of launch durations; durations = HIP event pairs around "
                      "every op record of the forward on the stream it runs on (aed_tape_profile), ops launched one at a time")
    eng = _find_engine(views, B_inv)
    if eng is None:
        return None
    torch.cuda.synchronize()
    ms = profile_forward(eng, whole)
    fams = family_table(eng, ms, 1.0)
    dom = fams.get("gemm_bf16x6") or fams.get("gemm_f32")
    if dom is None or "achieved_tflops" not in dom:
        return None
    if "gemm_bf16x6" not in fams:                   # --arith f32: the dominant family issues fp32-input MFMAs
        out.update(peak=PEAK_FP32_MFMA_TFLOPS, peak_note="fp32-input MFMA peak (v_mfma_f32_32x32x2_f32)",
                   kernel=out["kernel"].replace("conv_gemm_x6_kernel<*> (csrc/conv_gemm_x6.hip)", "conv_gemm_kernel / lin_gemm_kernel"))
    n_x6 = dom["launches"]
    alg = sum(mt["flops"] for mt in eng.tape.meta)
    out.update(achieved=dom["achieved_tflops"], frac=dom["achieved_tflops"] / out["peak"],
               achieved_fp32_equiv=dom["achieved_tflops_fp32_equiv"],
               fp32_equiv_over_fp32_mfma_peak=dom["achieved_tflops_fp32_equiv"] / PEAK_FP32_MFMA_TFLOPS,
               fp32_equiv_note="`*_fp32_equiv` = 2MNK per launch over the same durations, against the fp32-input MFMA peak (157.3): "
                               "the arithmetic the results are equivalent to.  `fp32_equiv_over_fp32_mfma_peak` compares the path with an ideal "
                               "fp32-MFMA implementation: a ratio, not a roofline fraction (it exceeds 1 when split-bf16 beats that ideal)",
               launches_per_forward=n_x6, avg_launch_us=dom["avg_launch_us"],
               algorithmic_gflop_per_launch=dom["achieved_tflops_fp32_equiv"] * dom["avg_launch_us"] * 1e-3,
               forward=dict(unet_batch=B_inv, where="whole chip, launches one at a time", ms_sum_of_launches=sum(ms),
                            ms_as_graph=graph_ms(eng, whole, 3), algorithmic_gflop=alg / 1e9, families=fams))
    # ---- the same forward on the pipeline's front partition (alone on it)
    if pipe is not None and pipe.plan == "partition":
        front = [w for w in pipe.workers if w.stage == "front"][0]
        cu_frac = (pipe.total - pipe.edit_cus) / pipe.total
        eng_f = _find_engine([front.view], B_inv) or eng
        torch.cuda.synchronize()
        ms_f = profile_forward(eng_f, front.lane.stream)
        fam_f = family_table(eng_f, ms_f, cu_frac)
        d = fam_f.get("gemm_bf16x6") or fam_f.get("gemm_f32") or {}
        out["on_partition"] = dict(cus=pipe.total - pipe.edit_cus, cu_fraction_of_chip=cu_frac,
                                   peak=out["peak"] * cu_frac, achieved=d.get("achieved_tflops"), frac=d.get("frac"),
                                   achieved_fp32_equiv=d.get("achieved_tflops_fp32_equiv"), ms_sum_of_launches=sum(ms_f),
                                   ms_as_graph=graph_ms(eng_f, front.lane.stream, 3), families=fam_f,
                                   note="the front stage's forward alone on its CU partition (nothing on the other CUs): the "
                                        "half-loaded chip holds a higher clock than the whole chip under the bf16 MFMA stream")
        # ---- the edit loop's step on the edit lane
        back = [w for w in pipe.workers if w.stage == "back"]
        steps = {}
        for g in (1,):
            e = _find_engine([back[0].view], 2 * g * NC)
            if e is None:
                continue
            lane_frac = pipe.edit_lane_cus / pipe.total
            torch.cuda.synchronize()
            ms_e = profile_forward(e, back[0].lane.stream)
            gm = graph_ms(e, back[0].lane.stream, 20 if g <= 2 else 6)
            steps[f"clips_{g}"] = dict(unet_batch=2 * g * NC, lane_cus=pipe.edit_lane_cus, ms_per_step_as_graph=gm,
                                       ms_per_clip_step=gm / g, launches=len(e.tape.ops),
                                       tflops_fp32_equiv_algorithmic=e.tape.flops / (gm * 1e-3) / 1e12,
                                       families=family_table(e, ms_e, lane_frac))
        out["edit_step"] = steps
    # ---- HBM-side traffic of the dominant family from the committed PMC passes (separate runs; only if taken on this source tree)
    traffic = traffic_note = None
    for name in ("r06_pmc_forward.json", "r05_pmc_forward.json", "r04_pmc_forward.json"):
        pmc_path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(pmc_path):
            continue
        try:
            with open(pmc_path) as f:
                pmc = json.load(f)
            if pmc.get("csrc_hash") == csrc_hash():
                traffic = pmc["measured_bytes_per_launch"]
                traffic_note = (f"bytes per conv_gemm-family launch, {pmc['counters']}; algorithmic "
                                f"{pmc['algorithmic_bytes_per_launch']:.3g} B per launch; {pmc['source']}")
            else:
                traffic_note = f"null: profiles/{name} was measured on csrc {pmc.get('csrc_hash')}, this binary is {csrc_hash()}"
            break
        except (OSError, KeyError, ValueError) as e:
            log(f"PMC summary unreadable: {e!r}")
    out.update(traffic=traffic, traffic_note=traffic_note)
    # ---- path level: one clip's U-Net work against the wall clock of the headline
    s_clip = dt / args.steps / NC
    n_inv = (args.T if sequential_headline else args.T // G)
    per_clip_alg = (n_inv * alg + args.tstart * alg * (2 * NC) / B_inv) / NC      # inversion forwards + edit forwards (same graph, batch 2)
    # matrix-pipe time a clip NEEDS at the instruction peaks (x6 families at 2500, fp32 families at 157.3) over the wall clock
    pipe_s = lambda table: sum(f.get("matrix_pipe_seconds_at_chip_peak", 0.0) for f in table.values())     # noqa: E731
    need_inv = pipe_s(fams)
    edit1 = (out.get("edit_step") or {}).get("clips_1")
    # the edit loop's forward has its own family mix (latency-regime fp32 kernels at batch 2); without a measured one (serial
    # plans) the inversion forward's mix is scaled by the batch ratio
    need_edit = pipe_s(edit1["families"]) if edit1 else need_inv * (2 * NC) / B_inv
    need_clip = (n_inv * need_inv + args.tstart * need_edit) / NC
    out["path"] = dict(clip_unet_tflop_algorithmic=per_clip_alg / 1e12, seconds_per_clip=s_clip,
                       tflops_fp32_equiv=per_clip_alg / s_clip / 1e12,
                       fp32_equiv_over_fp32_mfma_peak=per_clip_alg / s_clip / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                       matrix_pipe_seconds_needed_per_clip=need_clip, matrix_pipe_frac=need_clip / s_clip,
                       note="matrix_pipe_frac = [executed MFMA flops of a clip's 600 sample-forwards, each family over the peak of "
                            "its own instruction] / wall seconds per clip: the share of the chip's matrix-pipe time the headline uses")
    return out


def parity_leg(m, fn, wave, src, tgt, neg, T=8, tstart=4):
    """BASELINE's metric is clips/s + mel-L2 vs ref: the benched model (full-size AudioLDM2, same weights) edits one
    10 s clip at a SHORT schedule (T=8, tstart=4: the CPU oracle needs ~1 s per U-Net forward at this size) through the
    same wrapper path, and the CPU oracle (reference restatement, oracle/) edits it with the same conditioning and the
    same x_t noise.  Relative L2 of the edited latent, the decoded mel and the waveform."""
    from audioeditingcode_amd.main_run import edit_clip
    from oracle import hifigan as ohifi, loops as oloops, unet as ounet, vae as ovae
    from oracle.scheduler import OracleDDIMScheduler
    sched = m.model.scheduler
    T_keep = sched.num_inference_steps
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))      # noqa: E731
    t0 = time.time()
    try:
        sched.set_timesteps(T)
        mel, _, _ = fn.mel_spectrogram(wave)
        x0 = mel[0].T[:1024][None, None].contiguous()
        m.next_noise_seed = None
        torch.manual_seed(77)
        audio, _, w_edit = edit_clip(m, x0, src, tgt, neg, [3.0], [12.0], T, tstart)
        with torch.inference_mode():
            mel_dev = m.vae_decode(w_edit).cpu()
        torch.cuda.synchronize()
    finally:
        sched.set_timesteps(T_keep)
    cfg, sd = m.family["unet"], m.state_dicts["unet"]
    osched = OracleDDIMScheduler()
    osched.set_timesteps(T)

    def unet_fn(x, t, cond):
        hs, cl, mk = (v.cpu().expand(x.shape[0], *v.shape[1:]) for v in cond)
        return ounet.unet_forward(cfg, sd, x, t, encoder_hidden_states=hs, encoder_hidden_states_1=cl,
                                  encoder_attention_mask_1=mk)[0]
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    with torch.no_grad():
        ow = oloops.OracleWrapper(osched, unet_fn)
        w0 = ovae.vae_encode(m.family["vae"], m.state_dicts["vae"], x0.cpu())
        xts0 = ow.sample_xts_from_x0(w0, T, generator=torch.Generator().manual_seed(77))
        enc = lambda p, **k: tuple(None if t is None else t.cpu() for t in m.encode_text(p, **k))     # noqa: E731
        _, zs, xts = oloops.invert(ow, w0, enc(src), enc([""], negative=True), [3.0], T, xts=xts0)
        w_o = oloops.edit(ow, xts, torch.tensor([tstart]), enc(tgt), enc(neg, negative=True), [12.0], zs[:tstart],
                          eta=1.0)
        mel_o = ovae.vae_decode(m.family["vae"], m.state_dicts["vae"], w_o)
        wav_o = ohifi.hifigan_forward(m.family["vocoder"], m.state_dicts["vocoder"], mel_o[:, 0])
    out = dict(workload=f"benched AudioLDM2 weights, one 10 s clip, T={T}, tstart={tstart}, reference step order, HIP path "
                        f"vs CPU oracle on identical conditioning and x_t noise",
               latent_rel_l2=rel(w_edit.cpu(), w_o), mel_rel_l2=rel(mel_dev, mel_o), waveform_rel_l2=rel(audio, wav_o),
               tolerance=dict(latent=5e-3, mel=5e-3, waveform=2e-2), seconds=time.time() - t0)
    assert out["latent_rel_l2"] < 5e-3 and out["mel_rel_l2"] < 5e-3 and out["waveform_rel_l2"] < 2e-2, \
        f"HIP path deviates from the CPU oracle beyond the stated tolerance: {out}"
    return out


def parity_fixture_leg(m, src, tgt, neg, T, tstart):
    """The same comparison at the BENCHED schedule (T=200, tstart=100, reference step order): the oracle side is a committed
    fixture (oracle/make_bench_parity_golden.py: ~7 min of CPU, the bench's own weights / clip #4242 / prompts / seed), the HIP
    side edits the fixture's mel here.  Reported only (`parity_T200`); the asserted parity is the live T=8 leg."""
    import numpy as np
    from audioeditingcode_amd.main_run import edit_clip
    path = os.path.join(ROOT, "tests", "golden", "bench_parity_T200.npz")
    if not os.path.exists(path):
        return dict(skipped="tests/golden/bench_parity_T200.npz is missing (oracle/make_bench_parity_golden.py)")
    fx = np.load(path)
    if int(fx["T"]) != T or int(fx["tstart"]) != tstart or [str(p) for p in fx["prompts"]] != [src[0], tgt[0], neg[0]]:
        return dict(skipped=f"the fixture was made for T={int(fx['T'])}, tstart={int(fx['tstart'])} and other prompts")
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))      # noqa: E731
    t0 = time.time()
    x0 = torch.from_numpy(fx["x0"]).to(m.device)
    m.next_noise_seed = None
    torch.manual_seed(int(fx["seed"]))
    audio, _, w_edit = edit_clip(m, x0, src, tgt, neg, [3.0], [12.0], T, tstart)
    with torch.inference_mode():
        mel_dev = m.vae_decode(w_edit).cpu()
    torch.cuda.synchronize()
    return dict(workload=f"benched AudioLDM2 weights, clip #{int(fx['clip'])}, T={T}, tstart={tstart}, reference step order, HIP "
                         f"path vs the CPU oracle's run of the same schedule (committed fixture)",
                latent_rel_l2=rel(w_edit.cpu(), torch.from_numpy(fx["w_edit"])),
                mel_rel_l2=rel(mel_dev, torch.from_numpy(fx["mel"])),
                waveform_rel_l2=rel(audio.cpu(), torch.from_numpy(fx["wav"])), seconds=time.time() - t0)


def sub_benchmarks(elapsed_s):
    """BASELINE configs 3 (one rank's share: 8 clips per U-Net batch), 4 (PC extract + apply at full size) and 5 (Stable
    Audio Open, fp32) as bounded sub-processes on the same GPU, so the driver's single bench run sees them.  Each prints
    one JSON line; a failure or timeout is recorded, never fatal."""
    py, env = sys.executable, dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    jobs = [("config3_per_rank", [py, os.path.join(ROOT, "bench.py"), "--clips-per-gpu", "8", "--steps", "1", "--warmup",
                                  "1", "--no-extras", "--no-cpu-baseline", "--no-batched"], 240),
            ("config4_pc_extract_apply", [py, os.path.join(ROOT, "tools", "bench_config4.py")], 300),
            ("config5_stable_audio", [py, os.path.join(ROOT, "tools", "bench_stable_audio.py"), "--steps", "1",
                                           "--warmup", "1"], 300),
            ]
    # (round 5: the fp32-arithmetic A/B leg and the MX-FP8 experiment leg left the driver's run -- the full 50-step config-1 CPU
    # anchor took their time; `python bench.py --arith f32 ...` / `tools/bench_stable_audio.py --arith fp8` still run them by hand)
    start_by = {}
    out = {}
    for key, cmd, limit in jobs:
        if elapsed_s > start_by.get(key, 600):  # keep the whole default run bounded
            out[key] = dict(skipped=f"bench already ran {elapsed_s:.0f} s")
            continue
        t0 = time.time()
        try:
            r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=limit)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and line:
                d = json.loads(line[-1])
                keep = {k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "config", "checks",
                                          "phases_s_one_clip", "seconds", "pipeline", "pipeline_vs_one_clip_at_a_time", "parity_T200")
                        if k in d}
                if isinstance(d.get("roofline"), dict):
                    keep["roofline"] = {k: v for k, v in d["roofline"].items() if not isinstance(v, (dict, list))}
                out[key] = keep
            else:
                out[key] = dict(failed=f"rc={r.returncode}", stderr_tail=r.stderr[-400:])
        except subprocess.TimeoutExpired:
            out[key] = dict(failed=f"timeout after {limit} s")
        except Exception as e:                          # noqa: BLE001 -- reported-only legs
            out[key] = dict(failed=repr(e))
        dt = time.time() - t0
        elapsed_s += dt
        log(f"{key}: {json.dumps(out[key])[:300]} ({dt:.0f} s)")
    return out


if __name__ == "__main__":
    main()
