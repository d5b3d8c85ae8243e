#!/usr/bin/env python
"""bench.py -- edited-clips/sec of the DDPM-inversion audio-editing hot path on MI355X.

Workload (BASELINE.json configs[1]): AudioLDM2 text-based edit of ONE synthetic 10 s / 16 kHz clip per
step -- STFT/log-mel -> VAE encode -> 200-step edit-friendly DDPM inversion (cfg 3) -> 100-step edit from
tstart=100 (cfg 12) -> VAE decode -> HiFi-GAN vocoder (edited + original, main_run.py:184-185) -- fp32,
seeded-random weights of the real architecture (no checkpoints exist offline), synthetic conditioning.
One "step" = one whole clip.  N GPUs = N independent clips per step (weak scaling, no data-path
collective; weights broadcast once from rank 0 over RCCL, edited latents gathered to rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import glob
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


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def csrc_hash():
    """sha1 over the GEMM-family sources: ties a committed PMC traffic summary to the kernels being benched."""
    h = hashlib.sha1()
    for f in ("conv_gemm.hip", "lin_gemm.hip", "cg_params.h", "aed_common.h"):      # the kernel family `traffic` refers to
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
    return dict(workload=f"AudioLDM-S (185 M U-Net, seeded-random weights), --mode ddim, T={T}, one 10 s clip, "
                         f"{n_fwd[0]} U-Net sample-forwards + codec, oracle on {cores} host threads, measured end to end",
                seconds=dt, clips_per_sec=1.0 / dt, finite=bool(torch.isfinite(w_e).all()))


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
                         "-> 85 / 92 / 99 TFLOP/s per forward; 200 needs ~150 GB of activations, well inside 288 GB)")
    ap.add_argument("--schedule", default="batched", choices=["batched", "sequential"],
                    help="headline schedule of the forward inversion (the other one is timed as an extra)")
    ap.add_argument("--profile-forward", action="store_true", help="only run U-Net forwards (for rocprofv3)")
    ap.add_argument("--clips-per-gpu", type=int, default=1,
                    help="independent clips edited together per step and GPU (BASELINE configs[2]: 8; default: the "
                         "headline configs[1] shape, 1)")
    ap.add_argument("--cpu-anchor-steps", type=int, default=50,
                    help="DDIM steps of the un-extrapolated config-1 CPU anchor clip (0 = skip; BASELINE configs[0] is 50)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    from audioeditingcode_amd import configs, dist as adist, models, weights
    from audioeditingcode_amd.main_run import edit_clip
    from audioeditingcode_amd.utils import prepare_waveform, synthetic_clip

    rank, world, local = adist.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but {world} rank(s) are running (n_gpus must equal the ranks that ran)"
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    # ---- weights: rank 0 materialises them, everyone else receives them over RCCL
    fam = configs.get_family(args.model_id)
    shapes = dict(unet=weights.unet_param_shapes(fam["unet"]), vae=weights.vae_param_shapes(fam["vae"]),
                  vocoder=weights.vocoder_param_shapes(fam["vocoder"]))
    sds = None
    if rank == 0:
        sds = {k: weights.random_state_dict(shapes[k], seed=i) for i, k in enumerate(("unet", "vae", "vocoder"))}
    t0 = time.time()
    if world > 1:
        sds = {k: adist.broadcast_state_dict(None if sds is None else sds[k], shapes[k], dev, on_device=True)
               for k in shapes}
        torch.cuda.synchronize()
    t_bcast = time.time() - t0
    m = models.load_model(args.model_id, dev, args.T, state_dicts=sds, allow_synthetic=True)   # no checkpoint exists offline

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
        ed = m.editor(256, 16)
        torch.manual_seed(0)
        for _ in range(max(1, args.warmup)):
            run_clip(clip_wave(0), args.schedule)
        torch.cuda.synchronize()
        return

    def timed(schedule, K, W):
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
        local_lat = torch.cat(lat, 0)
        gathered = adist.gather_to_rank0(local_lat)
        torch.cuda.synchronize()
        adist.barrier()
        dt = adist.max_over_ranks(time.perf_counter() - t0, dev)
        if gathered is not None:            # a fast clip of NaNs is not a result
            for r, gl in enumerate(gathered):
                assert torch.isfinite(gl).all(), f"non-finite edited latents from rank {r} ({schedule} schedule)"
                assert gl.abs().max() > 0, f"all-zero edited latents from rank {r}"
        return dt, gathered

    # Headline schedule: timestep-batched forward inversion (G timesteps per U-Net call) + sequential edit.
    # Every one of the 600 sample-forwards of the clip is computed; only their grouping differs from the
    # reference's Python loop (the edit-friendly inversion draws all x_t independently from x_0,
    # models.py:67-83).  The reference's one-timestep-at-a-time order is timed too and reported beside it.
    log(f"model ready ({m.weights_source}); timing {args.steps} clip(s) after {args.warmup} warm-up")
    dt, gathered = timed(args.schedule, args.steps, args.warmup)
    value = world * NC * args.steps / dt
    log(f"{args.schedule}: {dt / args.steps:.3f} s/clip")
    extra = {}
    if not args.no_batched:
        other = "sequential" if args.schedule == "batched" else "batched"
        dto, gathered_o = timed(other, args.steps, 1)
        key = "reference_order" if other == "sequential" else "batched_inversion"
        extra[f"value_{key}"] = world * NC * args.steps / dto
        extra[f"ms_per_step_{key}"] = 1e3 * dto / args.steps
        log(f"{other}: {dto / args.steps:.3f} s/clip")
        # the two schedules edited the SAME clips with the SAME seeds: how far the timestep-batched inversion (x_t enters
        # the U-Net before its ~1-ulp numerical fix) moves the edited latents from the reference order -- reported, and
        # bounded by the GPU parity tests at the full length (rel < 5e-3)
        try:
            if gathered is not None and gathered_o is not None:
                extra["schedule_deviation_rel_l2"] = max(
                    float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))
                    for a, b in zip(gathered, gathered_o))
                log(f"edited latents, batched vs reference order: rel L2 {extra['schedule_deviation_rel_l2']:.2e}")
        except Exception as e:                      # reported-only leg: the headline line must survive a failure here
            log(f"schedule deviation not computed: {e!r}")

    # ---- roofline of the dominant kernel family (conv_gemm / lin_gemm, fp32 MFMA).  Durations are measured live, on the
    # stream the kernels run on, with HIP events: (1) the captured U-Net forward graph of each batch shape of the clip is
    # replayed n times between one event pair -> forward_ms (what the loops actually pay per U-Net call, no per-launch
    # event overhead); (2) one eager pass with an event pair per op gives every op's share; the family's time inside the
    # graph is forward_ms * (family share).  rocprofv3 --kernel-trace of the same command (profiles/) must agree.
    # ---- where one clip's time goes (one extra clip, not part of the metric, with a device sync after every phase)
    phases = None
    if rank == 0 and NC == 1:
        try:                    # reported-only leg: the headline line must survive a failure here
            phases = clip_phases(m, fn, clip_wave(777), src, tgt, neg, args)
            log(f"phases of one clip [ms]: {phases}")
        except Exception as e:
            log(f"phase timing failed: {e!r}")

    roof = None
    if rank == 0:
        ed = m.editor(256, 16)
        st = torch.cuda.Stream(device=dev)
        tot_fl = tot_ms = 0.0
        n_launch = 0
        detail = {}
        per_clip_flops = 0.0
        for (B, _, _), eng in ed._unets.items():
            calls = {2 * NC: args.tstart + (args.T if args.schedule == "sequential" else 0)}
            if args.schedule == "batched":
                G = max(1, min(args.group // NC if NC > 1 else args.group, args.T))
                while args.T % G:
                    G -= 1
                calls[2 * G * NC] = calls.get(2 * G * NC, 0) + args.T // G
            if B not in calls or not calls[B] or f"unet_batch_{B}" in detail:
                continue        # one engine per batch size (context lengths differ, launch shapes do not)
            with torch.inference_mode():
                ed.state.zero_()        # the time-embedding op indexes the timestep table with the loop counter
            torch.cuda.synchronize()
            with torch.cuda.stream(st):
                eng.tape.profile()
                ms = [eng.tape.profile() for _ in range(3)]
                ms = [sum(x) / len(ms) for x in zip(*ms)]
                eng.tape.capture()
                for _ in range(2):
                    eng.tape.replay()
                n_rep = 20 if B <= 8 else 5
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(st)
                for _ in range(n_rep):
                    eng.tape.replay()
                ev1.record(st)
                ev1.synchronize()
                fwd_ms = ev0.elapsed_time(ev1) / n_rep
            conv = [(mt["flops"], t) for mt, t in zip(eng.tape.meta, ms) if mt["code"] == 1]
            fl, share = sum(f for f, _ in conv), sum(t for _, t in conv) / sum(ms)
            tt = fwd_ms * share
            tot_fl += calls[B] * fl
            tot_ms += calls[B] * tt
            n_launch += calls[B] * len(conv)
            per_clip_flops += calls[B] * eng.tape.flops
            detail[f"unet_batch_{B}"] = dict(forwards_per_clip=calls[B], launches_per_forward=len(eng.tape.ops),
                                             conv_gemm_launches=len(conv), forward_ms=fwd_ms,
                                             forward_tflops=eng.tape.flops / (fwd_ms * 1e-3) / 1e12,
                                             conv_gemm_share_of_forward=share,
                                             conv_gemm_tflops=fl / (tt * 1e-3) / 1e12,
                                             eager_event_per_op_sum_ms=sum(ms),
                                             algorithmic_gflop=eng.tape.flops / 1e9)
        achieved = tot_fl / (tot_ms * 1e-3) / 1e12
        # HBM-side traffic per conv_gemm launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs; counter
        # passes serialise every dispatch and cannot run inside a timed bench) committed under profiles/, used only when
        # they were taken on THIS source tree (hash of csrc/); null otherwise
        traffic = traffic_note = None
        pmc_path = os.path.join(ROOT, "profiles", "r02_pmc_forward.json")
        if os.path.exists(pmc_path):
            try:
                with open(pmc_path) as f:
                    pmc = json.load(f)
                if pmc.get("csrc_hash") == csrc_hash():
                    traffic = pmc["measured_bytes_per_launch"]
                    traffic_note = (f"bytes per conv_gemm-family launch, {pmc['counters']}; algorithmic "
                                    f"{pmc['algorithmic_bytes_per_launch']:.3g} B per launch; {pmc['source']}")
                else:
                    traffic_note = (f"null: profiles/r02_pmc_forward.json was measured on csrc {pmc.get('csrc_hash')}, "
                                    f"this binary is {csrc_hash()}")
            except (OSError, KeyError, ValueError) as e:
                log(f"PMC summary unreadable: {e!r}")
        roof = dict(bound="mfma", achieved=achieved, peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                    frac=achieved / PEAK_FP32_MFMA_TFLOPS, traffic=traffic, traffic_note=traffic_note,
                    kernel="conv_gemm_kernel + lin_gemm_kernel (every conv / Linear of the U-Net forwards of one clip)",
                    method="HIP events around graph replays of each U-Net batch shape x per-op share from one eager "
                           "event-per-op pass (see comment in bench.py)",
                    launches_per_clip=n_launch, avg_launch_us=1e3 * tot_ms / n_launch, by_batch=detail,
                    csrc_hash=csrc_hash(), clip_unet_tflop=per_clip_flops / 1e12,
                    path_tflops=per_clip_flops / (dt / args.steps) / 1e12,
                    path_frac=per_clip_flops / (dt / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS)
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # many-core hosts (the MI355X box has 256) make torch's CPU kernels slower, not faster, on these
        # small tensors: use at most 32 threads and report that number as `cores`
        try:
            base = cpu_baseline(fam, m.state_dicts, min(os.cpu_count() or 1, 32), args.cpu_anchor_steps)
        except Exception as e:          # the headline line must survive a failure of the reported-only baseline leg
            log(f"cpu_baseline failed: {e!r}")

    if rank == 0:
        out = {"metric": "edited-clips/sec (200-step inv+edit, 10 s@16 kHz)", "value": value,
               "unit": "edited-clips/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"AudioLDM2 ({args.model_id}, 346.9M-param U-Net, seeded-random weights) "
                                      f"text-based edit, T={args.T}, tstart={args.tstart}, cfg 3/12, {NC} clip(s) of 10 s "
                                      f"@16 kHz per GPU per step; forward inversion schedule: {args.schedule}"
                                      + (f" ({args.group} timesteps per U-Net call)" if args.schedule == "batched" else
                                         " (reference order)"),
                          "clips_per_gpu_per_step": NC, "parallelism": f"clip-dp{world}",
                          "weights_broadcast_s": t_bcast if world > 1 else 0.0,
                          "gathered_latents": None if gathered is None else [list(g.shape) for g in gathered][:2]},
               "roofline": roof, "cpu_baseline": base, "phases_ms_one_clip": phases}
        out.update(extra)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
