"""Device-resident DDPM-inversion / edit loops (build-plan step 6).

What the reference does per diffusion step (ddm_inversion/inversion_utils.py:74-129 and :221-315):
two batch-1 U-Net calls, ~12 elementwise torch launches and a host<->device round trip
(`int(t)` dict lookups, `alphas_cumprod[t]` indexing).  Here one step is ONE fixed sequence of
native launches -- broadcast x_t into the cond/uncond slots, the U-Net op tape, the fused
CFG + step-math kernel (K1), `advance` -- whose step-dependent data (timestep, scheduler
coefficients, trajectory slices) is indexed ON THE DEVICE from a step counter.  The sequence is
captured once in a hipGraph and replayed T times; the host never synchronises inside the loop.

Batch layout of one U-Net call for n clips and P prompts:  [uncond x n | prompt0 x n | ... ].
Latents are channels-last inside the engine ([.., H, W, C]); NCHW only at the API boundary.

Two schedules for the forward inversion:
  * "sequential" -- the reference's order: step k's U-Net input is the numerically-fixed
    x_t written by step k-1;
  * "batched"    -- G timesteps per U-Net call.  Legal because the edit-friendly inversion draws
    every x_t independently from x_0 (models.py:67-83), so all U-Net inputs are known up front;
    the only deviation is that x_t enters the U-Net before its ~1-ulp numerical fix
    (models.py:114-115).  Measured deviation is reported by tests/bench; default is sequential.
"""
import math

import torch

from . import _lib as L
from .scheduler import coefficient_table
from . import tape as tape_mod
from .tape import Tape
from .unet import PackedUNetWeights, UNetEngine

PAD_BIAS = -1.0e30      # padding keys (beyond a sample's own context length): exp() underflows to exactly 0
MASK_BIAS = -10000.0    # the reference's additive mask value (models.py:740-755)


class Conditioning:
    """Conditioning of ONE U-Net batch row group, already encoded (SURVEY A15 is outside the loop).

    ehs0: [R, L0, d0] or None   (AudioLDM2: GPT-2 generated states; TANGO: T5 states)
    ehs1: [R, L1, d1] or None   (AudioLDM2: T5 states)
    mask0/mask1: [R, L] 0/1 or None
    class_labels: [R, d] or None (AudioLDM-1 CLAP embedding)
    """

    def __init__(self, ehs0=None, ehs1=None, mask0=None, mask1=None, class_labels=None):
        self.ehs0, self.ehs1, self.mask0, self.mask1, self.class_labels = ehs0, ehs1, mask0, mask1, class_labels

    @property
    def rows(self):
        for t in (self.ehs0, self.ehs1, self.class_labels):
            if t is not None:
                return t.shape[0]
        return 0

    def repeat(self, n):
        f = lambda t: None if t is None else t.repeat_interleave(n, 0) if t.shape[0] == 1 else t  # noqa: E731
        return Conditioning(f(self.ehs0), f(self.ehs1), f(self.mask0), f(self.mask1), f(self.class_labels))


def _pad_ctx(parts, L, attr_e, attr_m):
    """Stack per-group context tensors of different lengths into [R, L, d] + additive bias [R, L], on the device the
    conditioning already lives on (no host round trip)."""
    es, bs = [], []
    dev0 = getattr(parts[0], attr_e).device
    for c in parts:
        e = getattr(c, attr_e).float().to(dev0)
        m = getattr(c, attr_m)
        r, l, d = e.shape
        ep = torch.zeros(r, L, d, dtype=torch.float32, device=e.device)
        ep[:, :l] = e
        b = torch.full((r, L), PAD_BIAS, dtype=torch.float32, device=e.device)
        b[:, :l] = 0.0 if m is None else (1 - m.float().to(e.device)) * MASK_BIAS
        es.append(ep)
        bs.append(b)
    return torch.cat(es, 0), torch.cat(bs, 0)


class LoopPlumbing:
    """What the device-resident loop engines share (this module's EditEngine, stable_audio.StableAudioEditEngine): an LRU
    of loop plans (persistent buffers + tapes + one instantiated hipGraph per loop shape) and the graph runner.
    Subclasses provide `self.device`, `self.stream`, `self._plans`, `self.max_plans`."""

    def _drop_plan(self, key):
        old = self._plans.pop(key)
        g = old.get("graph")
        if g is not None:
            if self.stream is not None:
                torch.cuda.synchronize(self.device)       # the graph may be in flight on a pipeline lane, not only on self.stream
            L.check(L.lib().aed_graph_destroy(g), "aed_graph_destroy")

    def _get_plan(self, key):
        """LRU lookup of a loop plan; on a miss makes room for the plan the caller is about to build."""
        plan = self._plans.pop(key, None)
        if plan is not None:
            self._plans[key] = plan                 # re-insert: most recently used last
            return plan
        while self._plans and len(self._plans) >= self.max_plans:
            self._drop_plan(next(iter(self._plans)))
        return None

    def clear_plans(self):
        """Drop every cached loop plan (trajectory buffers + instantiated hipGraphs)."""
        for key in list(self._plans):
            self._drop_plan(key)

    def loop_stream(self):
        """The stream the step graphs are replayed on: the engine's own side stream, or -- while a clip pipeline has
        put the caller on a CU-partition lane (streams.py; `lane_stream` set by pipeline.ClipPipeline) -- that lane."""
        return getattr(self, "lane_stream", None) or self.stream

    def _run_graph(self, body, steps, use_graph=True, plan=None):
        """Run `body()` `steps` times on the loop stream.  The step sequence is captured into a hipGraph
        once per plan (same buffers => same graph for every later clip) and replayed."""
        cur = torch.cuda.current_stream(self.device)
        stream = self.loop_stream()
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            ev0 = torch.cuda.Event(enable_timing=True)
            ev1 = torch.cuda.Event(enable_timing=True)
            if use_graph and steps > 1 and not getattr(self, "eager_steps", False):
                g = plan.get("graph") if plan is not None else None
                if g is None:
                    # a clip pipeline's codec worker issues on the front lane's stream from ANOTHER thread (thread-local capture
                    # mode does not keep its launches out of a capturing stream): capture under the pipeline's lock
                    lock = getattr(self, "capture_lock", None)
                    if lock is None:
                        g = Tape.graph_capture(body)
                    else:
                        with lock:
                            g = Tape.graph_capture(body)
                    if plan is not None:
                        plan["graph"] = g
                ev0.record(stream)
                chooser = getattr(self, "lane_chooser", None)
                if chooser is None:
                    for _ in range(steps):
                        Tape.graph_replay(g)
                else:
                    stream = self._replay_in_chunks(g, steps, stream, chooser)
                ev1.record(stream)
                self._last_events = (ev0, ev1)
                if plan is None:
                    stream.synchronize()
                    L.check(L.lib().aed_graph_destroy(g), "aed_graph_destroy")
            else:
                ev0.record(stream)
                for _ in range(steps):
                    body()
                ev1.record(stream)
                self._last_events = (ev0, ev1)
        cur.wait_stream(stream)

    LANE_CHUNK = 5          # step-graph replays per lane decision (see _replay_in_chunks)

    def _replay_in_chunks(self, g, steps, stream, chooser):
        """`steps` replays of the step graph in chunks of LANE_CHUNK, asking `chooser()` before each chunk which stream the
        chunk goes to (None = stay on `stream`): a clip pipeline widens an edit lane's CU mask once the other stage has
        drained (pipeline.ClipPipeline._back).  The host stays at most two chunks ahead of the device, so the decision is
        taken ~10 steps before the chunk runs instead of 100; a stream change is ordered by an event.  Same graph, same
        kernels, same values on any stream.  Returns the stream the last chunk was issued on."""
        pending, cur_s, done = [], stream, 0
        while done < steps:
            if len(pending) >= 2:
                pending.pop(0).synchronize()
            s = chooser() or stream
            if s is not cur_s:
                hand = torch.cuda.Event()
                hand.record(cur_s)
                s.wait_event(hand)
                cur_s = s
            n = min(self.LANE_CHUNK, steps - done)
            with torch.cuda.stream(cur_s):
                for _ in range(n):
                    Tape.graph_replay(g)
            ev = torch.cuda.Event()
            ev.record(cur_s)
            pending.append(ev)
            done += n
        return cur_s

    def last_loop_ms(self):
        ev0, ev1 = self._last_events
        ev1.synchronize()
        return ev0.elapsed_time(ev1)


class EditEngine(LoopPlumbing):
    def __init__(self, unet_cfg, weights, scheduler, device, H, W, kind):
        self.cfg, self.sched, self.device = unet_cfg, scheduler, torch.device(device)
        self.H, self.W, self.C = H, W, unet_cfg["in_channels"]
        self.kind = kind
        self.weights = weights if isinstance(weights, PackedUNetWeights) else PackedUNetWeights(weights, device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._unets = {}
        # step counter of the loops without a cached plan (DDIM baseline).  Every cached loop plan owns its OWN counter
        # (plan["state"]): the inversion of one clip and the edit loop of another run concurrently in the clip pipeline.
        self.state = torch.zeros(4, dtype=torch.int32, device=self.device)
        self.ts_dev = torch.zeros(scheduler.config.num_train_timesteps, dtype=torch.int64, device=self.device)
        self._ts_host = None    # what ts_dev holds (uploaded again only when the schedule changes)
        self._plans = {}        # loop plans: persistent buffers + tapes + captured graph, keyed by loop shape
        self.max_plans = 8      # least-recently-used plans beyond this are dropped (cfg / tstart sweeps would otherwise
        #                         grow HBM without bound: every plan owns trajectory buffers and an instantiated hipGraph)

    # ------------------------------------------------------------------ helpers
    def _upload_timesteps(self, ts, n):
        """ts_dev[:n] = ts -- skipped when the table already holds these values (same schedule as the previous call: no
        host-to-device copy, and nothing rewrites a table another lane's loop is reading)."""
        ts = ts.detach().to("cpu", torch.int64).reshape(-1)[:n].clone()
        if self._ts_host is not None and self._ts_host.numel() >= n and torch.equal(self._ts_host[:n], ts):
            return
        self.ts_dev[:n] = ts.to(self.device)
        self._ts_host = ts

    # Arithmetic of the LDS-staged GEMMs of this engine's U-Nets (tape.arith_mode; set by PipelineWrapper.editor from
    # `model.arith`).  "bf16x6" (default since round 4): fp32 operands cut exactly into three bf16 pieces in the kernel's
    # loader, six piece products on the bf16 MFMAs, fp32 accumulation -- as close to fp64 as the fp32 MFMA chain
    # (csrc/conv_gemm_x6.hip); which GEMMs take it is decided per shape by the swept tables
    # (tape.X6_TABLES) or, without an entry, by "every LDS-staged tile".  "f32": v_mfma_f32_32x32x2_f32 everywhere.
    # Engines below ARITH_MIN_BATCH rows stay fp32.
    arith = "bf16x6"
    ARITH_MIN_BATCH = 2

    def _arith_for(self, B):
        return self.arith if B >= self.ARITH_MIN_BATCH else "f32"

    # CFG row sharing (unet.UNetEngine `share`): the loops below lay the batch out as [uncond | prompt_0 | ...] blocks that all
    # carry the same x_t and timestep; with ONE clip per engine (n = 1) the blocks of a timestep are adjacent rows and the
    # context-free head of the U-Net is computed once per timestep instead of once per row.  False = every row through the whole
    # graph (rounds 1-4; A/B switch).
    # Both loops share: the timestep-batched inversion (shared head at batch G >= 2) and, since round 6, the edit loop (shared head at
    # batch 1).  Round 5 kept the edit loop out because that engine came out different from run to run when VAE-encode kernels were
    # co-resident on the lane's CUs; round 6 named the node -- the gather loader of csrc/lin_gemm.hip (tiles 11 / 15), which the shared
    # head is the only user of at M = 1024 -- reproduced it outside the engine and fixed the kernel (profiles/r06_lin_gather_hazard.md;
    # tests/test_gpu_coresidency.py).  SHARE_IN_EDIT_LOOP stays as the A/B switch.
    SHARE_CFG_ROWS = True
    SHARE_IN_EDIT_LOOP = True

    def unet(self, B, L0=0, L1=0, share=1):
        arith = self._arith_for(B)
        share = int(share) if (self.SHARE_CFG_ROWS and share > 1 and self.kind != "audioldm") else 1
        key = (B, L0, L1) if arith == "f32" else (B, L0, L1, arith)
        if share > 1:
            key = key + (f"share{share}",)
        if key not in self._unets:
            with tape_mod.arith_mode(arith):
                self._unets[key] = UNetEngine(self.cfg, self.weights, self.device, B, self.H, self.W, ctx_len0=L0,
                                              ctx_len1=L1, use_ehs=self.kind != "audioldm",
                                              timesteps_dev=self.ts_dev, state_dev=self.state, share=share)
        return self._unets[key]

    def _set_cond(self, eng, groups, repeat=1):
        """groups: list of Conditioning, concatenated along the batch in order; the whole list `repeat` times (the
        timestep-batched inversion evaluates the same [uncond | prompt] rows for G timesteps: the rows are padded /
        concatenated once and tiled, not rebuilt G times on the host)."""
        tile = (lambda t: t) if repeat == 1 else (lambda t: t.repeat(repeat, *([1] * (t.dim() - 1))))
        if self.kind == "audioldm":
            d0 = groups[0].class_labels.device
            eng.set_conditioning(class_labels=tile(torch.cat([g.class_labels.float().to(d0) for g in groups], 0)))
        elif self.kind == "audioldm2":
            d0 = groups[0].ehs0.device
            e0 = torch.cat([g.ehs0.float().to(d0) for g in groups], 0)
            e1, b1 = _pad_ctx(groups, eng.L1, "ehs1", "mask1")
            eng.set_conditioning(ehs0=tile(e0), ehs1=tile(e1), bias1=tile(b1))
        else:
            e0, b0 = _pad_ctx(groups, eng.L0, "ehs0", "mask0")
            eng.set_conditioning(ehs0=tile(e0), bias0=tile(b0))

    def _coef_table(self, s, ts, eta, kind):
        """scheduler.coefficient_table, memoised: the rows are host scalar arithmetic in the reference's expression order
        (23 ms of Python for T = 200) and depend only on the schedule, eta and the table kind -- a serving loop asks for
        the same table clip after clip."""
        key = (kind, tuple(int(t) for t in ts), tuple(eta) if isinstance(eta, (list, tuple)) else float(eta),
               int(s.config.num_train_timesteps), int(s.num_inference_steps), str(s.config.prediction_type),
               float(s.alphas_cumprod[0]), float(s.alphas_cumprod[-1]), float(s.final_alpha_cumprod))
        cache = self.__dict__.setdefault("_coef_cache", {})
        tab = cache.pop(key, None)
        if tab is None:
            tab = coefficient_table(s, ts, eta=eta, kind=kind)
            while len(cache) >= 8:
                cache.pop(next(iter(cache)))
        cache[key] = tab
        return tab

    @staticmethod
    def _round_len(n):
        """Padded context length: the next power of two in [8, 32] (padding keys carry an exactly-zero weight, PAD_BIAS),
        so the folded cross-attention's per-head softmax groups fit a 32-column tile; longer contexts stay as they are."""
        for p2 in (8, 16, 32):
            if n <= p2:
                return p2
        return n

    def _ctx_lens(self, groups):
        if self.kind == "audioldm":
            return 0, 0
        if self.kind == "audioldm2":
            return groups[0].ehs0.shape[1], self._round_len(max(g.ehs1.shape[1] for g in groups))
        return self._round_len(max(g.ehs0.shape[1] for g in groups)), 0

    @torch.inference_mode()
    def to_nhwc(self, x, out=None):
        """[..., C, H, W] -> contiguous [..., H, W, C] on the device (native transpose kernel)."""
        lead = x.shape[:-3]
        C, H, W = x.shape[-3:]
        src = x.to(self.device, torch.float32).contiguous()
        dst = out if out is not None else torch.empty(*lead, H, W, C, device=self.device, dtype=torch.float32)
        tp = Tape(self.device)
        tp.transpose(src, dst, Bt=max(1, math.prod(lead)), R=C, C=H * W)
        tp.run()
        return dst

    @torch.inference_mode()
    def to_nchw(self, x):
        lead = x.shape[:-3]
        H, W, C = x.shape[-3:]
        src = x.contiguous()
        dst = torch.empty(*lead, C, H, W, device=self.device, dtype=torch.float32)
        tp = Tape(self.device)
        tp.transpose(src, dst, Bt=max(1, math.prod(lead)), R=H * W, C=C)
        tp.run()
        return dst

    # ------------------------------------------------------------------ A6: sample_xts_from_x0
    @torch.inference_mode()
    def sample_xts(self, x0, noise=None, generator=None):
        """models.py:67-83.  x0 [n,C,H,W]; noise [T,n,C,H,W] (drawn here on the CPU generator in the
        reference's order -- ascending t, one randn per step -- when not given).  Returns NCHW xts
        [T+1, n, C, H, W] on the device."""
        s = self.sched
        T = s.num_inference_steps
        x0 = x0.to(self.device, torch.float32).contiguous()
        if noise is None:
            noise = torch.stack([torch.randn(x0.shape, generator=generator, dtype=torch.float32) for _ in range(T)])
        noise = noise.to(self.device, torch.float32).contiguous()
        ts = s.timesteps.cpu()
        abar = s.alphas_cumprod
        # row r <-> idx = r+1 <-> t = timesteps[T - idx]
        t_rows = torch.stack([ts[T - (r + 1)] for r in range(T)])
        sa = (abar[t_rows] ** 0.5).to(self.device)
        sb = ((1 - abar) ** 0.5)[t_rows].to(self.device)
        xts = torch.empty((T + 1, *x0.shape), device=self.device, dtype=torch.float32)
        xts[0] = x0
        L.check(L.lib().aed_sample_xts_from_x0(x0.data_ptr(), noise.data_ptr(), sa.data_ptr(), sb.data_ptr(),
                                                xts[1:].data_ptr(), T, x0.numel(), L.current_stream_ptr()),
                "aed_sample_xts_from_x0")
        return xts

    @staticmethod
    def _etas_in_loop_order(eta, n_rows):
        """The reference indexes its per-step list as `etas[idx]` with idx DEscending along the loop (idx = T - k - 1 in
        the inversion, Z - k - 1 in the edit; inversion_utils.py:75,124 and :221-224,302): row k of the coefficient table
        gets etas[n_rows - 1 - k].  A scalar stays a scalar."""
        if not isinstance(eta, (list, tuple)) and not (torch.is_tensor(eta) and eta.dim() > 0):
            return float(eta)
        etas = [float(e) for e in eta]
        if len(etas) < n_rows:
            raise ValueError(f"{len(etas)} eta values for {n_rows} steps")
        return [etas[n_rows - 1 - k] for k in range(n_rows)]

    # ------------------------------------------------------------------ A7: forward inversion
    @torch.inference_mode()
    def invert(self, x0, cond_src, cond_uncond, cfg_src, eta=1.0, numerical_fix=True, noise=None, generator=None,
               xts=None, cfg_tensor=None, mode="sequential", group=8, use_graph=True):
        """inversion_forward_process (inversion_utils.py:8-144) for n clips.

        x0 [n,C,H,W]; cond_src: Conditioning with n*P rows ordered [prompt0 x n, prompt1 x n, ...] or None
        for an empty source prompt (cond pass skipped, inversion_utils.py:86); cond_uncond: 1 or n rows.
        Returns (zs, xts) channels-last on the device: zs [T,n,H,W,C], xts [T+1,n,H,W,C]."""
        s = self.sched
        T = s.num_inference_steps
        n = x0.shape[0]
        numel = n * self.C * self.H * self.W
        if xts is None:
            xts = self.sample_xts(x0, noise, generator)
        P = 0 if cond_src is None else cond_src.rows // n
        groups = [cond_uncond.repeat(n)] + ([cond_src] if P else [])
        v_pred = int(s.config.prediction_type == "v_prediction")
        G = 1 if mode == "sequential" else max(1, min(group, T))
        while T % G:
            G -= 1
        rows_per_t = n * (1 + P)
        L0, L1 = self._ctx_lens(groups)
        scalar = float(cfg_src[0]) if (cfg_tensor is None and P) else 1.0
        key = ("invert", n, P, T, G, L0, L1, bool(numerical_fix), v_pred, cfg_tensor is not None, scalar,
               self._arith_for(G * rows_per_t))
        plan = self._get_plan(key)
        if plan is None:
            plan = self._plans[key] = dict(
                state=torch.zeros(4, dtype=torch.int32, device=self.device),
                xts=torch.empty((T + 1, n, self.H, self.W, self.C), device=self.device, dtype=torch.float32),
                zs=torch.zeros((T, n, self.H, self.W, self.C), device=self.device, dtype=torch.float32),
                coef=torch.zeros((T, L.COEF_STRIDE), device=self.device, dtype=torch.float32),
                cfgt=(torch.empty((max(P, 1), n, self.H, self.W, self.C), device=self.device, dtype=torch.float32)
                      if cfg_tensor is not None else None))
            eng = plan["eng"] = self.unet(G * rows_per_t, L0, L1, share=(1 + P) if (n == 1 and G >= 2) else 1)
            pre, post = Tape(self.device), Tape(self.device)
            for g in range(G):
                for blk in range(1 + P):
                    dst = eng.x_in[(g * (1 + P) + blk) * n:(g * (1 + P) + blk + 1) * n]
                    pre.copy2d(plan["xts"], dst, rows=1, cols=numel, ld_src=numel, ld_dst=numel, state=plan["state"],
                               idx_off=T - g, idx_mul=-G, idx_stride=numel, name="x_in<-xts")
            for g in range(G):
                base = g * rows_per_t
                eps_u = eng.eps[base:base + n]
                eps_c = eng.eps[base + n:base + rows_per_t] if P else None
                post.step(L.OP_INVERT_STEP, xts=plan["xts"], zs=plan["zs"], eps_u=eps_u, eps_c=eps_c,
                          cfg=plan["cfgt"], coef=plan["coef"], state=plan["state"], out=None, numel=numel, P=max(P, 1),
                          T=T, v_pred=v_pred, flag=int(numerical_fix), cfg_scalar=scalar, s_mul=G, s_off=g)
            post.advance(plan["state"])
            pre.finalize()
            post.finalize()
            plan["pre"], plan["post"] = pre, post
        eng, pre, post = plan["eng"], plan["pre"], plan["post"]
        xts = self.to_nhwc(xts, out=plan["xts"])                  # [T+1, n, H, W, C]
        zs = plan["zs"]
        plan["coef"].copy_(self._coef_table(s, s.timesteps.cpu(), self._etas_in_loop_order(eta, T), "ddpm"))
        self._upload_timesteps(s.timesteps, T)
        if cfg_tensor is not None:
            self.to_nhwc(cfg_tensor.reshape(P, n, self.C, self.H, self.W), out=plan["cfgt"])
        # batch rows: for g in G: [uncond x n | prompt_p x n ...]
        self._set_cond(eng, groups, repeat=G)
        self._patch_time(eng, self.ts_dev, G, rows_per_t, state=plan["state"])
        plan["state"].zero_()

        def body():
            pre.run()
            eng.tape.run()
            post.run()
        self._run_graph(body, T // G, use_graph, plan)
        zs[0].zero_()                                              # inversion_utils.py:131-133
        return zs, xts          # persistent buffers of this plan: valid until the next invert() of the same shape

    def _patch_time(self, eng, ts_dev, G, rows_per_t, offset=0, state=None):
        """Point the U-Net's time-embedding op at (table + offset) with G timesteps per call, stepped by `state`."""
        state = self.state if state is None else state
        op = eng.tape.ops[eng.time_op]
        arr = eng.tape.finalize()
        # cached: captured graphs keep pointing at these index tables
        cache = eng.__dict__.setdefault("_row_tidx_cache", {})
        if rows_per_t not in cache:
            cache[rows_per_t] = (torch.arange(eng.B, dtype=torch.int32) // max(1, rows_per_t)).to(self.device)
        eng._row_tidx = cache[rows_per_t]
        for o in (op, arr[eng.time_op]):
            o.p[1] = ts_dev.data_ptr() + 8 * offset
            o.p[2] = state.data_ptr()
            o.p[4] = eng._row_tidx.data_ptr()
            o.i[5] = G

    # ------------------------------------------------------------------ A11: reverse / edit
    @torch.inference_mode()
    def edit(self, xts, zs, tstart, cond_tgt, cond_neg, cfg_tar, eta=1.0, cfg_tensor=None, use_graph=True,
             table_kind="ddpm", n_steps=None):
        """inversion_reverse_process (inversion_utils.py:147-323) from x_{tstart}, noise maps zs[:tstart].
        xts/zs channels-last as returned by invert().  Returns the edited latent [n,H,W,C].
        n_steps < tstart stops early and returns x_{tstart - n_steps} (trajectory-replay checks)."""
        s = self.sched
        T = s.num_inference_steps
        n = xts.shape[1]
        numel = n * self.C * self.H * self.W
        Z = int(tstart)
        P = cond_tgt.rows // n
        groups = [cond_neg.repeat(n), cond_tgt]
        ts = s.timesteps.cpu()[T - Z:]
        v_pred = int(s.config.prediction_type == "v_prediction")
        scalar = float(cfg_tar[0]) if cfg_tensor is None else 1.0
        L0, L1 = self._ctx_lens(groups)
        eta_rows = self._etas_in_loop_order(eta, Z)
        any_noise = (eta_rows > 0) if isinstance(eta_rows, float) else any(e > 0 for e in eta_rows)
        has_noise = int(any_noise and zs is not None)
        key = ("edit", n, P, T, Z, L0, L1, v_pred, cfg_tensor is not None, scalar, has_noise, table_kind,
               self._arith_for(n * (1 + P)))
        plan = self._get_plan(key)
        if plan is None:
            plan = self._plans[key] = dict(
                state=torch.zeros(4, dtype=torch.int32, device=self.device),
                cur=torch.empty((n, self.H, self.W, self.C), device=self.device, dtype=torch.float32),
                zs=torch.zeros((Z, n, self.H, self.W, self.C), device=self.device, dtype=torch.float32),
                coef=torch.zeros((Z, L.COEF_STRIDE), device=self.device, dtype=torch.float32),
                cfgt=(torch.empty((P, n, self.H, self.W, self.C), device=self.device, dtype=torch.float32)
                      if cfg_tensor is not None else None))
            eng = plan["eng"] = self.unet(n * (1 + P), L0, L1, share=(1 + P) if (n == 1 and self.SHARE_IN_EDIT_LOOP) else 1)
            pre, post = Tape(self.device), Tape(self.device)
            for blk in range(1 + P):
                pre.copy2d(plan["cur"], eng.x_in[blk * n:(blk + 1) * n], rows=1, cols=numel, ld_src=numel,
                           ld_dst=numel, name="x_in<-x_t")
            post.step(L.OP_REVERSE_STEP, xts=plan["cur"], zs=plan["zs"] if has_noise else None, eps_u=eng.eps[:n],
                      eps_c=eng.eps[n:], cfg=plan["cfgt"], coef=plan["coef"], state=plan["state"], out=plan["cur"],
                      numel=numel, P=P, T=Z if has_noise else 0, v_pred=v_pred, flag=has_noise, cfg_scalar=scalar)
            post.advance(plan["state"])
            pre.finalize()
            post.finalize()
            plan["pre"], plan["post"] = pre, post
        eng, pre, post, cur = plan["eng"], plan["pre"], plan["post"], plan["cur"]
        cur.copy_(xts[Z])                                          # inversion_utils.py:203
        if has_noise:
            plan["zs"].copy_(zs[:Z])
        plan["coef"].copy_(self._coef_table(s, ts, eta_rows, table_kind))
        self._upload_timesteps(s.timesteps, T)
        if cfg_tensor is not None:
            self.to_nhwc(cfg_tensor.reshape(P, n, self.C, self.H, self.W), out=plan["cfgt"])
        self._set_cond(eng, groups)
        self._patch_time(eng, self.ts_dev, 1, n * (1 + P), offset=T - Z, state=plan["state"])
        plan["state"].zero_()

        def body():
            pre.run()
            eng.tape.run()
            post.run()
        self._run_graph(body, Z if n_steps is None else max(0, min(int(n_steps), Z)), use_graph, plan)
        return cur.clone()

    # ------------------------------------------------------------------ A16: DDIM baseline
    @torch.inference_mode()
    def ddim_invert(self, w0, cond_src, cond_uncond, cfg_scale, skip=0, use_graph=True):
        """ddim_inversion (ddim_inversion.py:44-56): deterministic, ascending t.  w0 [n,C,H,W] -> [n,H,W,C]."""
        s = self.sched
        T = s.num_inference_steps
        n = w0.shape[0]
        numel = n * self.C * self.H * self.W
        steps = T - skip
        ts_asc = torch.flip(s.timesteps.cpu(), dims=[0])[:steps]
        coef = coefficient_table(s, ts_asc, kind="ddim_next").to(self.device)
        self.ts_dev[:steps] = ts_asc.to(self.device)
        self._ts_host = None                            # the table no longer holds the descending schedule
        groups = [cond_uncond.repeat(n), cond_src]
        L0, L1 = self._ctx_lens(groups)
        eng = self.unet(2 * n, L0, L1)
        self._set_cond(eng, groups)
        cur = self.to_nhwc(w0).contiguous()
        pre, post = Tape(self.device), Tape(self.device)
        for blk in range(2):
            pre.copy2d(cur, eng.x_in[blk * n:(blk + 1) * n], rows=1, cols=numel, ld_src=numel, ld_dst=numel)
        self._patch_time(eng, self.ts_dev, 1, 2 * n)
        post.step(L.OP_REVERSE_STEP, xts=cur, zs=None, eps_u=eng.eps[:n], eps_c=eng.eps[n:], cfg=None, coef=coef,
                  state=self.state, out=cur, numel=numel, P=1, T=0, flag=0, cfg_scalar=float(cfg_scale))
        post.advance(self.state)
        self.state.zero_()

        pre.finalize()
        post.finalize()

        def body():
            pre.run()
            eng.tape.run()
            post.run()
        self._run_graph(body, steps, use_graph)
        return cur

    @torch.inference_mode()
    def ddim_sample(self, xt, cond_tgt, cond_uncond, guidance_scale, skip=0, use_graph=True):
        """text2image_ldm_stable (ddim_inversion.py:59-84): scheduler.step(eta=0) from timesteps[skip:]."""
        s = self.sched
        T = s.num_inference_steps
        xts_like = xt.unsqueeze(0).expand(T - skip + 1, *xt.shape)
        return self.edit(xts_like, None, T - skip, cond_tgt, cond_uncond, [guidance_scale], eta=0.0,
                         use_graph=use_graph, table_kind="ddim_prev")

    # ------------------------------------------------------------------ n clips, latent in -> edited latent out
    @torch.inference_mode()
    def edit_latents(self, x0, cond_src, cond_uncond, cond_tgt, cond_neg, cfg_src, cfg_tar, tstart, eta=1.0,
                     schedule="sequential", group=8, noise=None, generator=None):
        """Inversion + edit of n independent clips as ONE U-Net batch per step (BASELINE config 3: 8 clips per GPU):
        x0 [n,C,H,W]; every Conditioning has 1 row (shared by the clips) or n rows (per clip); cond_src may be None
        (empty source prompt).  x_t noise is drawn per timestep for all clips at once on the CPU generator.
        Returns the edited latents [n,C,H,W] on the device."""
        n = x0.shape[0]
        rep = lambda c: None if c is None else c.repeat(n)                  # noqa: E731  (1 row -> n rows)
        xts0 = self.sample_xts(x0, noise, generator)
        zs, xts = self.invert(x0, rep(cond_src), cond_uncond, cfg_src, eta=eta, numerical_fix=True, xts=xts0,
                              mode=schedule, group=group)
        w = self.edit(xts, zs, int(tstart), rep(cond_tgt), cond_neg, cfg_tar, eta=eta)
        return self.to_nchw(w)
