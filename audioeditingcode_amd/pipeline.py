"""Several clips in flight on one MI355X (BASELINE config 2 served as a stream of independent clips).

Why.  One clip of the reference path is a sequential chain of two regimes: the forward inversion
(inversion_utils.py:75-133; timestep-batched here: two U-Net calls at batch 200, throughput-bound: 0.39 of the bf16 MFMA peak
on the whole chip in the split-bf16 arithmetic, 0.49 alone on a 128-CU partition) and the 100-step edit loop (:221-315; ~570
dependent launches per step at U-Net batch 2, bound by launch / first-operand latency: 7.9 ms per step on 256 CUs, 13.0 ms on a
64-CU lane -- it cannot use the chip).  A clip's own arithmetic cannot be reordered, so the only work that can fill the idle
compute units is ANOTHER clip.  Since round 5 the steady state is bound by the chip's POWER budget, not by idle CUs
(profiles/r05_power_probe.md): variants that only re-schedule work (lockstep edit groups, work stealing between the stages, whole
clips on unpartitioned lanes) measured within 2 % of this layout and were removed from the package in round 6 (git history, and
their write-ups: profiles/r05_group_plan.md, profiles/r03_lanes.md).

The plan (measured on the MI355X, profiles/r03_cu_partition.md):
  * a two-stage pipeline on DISJOINT CU partitions (streams.PartitionStream: hardware queues with
    CU masks).  Stage "front" (one worker, CUs [edit_cus, total)): waveform -> mel -> VAE encode -> forward inversion.
    Stage "back" (edit_lanes workers sharing CUs [0, edit_cus)): edit loop -> VAE decode -> vocoder.  While clip i is in
    its edit loop, clip i+1 is being inverted; the two kernel classes never queue behind each other's workgroups (without
    masks a batch-2 kernel waits for 128x128-tile workgroups that hold a CU for 0.2-1 ms: 28.9 ms per edit step).
    Fill and drain (no other stage to share with) run on the whole chip.

Nothing about a clip's computation changes: a worker's plans, tapes, hipGraphs and kernels run in the same order on the
same values whether other clips are in flight or not -- only the stream they are launched on differs -- so a pipelined clip
is bit-identical to the same clip pushed through the SAME ENGINES alone (tests/test_gpu_pipeline.py, asserted by bench.py).
Against main_run.edit_clip on the model's own whole-chip engines the values agree to ~1e-6 relative, not bit for bit: a back-stage
worker builds its engines under tape.tile_regime("cus128" / "cus64"), i.e. with other tiles and split-K summation orders.

Workers are host threads (one per lane; the HIP calls release the GIL), each with a lane view of the wrapper
(models.PipelineWrapper.lane_view: shares the frozen weights / scheduler / text encoders, owns every mutable buffer --
engines, loop plans, hipGraphs).  RNG: the reference draws a clip's T noise maps from torch's global CPU generator
(models.py:76-81); clip i draws (after an optional `torch.manual_seed(seeds[i])`) only when clips 0..i-1 have drawn
(`_DrawGate`), so every clip sees the noise a serial loop over the clips would have given it.
"""
import contextlib
import queue
import threading
import time

import torch

from . import tape as tape_mod
from .ddm_inversion.inversion_utils import (conditioning_from_text, inversion_reverse_process, prepare_forward,
                                             run_forward)
from .streams import PartitionStream

DEFAULT_EDIT_CUS = 128          # CUs of the edit-loop partition (the two stages' per-clip times cross near 128 of 256)
_STOP = object()


class _DrawGate:
    """Orders the per-clip draws from the global CPU generator by clip index."""

    def __init__(self):
        self.cv = threading.Condition()
        self.next = 0
        self.failed = None

    @contextlib.contextmanager
    def turn(self, index, seed):
        with self.cv:
            while self.next != index and self.failed is None:
                self.cv.wait(timeout=1.0)
            if self.failed is not None:
                raise RuntimeError(f"clip {self.failed} failed before its noise draw; clip {index} cannot keep the "
                                   f"serial draw order")
        try:
            if seed is not None:
                torch.manual_seed(seed)
            yield
        finally:
            self.done(index)

    def done(self, index, failed=False):
        """Clip `index` has drawn (or will never draw): let the next one through."""
        with self.cv:
            if failed and self.next <= index:
                self.failed = index
            if self.next == index:
                self.next = index + 1
            self.cv.notify_all()


class _Worker:
    """One lane: a lane view of the model, its stream on the stage's CU partition and (partition plan) an unmasked
    stream for fill / drain."""

    def __init__(self, stage, k, view, lane, full, regime=None, prep=None):
        self.stage, self.k, self.view, self.lane, self.full = stage, k, view, lane, full
        self.regime = regime        # tile regime its engines are built under (tape.tile_regime)
        self.prep = prep            # side stream for the part of a front half that does not touch the loop engine
        self.wide = None            # edit lanes: this lane's CUs + a slice of the inversion partition (drain, see _back)
        self.last = None            # (event, stream) of this worker's previous job
        self.warm = False


class ClipPipeline:
    # the HIP touch points (the CPU host-logic tests substitute recording stand-ins)
    lane_type = PartitionStream
    event_type = torch.cuda.Event

    def __init__(self, model, plan="partition", edit_cus=None, edit_lanes=1, launch="graph", timestep_group=100,
                 overlap_prep=True, separate_queues=None, codec_stage=None, widen_on_drain=True, codec_queue="front",
                 mask_prep=True):
        if getattr(model, "kind", None) == "stable_audio":
            raise NotImplementedError("ClipPipeline drives the mel-latent families (AudioLDM / AudioLDM2 / TANGO)")
        if plan != "partition":
            raise ValueError("plan must be 'partition' (the unpartitioned 'lanes' plan was removed in round 6)")
        if launch not in ("eager", "graph"):
            raise ValueError("launch must be 'eager' or 'graph'")
        self.model, self.plan, self.launch = model, plan, launch
        self.timestep_group = int(timestep_group)
        dev = model.device
        acquire = getattr(self.lane_type, "acquire", None)

        def Lane(device, cus=None, total=None, index=0):           # process-lifetime streams (streams.PartitionStream)
            if acquire is not None:
                return acquire(device, cus=cus, total=total, index=index)
            return self.lane_type(device, cus=cus, total=total)
        self.full = Lane(dev)                                      # the whole chip
        self.total = self.full.total
        self.stages = []                                           # [(name, halves, [workers])]
        self.edit_cus = DEFAULT_EDIT_CUS if edit_cus is None else int(edit_cus)
        if not 0 < self.edit_cus < self.total:
            raise ValueError(f"edit_cus={self.edit_cus} must leave CUs for both partitions of {self.total}")
        if self.timestep_group < 2:
            raise ValueError("the partition plan needs the timestep-batched inversion (timestep_group >= 2): the "
                             "front stage must not share the edit loop's batch-2 engine regime")
        self.edit_lanes = max(1, int(edit_lanes))
        # The next clip's mel / VAE encode / text conditioning / x_t draws + upload do not touch the batch-2G loop engine:
        # they go to a side stream, so their host-blocking copies and checks wait for THAT stream and the work itself
        # overlaps the inversion still running on the partition (otherwise ~50 ms of set-up per clip sit exposed
        # between two inversions: `assert min(y) >= -1` alone drains the lane before anything else is enqueued).
        front_regime = None
        for name, lo, hi in (("cus128", 96, 160), ("cus64", 24, 80)):       # tables swept on a stream of about that size
            if lo <= self.total - self.edit_cus <= hi and name in tape_mod.REGIME_TABLES:
                front_regime = name
        self._front_regime = front_regime
        # The side stream is MASKED to the inversion partition (round 6): the next clip's VAE encode has no business on the edit
        # lanes' CUs -- co-resident split-bf16 workgroups are what exposed the round-5 hazard (profiles/r06_lin_gather_hazard.md),
        # and the edit lanes' latency-bound kernels lose issue slots to them.  mask_prep=False: the unmasked queue of rounds 3-5 (A/B).
        inv_cus = range(self.edit_cus, self.total)
        prep = None
        if overlap_prep:
            prep = Lane(dev, cus=inv_cus, total=self.total, index=17) if mask_prep else Lane(dev, index=17)
        self.mask_prep = bool(mask_prep) and overlap_prep
        front = [_Worker("front", 0, self._view(), Lane(dev, cus=inv_cus, total=self.total), self.full, regime=front_regime,
                         prep=prep)]
        # the edit loop's batch-2 kernels on half the chip are no longer purely latency-bound: their tiles come from
        # the sweep taken on a 128-CU stream (tile_table_cus128.py) when the partition is about that size
        # Several edit lanes get DISJOINT slices of the edit partition when it splits into multiples of 32 CUs (a mask must
        # give every shader engine of every XCD the same number of CUs; 64 consecutive mask bits = 8 CUs on each XCD):
        # lane k runs on CUs [k * edit_cus / n, (k + 1) * edit_cus / n).  Otherwise the lanes SHARE the partition's CUs --
        # measured in round 3 (128 CUs, 2 lanes: 2 x 2.87 s per clip per lane = the one-lane rate) and again in round 4 with
        # one dispatch pipe per lane attempted (queues with the SAME mask cannot be separated: 0.71-0.78 clips/s against
        # 0.95 for disjoint slices, profiles/r04_pipeline_variants.md).
        n = self.edit_lanes
        per = self.edit_cus // n if (n > 1 and self.edit_cus % n == 0 and (self.edit_cus // n) % 32 == 0) else None
        self.edit_lane_cus = per or self.edit_cus
        lane_cus = (lambda k: range(k * per, (k + 1) * per)) if per else (lambda k: range(self.edit_cus))
        regime = None
        for name, lo, hi in (("cus128", 96, 160), ("cus64", 24, 80)):       # tables swept on a stream of about that size
            if lo <= self.edit_lane_cus <= hi and name in tape_mod.REGIME_TABLES:
                regime = name
        back = [_Worker("back", k, self._view(), Lane(dev, cus=lane_cus(k), total=self.total, index=k),
                        Lane(dev, index=1 + k), regime=regime) for k in range(n)]
        # Several edit lanes: the edited latent's VAE decode + vocoder (throughput kernels, 44 ms on the whole chip, ~150 ms on
        # a 64-CU lane) leave the lane.  A third stage with ONE worker decodes ON THE INVERSION PARTITION'S OWN QUEUE, in
        # stream order between that stage's inversions: the front stage has the slack once two lanes share the back stage's
        # work, and a queue of its own over the same CUs was measured 8x slower (codec and batch-200 workgroups then
        # compete for the same CUs one workgroup at a time: 600 ms per clip, and the inversion slows as well).  The next
        # clip's set-up stays on the front lane too: three busy hardware queues, each on a dispatch pipe of its own
        # (streams.py).  With one edit lane the codec stays in the back stage on an unmasked stream (round 3).
        self.codec_stage = n > 1 if codec_stage is None else bool(codec_stage)
        self.stages = [("front", ("front",), front), ("back", ("back",), back)]
        if codec_queue not in ("front", "chip", "lane"):
            raise ValueError("codec_queue must be 'front' (the inversion partition's queue), 'chip' (an unmasked queue) or "
                             "'lane' (the edit lane that edited the clip)")
        self.codec_queue = codec_queue
        if codec_queue == "lane":
            # Round 5: with the split-K tables an edit lane needs 1.3 s per clip and gets one every 1.66 s (two lanes, front
            # stage 0.83 s per clip): the lanes have the slack, the inversion queue has none.  The edited latent's VAE decode +
            # two vocoder passes (44 ms on the whole chip, ~110 ms on 64 CUs) run on the lane that edited the clip, in stream
            # order after its loop; no codec stage, and the next clip's set-up goes back to a side stream (round 3's
            # overlap_prep) so that the inversion queue carries the two U-Net calls and little else.
            self.codec_stage = False
        if self.codec_stage:
            front[0].prep = None
            # "chip" (round 5 A/B): the codec jobs on an UNMASKED queue of their own -- 44 ms of throughput kernels per clip
            # leave the inversion queue (the critical stage once the edit lanes got faster) and take whatever CU is free
            cq = front[0].lane if codec_queue == "front" else Lane(dev, index=70)
            self.stages.append(("codec", ("codec",), [_Worker("codec", 0, self._view(), cq, None)]))
        self.queue_log = []
        if separate_queues is None:
            separate_queues = n > 1
        if separate_queues and hasattr(self.lane_type, "respin") and dev.type == "cuda":
            # every lane that is busy at the same time needs its own dispatch pipe (streams.py)
            from .streams import separate_queues as _separate
            busy = [front[0].lane] + [w.lane for w in back]            # (the first stream is never replaced)
            kept = _separate(busy, log=self.queue_log)
            for w, ps in zip(back, kept[1:1 + n]):
                w.lane = ps
        # Drain: when the front stage has finished its last inversion, its CUs idle while the last edit loops run on
        # their 64-CU lanes for another ~1.6 s.  Each edit lane gets a second queue over its own CUs PLUS its share of
        # the inversion partition; the edit loop is issued in chunks of a few steps and moves there once the front
        # stage is done (editing.LoopPlumbing._replay_in_chunks).  Same engines, same graphs, same values.
        inv = self.total - self.edit_cus
        if widen_on_drain and per and n > 1 and inv % 32 == 0 and inv // 32 >= n:
            # the inversion partition is handed out in 32-CU units (a legal mask gives every shader engine of every XCD the
            # same number of CUs): lane k gets units // n of them, the first units % n lanes one more
            units, lo = inv // 32, self.edit_cus
            for k, w in enumerate(back):
                take = 32 * (units // n + (1 if k < units % n else 0))
                w.wide = Lane(dev, cus=list(lane_cus(k)) + list(range(lo, lo + take)), total=self.total, index=50 + k)
                lo += take
            if separate_queues and hasattr(self.lane_type, "respin") and dev.type == "cuda":
                # busy together in the drain: the inversion queue (codec jobs) and the widened lanes
                from .streams import separate_queues as _separate
                self.drain_queue_log = []
                kept = _separate([front[0].lane] + [w.wide for w in back], log=self.drain_queue_log)
                for w, ps in zip(back, kept[1:]):
                    w.wide = ps
        self._build_lock = threading.Lock()     # an un-warmed worker builds engines (and lazily folds shared weights)
        # The codec worker issues on the FRONT lane's stream from its own thread.  hipStreamBeginCapture(ThreadLocal) does not keep
        # another thread's launches out of a capturing stream: a late graph capture of the front worker (a new loop shape after
        # warm-up, an LRU plan eviction) must not interleave with a codec job.  Every lane view carries this lock;
        # LoopPlumbing._run_graph holds it while capturing, _codec while issuing.  The lock is one-directional on purpose:
        # the codec engines (codec.py) never capture -- their tapes are launched record by record -- so the only captures on the front
        # lane's stream are the front worker's own loop plans, taken by the thread that also issues everything else on that stream.
        self._capture_lock = threading.Lock()
        for w in self.workers:
            w.view._capture_lock = self._capture_lock
        self.stats = []
        self.widened = set()                    # edit lanes that moved to their widened queue during the last edit_clips

    def _view(self):
        v = self.model.lane_view()
        # "graph": one hipGraphLaunch per diffusion step (0.28 ms of host time for ~610 kernel nodes); "eager": the step's
        # launches issued one by one from C++ (2.2 ms).  Same kernels and values either way; measured equal on the device.
        v._lane_eager = self.launch == "eager"
        return v

    @property
    def workers(self):
        return [w for _, _, ws in self.stages for w in ws]

    @property
    def clips_in_flight(self):
        return len(self.workers)

    @staticmethod
    def _stream_ctx(stream):
        return torch.cuda.stream(stream)

    # ------------------------------------------------------------------ lane plumbing
    @contextlib.contextmanager
    def _on(self, w, lane):
        """Run the enclosed host code with `lane.stream` as torch's current stream and as the loop engines' replay stream.
        A worker's engines and plans are reused clip after clip: when it moves to its other stream (fill / drain), that
        stream first waits for the worker's previous job."""
        st = lane.stream
        if w.last is not None and w.last[1] is not st:
            st.wait_event(w.last[0])
        w.view._lane_stream = st
        t0, t1 = self.event_type(enable_timing=True), self.event_type(enable_timing=True)
        try:
            with self._stream_ctx(st):
                t0.record(st)
                yield st
                t1.record(st)
        finally:
            w.view._lane_stream = None
        w.last = (t1, st)
        where = "chip" if (lane is w.full and w.full is not None) else "lane"
        self.stats.append((w.stage, w.k, where, t0, t1))
        for name, ev in getattr(w, "marks", None) or ():        # sub-phase marks a half recorded on this stream
            self.stats.append((f"{w.stage}.{name}", w.k, where, t0, ev))
        w.marks = None

    # ------------------------------------------------------------------ the two halves of main_run.edit_clip
    @staticmethod
    def _draw(gate, index, seed, shape, T):
        """Clip `index`'s T noise maps from the global CPU generator, when its turn comes (models.py:76-81 order)."""
        with gate.turn(index, seed):
            noise = torch.stack([torch.randn(shape, dtype=torch.float32) for _ in range(T)])
        return noise.pin_memory() if torch.cuda.is_available() else noise

    def _claim_noise(self, job, index, shape, T, helper=False):
        """Clip `index`'s T noise maps, drawn ONCE per job: whoever claims the index first -- the consumer itself or a helper
        thread running ahead -- draws them at the clip's turn; everybody else waits for that draw (two draws of one clip would
        both pass the gate at its turn and interleave on the global generator)."""
        with job["lock"]:
            box = job["prefetch"].get(index)
            mine = box is None
            if mine:
                box = job["prefetch"][index] = dict(shape=tuple(shape), T=T, ready=threading.Event())
        if mine:
            def work():
                try:
                    box["noise"] = self._draw(job["gate"], index, job["seeds"][index], box["shape"], T)
                except BaseException as e:                      # noqa: BLE001 -- re-raised by the consumer
                    box["error"] = e
                finally:
                    box["ready"].set()
            if helper:
                threading.Thread(target=work, name=f"aed-noise-{index}", daemon=True).start()
            else:
                work()
        return box

    def _gated_sample(self, view, job):
        """view.sample_xts_from_x0 with the clip's T draws taken from the global generator in clip order.  When every clip
        has the same shape, clip i+1's maps (40 ms of single-threaded CPU RNG for a 10 s clip) are drawn on a helper thread
        right after clip i's -- same generator, same order -- so they are ready when clip i+1 reaches this point."""

        def sample_xts_from_x0(x0, num_inference_steps=50):
            ed = view.editor(x0.shape[-2], x0.shape[-1])
            x = x0.reshape(1, *x0.shape[-3:])
            i, T = view._clip_index, int(num_inference_steps)
            box = self._claim_noise(job, i, x.shape, T)
            box["ready"].wait()
            if "error" in box:
                raise box["error"]
            assert box["shape"] == tuple(x.shape) and box["T"] == T, "uniform clips were promised (prefetched noise)"
            noise, box["noise"] = box["noise"], None            # the maps are consumed once; the claim stays
            view._clip_drew = True
            if job["uniform"] and i + 1 < len(job["items"]):
                self._claim_noise(job, i + 1, x.shape, T, helper=True)
            return ed.sample_xts(x, noise=noise)[:, 0]
        return sample_xts_from_x0

    def _front(self, w, st, job, i):
        """main_run.py:113-150: (waveform -> mel ->) VAE encode -> forward inversion."""
        v, a = w.view, job["a"]
        item = job["items"][i]
        ps = st if w.prep is None else w.prep.stream
        with self._stream_ctx(ps):
            x0 = job["prepare"](v, item) if job["prepare"] is not None else item
            w0 = v.vae_encode(x0)
            prepared = prepare_forward(v, w0, a["src"], a["cfg_src"], a["T"])
        if ps is not st:
            ready = self.event_type()
            ready.record(ps)
            st.wait_event(ready)
            conds = [getattr(c, n) for c in (prepared["cond_src"], prepared["cond_unc"]) if c is not None
                     for n in ("ehs0", "ehs1", "mask0", "mask1", "class_labels")]
            for t in (x0, w0, prepared["xts0"], *conds):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(st)
        setup = self.event_type(enable_timing=True)
        setup.record(st)                # everything before this on the lane is the clip's set-up (report(): front.setup_*)
        w.marks = [("setup", setup)]
        _, zs, wts, _ = run_forward(v, w0, prepared, a["eta"], a["cfg_src"], True, a["schedule"], a["group"])
        done = self.event_type()
        done.record(st)
        return dict(x0=x0, zs=zs, wts=wts, done=done)

    def _back(self, w, st, job, f, with_codec=True):
        """main_run.py:152-185: edit loop from x_tstart (-> VAE decode -> vocoder of edited + original unless a codec stage
        follows)."""
        v, a = w.view, job["a"]
        st.wait_event(f["done"])
        for t in (f["x0"], f["zs"], f["wts"]):
            if t.is_cuda:
                t.record_stream(st)
        tstart = a["tstart"]
        if w.wide is not None and st is w.lane.stream:
            def chooser():          # the widened lane once the front stage has issued AND finished its last inversion
                with job["lock"]:
                    ev = job["front_event"]
                    drained = job["stage_done"][0] and (ev is None or ev.query())
                if drained:
                    self.widened.add(w.k)
                return w.wide.stream if drained else None
            v._lane_chooser = chooser
        try:
            w_edit, _ = inversion_reverse_process(v, xT=f["wts"], tstart=torch.tensor([tstart], dtype=torch.int),
                                                  etas=a["eta"], prompts=a["tgt"], neg_prompts=a["neg"],
                                                  cfg_scales=a["cfg_tar"], zs=f["zs"][:tstart])
        finally:
            v._lane_chooser = None
        if not with_codec:
            edited = self.event_type()
            edited.record(st)
            return dict(x0=f["x0"], w_edit=w_edit, done=edited)
        # VAE decode + vocoder are throughput kernels (44 ms alone on 256 CUs, 71 ms on the 128-CU partition, 64 ms unmasked
        # beside a busy inversion partition: profiles/r03_codec_partition.md): they run unmasked, and the edit partition is
        # free for the next clip's set-up meanwhile
        cs = st
        if getattr(self, "codec_queue", "front") != "lane" and w.full is not None and w.full.stream is not st:
            cs = w.full.stream
            edited = self.event_type()
            edited.record(st)
            cs.wait_event(edited)
            for t in (w_edit, f["x0"]):
                if t.is_cuda:
                    t.record_stream(cs)
        with self._stream_ctx(cs):
            return self._decode(v, w_edit, f["x0"])

    @staticmethod
    def _decode(v, w_edit, x0):
        x0_dec = v.vae_decode(w_edit)
        if x0_dec.dim() < 4:
            x0_dec = x0_dec[None]
        audio = v.decode_to_mel(x0_dec)              # CPU tensors: the host blocks here until this clip is done
        orig = v.decode_to_mel(x0)                   # (moving this pass to the front stage's side stream measured +-0.4 %: round 6)
        return audio, orig, w_edit

    def _codec(self, w, st, job, e):
        """main_run.py:184-185 on the codec stage's own queue: edited latent -> mel -> waveform, original mel -> waveform."""
        st.wait_event(e["done"])
        for t in (e["w_edit"], e["x0"]):
            if t.is_cuda:
                t.record_stream(st)
        with self._capture_lock:                    # never issue into a stream another thread is capturing on (see __init__)
            return self._decode(w.view, e["w_edit"], e["x0"])

    # ------------------------------------------------------------------ workers
    def _pick_lane(self, w, job, stage_idx):
        """The worker's partition -- or the whole chip while no other stage has work (fill / drain)."""
        if w.full is None or len(self.stages) == 1:
            return w.lane
        with job["lock"]:
            if stage_idx == 0:
                idle = job["busy"][1] == 0 and job["queues"][1].qsize() == 0
            else:       # drain: the front stage has issued everything AND its last job has finished on the device
                ev = job["front_event"]
                idle = job["stage_done"][0] and (ev is None or ev.query())
        return w.full if idle else w.lane

    def _process(self, w, stage_idx, halves, job, i, payload, lane):
        """One clip through this worker's halves on `lane`."""
        guard = self._build_lock if not w.warm else contextlib.nullcontext()
        with guard, tape_mod.tile_regime(w.regime), torch.inference_mode(), self._on(w, lane) as st:
            if "front" in halves:
                payload = self._front(w, st, job, i)
            if "back" in halves:
                payload = self._back(w, st, job, payload, with_codec=not getattr(self, "codec_stage", False))
            if "codec" in halves:
                payload = self._codec(w, st, job, payload)
        w.warm = True
        return payload

    def _run_worker(self, w, stage_idx, halves, job):
        gate = job["gate"]
        v = w.view
        if "front" in halves:
            v.sample_xts_from_x0 = self._gated_sample(v, job)
        last_stage = stage_idx == len(self.stages) - 1
        try:
            while True:
                if stage_idx == 0:
                    with job["lock"]:
                        i = job["next"]
                        job["next"] += 1
                    if i >= len(job["items"]) or job["error"] is not None:
                        return
                    payload = None
                else:
                    got = job["queues"][stage_idx].get()
                    if got is _STOP or job["error"] is not None:
                        return
                    i, payload = got
                v._clip_index, v._clip_seed, v._clip_drew = i, job["seeds"][i], False
                with job["lock"]:
                    job["busy"][stage_idx] += 1
                t0 = time.perf_counter()
                try:
                    payload = self._process(w, stage_idx, halves, job, i, payload, self._pick_lane(w, job, stage_idx))
                except BaseException as e:                          # noqa: BLE001 -- reported by edit_clips
                    with job["lock"]:
                        if job["error"] is None:
                            job["error"] = (i, e)
                    if "front" in halves:
                        gate.done(i, failed=not v._clip_drew)
                    return
                finally:
                    with job["lock"]:
                        job["busy"][stage_idx] -= 1
                job["times"].append(dict(clip=i, stage=w.stage, worker=w.k, start=t0 - job["t0"],
                                         end=time.perf_counter() - job["t0"]))
                if last_stage:
                    job["out"][i] = payload
                else:
                    if stage_idx == 0:
                        job["front_event"] = payload["done"]
                    job["queues"][stage_idx + 1].put((i, payload))
        finally:
            v.__dict__.pop("sample_xts_from_x0", None)

    def _run_stage(self, stage_idx, job):
        """Start the stage's workers, wait for them, then tell the next stage that nothing more will come."""
        _, halves, ws = self.stages[stage_idx]
        ths = [threading.Thread(target=self._run_worker, args=(w, stage_idx, halves, job),
                                name=f"aed-{w.stage}-{w.k}", daemon=True) for w in ws]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        with job["lock"]:
            job["stage_done"][stage_idx] = True
        if stage_idx + 1 < len(self.stages):
            for _ in self.stages[stage_idx + 1][2]:
                job["queues"][stage_idx + 1].put(_STOP)

    # ------------------------------------------------------------------ driver
    def _job(self, items, seeds, prepare, a):
        K = len(items)
        seeds = [None] * K if seeds is None else list(seeds)
        if len(seeds) != K:
            raise ValueError("one seed (or None) per clip")
        shapes = {tuple(it.shape) if torch.is_tensor(it) else None for it in items}
        # noise prefetch needs the next clip's latent shape before that clip is prepared: only when all clips look alike,
        # and only with ONE front worker (several would each need the draw order of clips they have not reached yet)
        n_front = sum(len(ws) for _, halves, ws in self.stages if "front" in halves)
        uniform = len(shapes) == 1 and None not in shapes and n_front == 1
        return dict(items=list(items), seeds=seeds, prepare=prepare, a=a, out=[None] * K, next=0, uniform=uniform,
                    prefetch={}, lock=threading.Lock(), error=None, gate=_DrawGate(), t0=time.perf_counter(), times=[],
                    queues=[queue.Queue() for _ in self.stages], busy=[0] * len(self.stages),
                    stage_done=[False] * len(self.stages), front_event=None)

    def _args(self, source_prompt, target_prompt, target_neg_prompt, cfg_src, cfg_tar, T, tstart, eta):
        if len(source_prompt) != 1 or len(target_prompt) != 1:
            raise ValueError("ClipPipeline edits with one source and one target prompt")
        if isinstance(tstart, (list, tuple)):
            tstart = tstart[0]
        return dict(src=source_prompt, tgt=target_prompt, neg=target_neg_prompt, cfg_src=cfg_src, cfg_tar=cfg_tar, T=T,
                    tstart=int(tstart), eta=eta, schedule="batched", group=self.timestep_group)

    def edit_clips(self, items, source_prompt, target_prompt, target_neg_prompt, cfg_src, cfg_tar, T, tstart, eta=1.0,
                   prepare=None, seeds=None):
        """Edit `items` -- mels [1,1,T_mel,64], or anything `prepare(lane_view, item)` turns into one (e.g. waveforms
        through `lane_view.get_fn_STFT()`: the STFT engine then belongs to the lane like every other buffer) -- with
        main_run.edit_clip's arguments; clips enter the pipeline in order.  seeds[i]: `torch.manual_seed(seeds[i])` right
        before clip i's noise draws (a serial loop's per-clip seeding); None = the global generator simply continues
        from clip to clip.  Returns [(edited waveform, original-vocoded waveform, edited latent)] in input order."""
        job = self._job(items, seeds, prepare, self._args(source_prompt, target_prompt, target_neg_prompt, cfg_src,
                                                          cfg_tar, T, tstart, eta))
        self.stats, self._times = [], job["times"]
        self.widened = set()
        if not job["items"]:
            return []
        self._base = None
        if self.model.device.type == "cuda":
            self._base = self.event_type(enable_timing=True)     # origin of report()'s per-job timeline
            self._base.record(self.full.stream)
        ths = [threading.Thread(target=self._run_stage, args=(s, job), daemon=True) for s in range(len(self.stages))]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        if job["error"] is not None:
            i, e = job["error"]
            raise RuntimeError(f"clip {i} failed in the clip pipeline: {e!r}") from e
        missing = [i for i, o in enumerate(job["out"]) if o is None]
        if missing:
            raise RuntimeError(f"clips {missing} left the clip pipeline without an output")
        return job["out"]

    def warm_up(self, item, source_prompt, target_prompt, target_neg_prompt, cfg_src, cfg_tar, T, tstart, eta=1.0,
                prepare=None, seeds=None):
        """Build every worker's engines, loop plans and hipGraphs by pushing one clip through each worker, one worker at a
        time, on the caller's thread (engine construction folds shared weights lazily and is not meant to race)."""
        a = self._args(source_prompt, target_prompt, target_neg_prompt, cfg_src, cfg_tar, T, tstart, eta)
        seed = None if seeds is None else seeds[0]
        payload = None
        for s, (_, halves, ws) in enumerate(self.stages):
            outs = []
            for w in ws:
                job = self._job([item], [seed], prepare, a)
                v = w.view
                v._clip_index, v._clip_seed, v._clip_drew = 0, seed, False
                if "front" in halves:
                    v.sample_xts_from_x0 = self._gated_sample(v, job)
                try:
                    outs.append(self._process(w, s, halves, job, 0, payload, w.lane))
                finally:
                    v.__dict__.pop("sample_xts_from_x0", None)
            payload = outs[0]
        self.stats = []

    # ------------------------------------------------------------------ reporting
    def report(self):
        """Plan, partitions, and the average device time of each stage's jobs (HIP events on the lane streams) split by
        where they ran (their CU partition, or the whole chip during fill / drain)."""
        if self.model.device.type == "cuda":
            torch.cuda.synchronize(self.model.device)
        acc, timeline = {}, []
        base = getattr(self, "_base", None)
        for stage, k, where, t0, t1 in self.stats:
            acc.setdefault(f"{stage}_{where}", []).append(t0.elapsed_time(t1))
            if base is not None and "." not in stage:
                # device-side start / end of every job relative to the start of edit_clips: which queue was busy when
                timeline.append([f"{stage}{k}", where, round(base.elapsed_time(t0), 1), round(base.elapsed_time(t1), 1)])
        timeline.sort(key=lambda r: r[2])
        lat = {}
        for t in getattr(self, "_times", []):
            d = lat.setdefault(t["clip"], [t["start"], t["end"]])
            d[0], d[1] = min(d[0], t["start"]), max(d[1], t["end"])
        lats = [1e3 * (b - a) for a, b in lat.values()]
        return dict(plan=self.plan, launch=self.launch, clips_in_flight=self.clips_in_flight, total_cus=self.total,
                    edit_cus=self.edit_cus, edit_lanes=self.edit_lanes, edit_lane_cus=getattr(self, "edit_lane_cus", None),
                    inversion_cus=self.total - self.edit_cus, setup_stream_masked_to_inversion_partition=getattr(self, "mask_prep", False),
                    device_ms={k: dict(n=len(v), avg=sum(v) / len(v)) for k, v in acc.items()},
                    queue_separation=getattr(self, "queue_log", None), timeline=timeline,
                    widened_on_drain=sorted(getattr(self, "widened", ())),
                    drain_queue_separation=getattr(self, "drain_queue_log", None),
                    clip_latency_ms_avg=(sum(lats) / len(lats)) if lats else None,
                    clip_latency_ms_max=max(lats) if lats else None)

    def close(self):
        """Drain every lane (the streams themselves are process-lifetime objects, see streams.PartitionStream)."""
        for w in self.workers:
            for ps in (w.lane, w.full):
                if ps is not None:
                    ps.close()
        self.full.close()
