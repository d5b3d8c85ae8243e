"""Several clips in flight on one MI355X (BASELINE config 2 served as a stream of independent clips).

Why.  One clip of the reference path is a sequential chain of two regimes: the forward inversion
(inversion_utils.py:75-133; timestep-batched here: two U-Net calls at batch 200, throughput-bound: 0.39 of the bf16 MFMA peak
on the whole chip in the split-bf16 arithmetic, 0.49 alone on a 128-CU partition) and the 100-step edit loop (:221-315; ~570
dependent launches per step at U-Net batch 2, bound by launch / first-operand latency: 7.9 ms per step on 256 CUs, 13.0 ms on a
64-CU lane -- it cannot use the chip).  A clip's own arithmetic cannot be reordered, so the only work that can fill the idle
compute units is ANOTHER clip.  Since round 5 the steady state is bound by the chip's POWER budget, not by idle CUs
(profiles/r05_power_probe.md): the options below that only re-schedule work (edit_group, steal, codec_queue="chip", more lanes)
measure within 2 % of the default; they stay because a serving loop with other clip mixes may want them, each parity-tested.

Plans (measured on the MI355X, profiles/r03_cu_partition.md and profiles/r03_lanes.md):
  * "partition" (default): a two-stage pipeline on DISJOINT CU partitions (streams.PartitionStream: hardware queues with
    CU masks).  Stage "front" (one worker, CUs [edit_cus, total)): waveform -> mel -> VAE encode -> forward inversion.
    Stage "back" (edit_lanes workers sharing CUs [0, edit_cus)): edit loop -> VAE decode -> vocoder.  While clip i is in
    its edit loop, clip i+1 is being inverted; the two kernel classes never queue behind each other's workgroups (without
    masks a batch-2 kernel waits for 128x128-tile workgroups that hold a CU for 0.2-1 ms: 28.9 ms per edit step).
    Fill and drain (no other stage to share with) run on the whole chip.
  * "lanes": L workers, each running whole clips in the reference's step order on its own stream, unpartitioned.
    Measured: chip time per U-Net forward 8.6 -> 6.0 / 5.2 / 6.1 ms at L = 2 / 3 / 4 -- the batch-2 kernels' CU-time
    saturates the chip at ~1.6x, i.e. 1.6-1.9 s per clip: no better than one clip at a time with the batched inversion.
    Kept as a plan because it is the only one that needs no timestep regrouping.

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
DEFAULT_LANES = 3
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

    def __init__(self, model, plan="partition", edit_cus=None, edit_lanes=1, lanes=None, launch="graph",
                 timestep_group=100, overlap_prep=True, lane_cus=None, separate_queues=None, codec_stage=None,
                 widen_on_drain=True, edit_group=1, group_sizes=None, group_wait_s=0.0, codec_queue="front", steal=False,
                 steal_min_remaining=4):
        if getattr(model, "kind", None) == "stable_audio":
            raise NotImplementedError("ClipPipeline drives the mel-latent families (AudioLDM / AudioLDM2 / TANGO)")
        if plan not in ("partition", "lanes"):
            raise ValueError("plan must be 'partition' or 'lanes'")
        if launch not in ("eager", "graph"):
            raise ValueError("launch must be 'eager' or 'graph'")
        self.model, self.plan, self.launch = model, plan, launch
        self.timestep_group = int(timestep_group)
        # edit_group > 1 ("group plan", round 5): an edit lane steps up to `edit_group` clips IN LOCKSTEP -- one U-Net call per
        # diffusion step at batch 2g for the g clips whose inversions are ready when the lane becomes free (the per-rank shape
        # of BASELINE config 3, SURVEY 8e: "processed as one batch ... with cond+uncond stacked").  The batch-2 edit step is
        # latency-bound (572 launches of ~25 us whatever their size: profiles/r04_lane_perop_cus64.json); at batch 16 the same
        # lane runs 8 clip-steps in 3.5x the time of one (profiles/r05_batch_scaling.md).  Every clip's arithmetic is still its
        # own rows of the batch; against the clip edited alone the values agree to fp32 rounding (other tiles / split-K orders
        # per batch shape), not bit for bit -- tests and bench.py assert the tolerance instead of identity for this plan.
        self.edit_group = max(1, int(edit_group))
        if self.edit_group > 1 and plan != "partition":
            raise ValueError("edit_group > 1 needs the partition plan")
        sizes = sorted({int(g) for g in (group_sizes or [1 << k for k in range(self.edit_group.bit_length())])
                        if 1 <= int(g) <= self.edit_group} | {1})
        self.group_sizes = sizes                # U-Net batch 2g engines exist for exactly these g (built by warm_up)
        self.group_log = []                     # sizes of the groups the last edit_clips formed
        self.group_wait_s = float(group_wait_s)  # how long a free lane waits for a FULL group before taking what is ready
        # steal (round 5): an edit lane whose queue is empty takes the next UNSTARTED clip and runs its whole chain -- forward
        # inversion included -- on its own CUs.  Once the split-K tables made the batch-2 step 13 ms on a 64-CU lane, the lanes
        # (1.3 s per clip each) wait for the front stage (0.84 s per clip); a lane's idle 0.4 s per cycle cannot host another
        # edit loop, but summed over a run it is whole clips: front-fed clips cost a lane 1.3 s, a stolen one 2.7 s (two batch-200
        # forwards at ~0.7 s on 64 CUs), the optimum of that mix is +14 % over the front-bound rate.  A
        # lane's inversion engines are built under the FRONT stage's tile regime: same tiles, same split-K orders, so a clip's
        # values do not depend on who inverted it (bit-identical; tests/test_gpu_pipeline.py).  Not before `steal_min_remaining`
        # unstarted clips are left: the tail of a run belongs to the front stage (a stolen clip takes 2.7 s, the front's 0.84 s
        # + a widened edit loop).
        self.steal = bool(steal) and plan == "partition" and self.edit_group == 1
        self.steal_min_remaining = int(steal_min_remaining)
        self.stolen = []                        # clips the lanes inverted themselves in the last edit_clips
        dev = model.device
        acquire = getattr(self.lane_type, "acquire", None)

        def Lane(device, cus=None, total=None, index=0):           # process-lifetime streams (streams.PartitionStream)
            if acquire is not None:
                return acquire(device, cus=cus, total=total, index=index)
            return self.lane_type(device, cus=cus, total=total)
        self.full = Lane(dev)                                      # the whole chip
        self.total = self.full.total
        self.stages = []                                           # [(name, halves, [workers])]
        if plan == "partition":
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
            front = [_Worker("front", 0, self._view(), Lane(dev, cus=range(self.edit_cus, self.total), total=self.total),
                             self.full, regime=front_regime, prep=Lane(dev, index=17) if overlap_prep else None)]
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
            self.codec_stage = (n > 1 or self.edit_group > 1) if codec_stage is None else bool(codec_stage)
            if self.edit_group > 1 and not self.codec_stage:
                raise ValueError("the group plan (edit_group > 1) hands every edited latent to the codec stage")
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
                if self.edit_group > 1:
                    raise ValueError("the group plan hands its clips to a codec stage (codec_queue 'front' or 'chip')")
                self.codec_stage = False
            if self.codec_stage:
                front[0].prep = None
                # "chip" (round 5 A/B): the codec jobs on an UNMASKED queue of their own -- 44 ms of throughput kernels per clip
                # leave the inversion queue (the critical stage once the edit lanes got faster) and take whatever CU is free
                cq = front[0].lane if codec_queue == "front" else Lane(dev, index=70)
                self.stages.append(("codec", ("codec",), [_Worker("codec", 0, self._view(), cq, None)]))
            self.queue_log = []
            if separate_queues is None:
                separate_queues = n > 1 or self.edit_group > 1
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
            if widen_on_drain and self.edit_group > 1 and n == 1:
                # one group lane: when the front stage has drained, the running group loop continues on the lane's unmasked queue
                back[0].wide = back[0].full
        else:
            n = DEFAULT_LANES if lanes is None else int(lanes)
            if n < 1:
                raise ValueError("lanes must be >= 1")
            self.edit_cus, self.edit_lanes = None, n
            # lane_cus: every lane is a mini-chip of its own -- lane k owns CUs [k * lane_cus, (k + 1) * lane_cus) and runs whole
            # clips there with the timestep-batched inversion (the inversion's CU-time does not depend on the partition size, the
            # edit loop's falls with it: NOTES.md).  None: unmasked streams and the reference's step order (round 3's
            # measurement: saturates at 1.6x of one chain).
            self.lane_cus = None if lane_cus is None else int(lane_cus)
            if self.lane_cus is not None:
                if self.lane_cus % 32 or self.lane_cus < 32 or n * self.lane_cus > self.total:
                    raise ValueError(f"{n} lanes of {self.lane_cus} CUs: a lane must be a multiple of 32 CUs and all of them "
                                     f"fit the {self.total} CUs of the chip")
                if self.timestep_group < 2:
                    raise ValueError("masked lanes run the timestep-batched inversion (timestep_group >= 2)")
            regime = None
            for name, lo, hi in (("cus128", 96, 160), ("cus64", 24, 80)):
                if self.lane_cus is not None and lo <= self.lane_cus <= hi and name in tape_mod.REGIME_TABLES:
                    regime = name
            cus_of = (lambda k: None) if self.lane_cus is None else \
                (lambda k: range(k * self.lane_cus, (k + 1) * self.lane_cus))
            self.stages = [("clip", ("front", "back"),
                            [_Worker("clip", k, self._view(), Lane(dev, cus=cus_of(k), total=self.total, index=1 + k), None,
                                     regime=regime) for k in range(n)])]
            self.queue_log = []
            if separate_queues is None:
                separate_queues = self.lane_cus is not None and n > 1
            if separate_queues and hasattr(self.lane_type, "respin") and dev.type == "cuda":
                from .streams import separate_queues as _separate     # one dispatch pipe per busy lane (streams.py)
                ws = self.stages[0][2]
                for w, ps in zip(ws, _separate([w.lane for w in ws], log=self.queue_log)):
                    w.lane = ps
        self._build_lock = threading.Lock()     # an un-warmed worker builds engines (and lazily folds shared weights)
        # The codec worker issues on the FRONT lane's stream from its own thread.  hipStreamBeginCapture(ThreadLocal) does not keep
        # another thread's launches out of a capturing stream: a late graph capture of the front worker (a new loop shape after
        # warm-up, an LRU plan eviction) must not interleave with a codec job.  Every lane view carries this lock;
        # LoopPlumbing._run_graph holds it while capturing, _codec while issuing.
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
        thread running ahead -- draws them at the clip's turn; everybody else waits for that draw.  (With one front lane there
        is one claimant per clip anyway; with work stealing several lanes reach their draws concurrently, and two draws of one
        clip would both pass the gate at its turn and interleave on the global generator.)"""
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

    def _back_group(self, w, st, job, fs):
        """The edit loops of len(fs) clips in LOCKSTEP (inversion_utils.py:221-315 for each of them): one U-Net call per
        diffusion step over the rows [uncond x g | target x g] (editing.EditEngine.edit with n = g: the layout of BASELINE
        config 3's per-rank batch).  Returns one codec payload per clip."""
        v, a = w.view, job["a"]
        Z, g = int(a["tstart"]), len(fs)
        for f in fs:
            st.wait_event(f["done"])
            for t in (f["x0"], f["zs"], f["wts"]):
                if t.is_cuda:
                    t.record_stream(st)
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
            ed = v.editor(fs[0]["wts"].shape[-2], fs[0]["wts"].shape[-1])
            x_z = ed.to_nhwc(torch.stack([f["wts"][Z] for f in fs]))                  # x_tstart of every clip   [g, H, W, C]
            zs = ed.to_nhwc(torch.stack([f["zs"][:Z] for f in fs], 1))                # their noise maps         [Z, g, H, W, C]
            cond_tgt = conditioning_from_text(v, v.encode_text(a["tgt"])).repeat(g)
            cond_neg = conditioning_from_text(v, v.encode_text(a["neg"], negative=True))
            w_edit = ed.to_nchw(ed.edit(x_z.unsqueeze(0).expand(Z + 1, *x_z.shape), zs, Z, cond_tgt, cond_neg, a["cfg_tar"],
                                        eta=float(a["eta"])))
        finally:
            v._lane_chooser = None
        edited = self.event_type()
        edited.record(st)           # (_run_graph made st wait for whichever stream the last chunk of the loop ran on)
        return [dict(x0=f["x0"], w_edit=w_edit[k:k + 1], done=edited) for k, f in enumerate(fs)]

    @staticmethod
    def _decode(v, w_edit, x0):
        x0_dec = v.vae_decode(w_edit)
        if x0_dec.dim() < 4:
            x0_dec = x0_dec[None]
        audio = v.decode_to_mel(x0_dec)              # CPU tensors: the host blocks here until this clip is done
        orig = v.decode_to_mel(x0)
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
        if stage_idx == 0 and self.steal:
            return w.lane               # the lanes invert clips of their own from the first moment: no whole-chip fill
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
                if w.stage == "back":       # a stolen clip: the lane's inversion engines take the FRONT stage's tiles (same values)
                    with tape_mod.tile_regime(getattr(self, "_front_regime", None)):
                        payload = self._front(w, st, job, i)
                else:
                    payload = self._front(w, st, job, i)
            if "back" in halves:
                payload = self._back(w, st, job, payload, with_codec=not getattr(self, "codec_stage", False))
            if "codec" in halves:
                payload = self._codec(w, st, job, payload)
        w.warm = True
        return payload

    def _process_group(self, w, job, fs, lane):
        """A group of front-stage payloads through the back half on `lane` (group plan)."""
        guard = self._build_lock if not w.warm else contextlib.nullcontext()
        with guard, tape_mod.tile_regime(w.regime), torch.inference_mode(), self._on(w, lane) as st:
            out = self._back_group(w, st, job, fs)
        return out

    def _run_group_worker(self, w, stage_idx, job):
        """Back-stage worker of the group plan: takes every clip whose inversion is ready -- up to `edit_group`, rounded down to
        a size an engine exists for (`group_sizes`) -- and steps them in lockstep.  Greedy on purpose: with a slow front stage
        the groups stay small (latency), with a fast one they grow until the lane keeps up (throughput)."""
        q = job["queues"][stage_idx]
        pending, stopped = [], False
        while True:
            if not pending:
                if stopped:
                    return
                got = q.get()
                if got is _STOP:
                    return
                pending.append(got)
            deadline = time.perf_counter() + self.group_wait_s
            while len(pending) < self.edit_group and not stopped:
                try:
                    left = deadline - time.perf_counter()
                    got = q.get(timeout=left) if left > 0 else q.get_nowait()
                except queue.Empty:
                    break
                if got is _STOP:
                    stopped = True
                else:
                    pending.append(got)
            if job["error"] is not None:
                return
            g = max(s for s in self.group_sizes if s <= len(pending))
            batch, pending = pending[:g], pending[g:]
            self.group_log.append(g)
            with job["lock"]:
                job["busy"][stage_idx] += 1
            t0 = time.perf_counter()
            try:
                outs = self._process_group(w, job, [f for _, f in batch], self._pick_lane(w, job, stage_idx))
                w.warm = True
            except BaseException as e:                          # noqa: BLE001 -- reported by edit_clips
                with job["lock"]:
                    if job["error"] is None:
                        job["error"] = (batch[0][0], e)
                return
            finally:
                with job["lock"]:
                    job["busy"][stage_idx] -= 1
            t1 = time.perf_counter()
            for (i, _), payload in zip(batch, outs):
                job["times"].append(dict(clip=i, stage=w.stage, worker=w.k, start=t0 - job["t0"], end=t1 - job["t0"],
                                         group=g))
                job["queues"][stage_idx + 1].put((i, payload))

    def _run_worker(self, w, stage_idx, halves, job):
        if self.edit_group > 1 and halves == ("back",):
            return self._run_group_worker(w, stage_idx, job)
        gate = job["gate"]
        v = w.view
        stealing = self.steal and w.stage == "back"
        if "front" in halves or stealing:
            v.sample_xts_from_x0 = self._gated_sample(v, job)
        last_stage = stage_idx == len(self.stages) - 1
        try:
            while True:
                run = halves
                if stage_idx == 0:
                    with job["lock"]:
                        i = job["next"]
                        job["next"] += 1
                    if i >= len(job["items"]) or job["error"] is not None:
                        return
                    payload = None
                else:
                    got = None
                    while got is None:
                        try:
                            got = job["queues"][stage_idx].get(timeout=0.005 if stealing else None)
                        except queue.Empty:
                            with job["lock"]:       # nothing to edit: take an unstarted clip, unless the run's tail has begun
                                i = job["next"]
                                take = len(job["items"]) - i >= self.steal_min_remaining and job["error"] is None
                                if take:
                                    job["next"] += 1
                                    job["stealers"] += 1
                            if take:
                                got, run = (i, None), ("front",) + tuple(halves)
                                self.stolen.append(i)
                    if got is _STOP or job["error"] is not None:
                        return
                    ev = got[1].get("done") if (stealing and run == halves and isinstance(got[1], dict)) else None
                    if ev is not None and not ev.query():
                        # the queued clip's inversion is still running on the front partition (the front worker's HOST thread
                        # runs ahead of its queue): editing it now means waiting.  If no other lane is inverting a clip of its
                        # own, hand the payload back and take an unstarted clip instead.
                        with job["lock"]:
                            i2 = job["next"]
                            take = (len(job["items"]) - i2 >= self.steal_min_remaining and job["error"] is None
                                    and job["stealers"] == 0)
                            if take:
                                job["next"] += 1
                                job["stealers"] += 1
                        if take:
                            job["queues"][stage_idx].put(got)
                            got, run = (i2, None), ("front",) + tuple(halves)
                            self.stolen.append(i2)
                    i, payload = got
                v._clip_index, v._clip_seed, v._clip_drew = i, job["seeds"][i], False
                with job["lock"]:
                    job["busy"][stage_idx] += 1
                t0 = time.perf_counter()
                try:
                    payload = self._process(w, stage_idx, run, job, i, payload,
                                            w.lane if "front" in run and stage_idx else self._pick_lane(w, job, stage_idx))
                except BaseException as e:                          # noqa: BLE001 -- reported by edit_clips
                    with job["lock"]:
                        if job["error"] is None:
                            job["error"] = (i, e)
                    if "front" in run:
                        gate.done(i, failed=not v._clip_drew)
                    return
                finally:
                    with job["lock"]:
                        job["busy"][stage_idx] -= 1
                        if stage_idx and "front" in run:
                            job["stealers"] -= 1
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
                    stage_done=[False] * len(self.stages), front_event=None, stealers=0)

    def _args(self, source_prompt, target_prompt, target_neg_prompt, cfg_src, cfg_tar, T, tstart, eta):
        if len(source_prompt) != 1 or len(target_prompt) != 1:
            raise ValueError("ClipPipeline edits with one source and one target prompt")
        if isinstance(tstart, (list, tuple)):
            tstart = tstart[0]
        batched = self.plan == "partition" or getattr(self, "lane_cus", None) is not None
        return dict(src=source_prompt, tgt=target_prompt, neg=target_neg_prompt, cfg_src=cfg_src, cfg_tar=cfg_tar, T=T,
                    tstart=int(tstart), eta=eta, schedule="batched" if batched else "sequential",
                    group=self.timestep_group if batched else 1)

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
        self.group_log = []
        self.stolen = []
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
            if self.edit_group > 1 and halves == ("back",):
                # one engine + loop plan + step graph per group size (U-Net batch 2g); the clip is simply repeated
                for w in ws:
                    job = self._job([item], [seed], prepare, a)
                    for g in self.group_sizes:
                        outs = self._process_group(w, job, [payload] * g, w.lane)
                    w.warm = True
                payload = outs[0]
                continue
            for w in ws:
                job = self._job([item], [seed], prepare, a)
                v = w.view
                v._clip_index, v._clip_seed, v._clip_drew = 0, seed, False
                run = ("front",) + tuple(halves) if (self.steal and w.stage == "back") else halves
                if "front" in run:
                    v.sample_xts_from_x0 = self._gated_sample(v, job)
                try:
                    outs.append(self._process(w, s, run, job, 0, payload, w.lane))
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
                    edit_cus=self.edit_cus, edit_lanes=self.edit_lanes, edit_lane_cus=getattr(self, "edit_lane_cus", None), lane_cus=getattr(self, "lane_cus", None),
                    inversion_cus=None if self.edit_cus is None else self.total - self.edit_cus,
                    device_ms={k: dict(n=len(v), avg=sum(v) / len(v)) for k, v in acc.items()},
                    queue_separation=getattr(self, "queue_log", None), timeline=timeline,
                    widened_on_drain=sorted(getattr(self, "widened", ())),
                    edit_group=self.edit_group, group_sizes=self.group_sizes, groups_formed=list(self.group_log),
                    steal=self.steal, clips_inverted_by_edit_lanes=sorted(self.stolen),
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
