"""Several clips in flight on one MI355X: L independent edit lanes (BASELINE config 2 served as a stream of clips).

Why.  One clip of the reference path is a strictly sequential chain -- 200 inversion steps + 100 edit steps
(inversion_utils.py:75-133, :221-315), each one U-Net forward at batch 2 (uncond | cond) = ~600 dependent kernel launches
of 5-30 us.  That chain is bound by launch / first-operand latency, not by arithmetic: measured on the MI355X
(profiles/r03_lanes.md) one chain keeps the chip ~25 % busy (8.6 ms per step), while 2 / 3 / 4 INDEPENDENT chains on
separate HIP streams run at 10.3 / 11.9 / 11.9 ms per step each, i.e. 5.1 / 4.0 / 3.0 ms of chip time per clip-step.  The
idle resource is compute units, and the only work that can use them without touching a clip's arithmetic is ANOTHER
clip.  (A CU-partitioned two-clip variant -- edit loop of clip i beside the batch-200 inversion of clip i+1 on disjoint
CU masks, streams.PartitionStream -- was measured first and is slower: both partitions become CU-time bound,
profiles/r03_cu_partition.md.)

What.  `ClipPipeline` owns L lanes = (HIP stream, lane view of the wrapper, host thread).  A lane view shares the frozen
weights / scheduler / text encoders of the wrapper and owns every mutable buffer (U-Net, VAE, vocoder and STFT engines,
loop plans, hipGraphs).  Each lane thread pulls the next clip and runs the UNCHANGED per-clip path on its stream --
main_run.edit_clip: mel -> VAE encode -> forward inversion -> edit loop -> VAE decode -> vocoder -- in the reference's
step order by default (`schedule="sequential"`: no timestep regrouping at all).  Every clip's launches, values and
results are those of the serial run, bit for bit (tests/test_gpu_pipeline.py); only the interleaving on the GPU differs.

RNG.  The reference draws a clip's T noise maps from torch's global CPU generator (models.py:76-81).  Lanes keep that
stream and its order: clip i draws (after an optional `torch.manual_seed(seeds[i])`) only when clips 0..i-1 have drawn
(`_DrawGate`), so the noise every clip sees is what a serial loop over the clips would have produced.
"""
import contextlib
import threading
import time

import torch

from .main_run import edit_clip

DEFAULT_LANES = 4


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


class ClipPipeline:
    def __init__(self, model, lanes=None, launch="eager"):
        if getattr(model, "kind", None) == "stable_audio":
            raise NotImplementedError("ClipPipeline drives the mel-latent families (AudioLDM / AudioLDM2 / TANGO)")
        self.model = model
        self.n_lanes = DEFAULT_LANES if lanes is None else int(lanes)
        if self.n_lanes < 1:
            raise ValueError("lanes must be >= 1")
        if launch not in ("eager", "graph"):
            raise ValueError("launch must be 'eager' or 'graph'")
        # How a lane issues one diffusion step.  "graph": one hipGraphLaunch per step (what the single-clip loops do).
        # "eager": the step's ~610 launches are issued one by one from C++ (aed_tape_run, GIL released).  On the device the
        # two are equivalent (same kernels, same dependent-launch boundary cost); on the host a hipGraphLaunch of a
        # 600-node graph costs milliseconds and the runtime serialises such launches across threads, which starves
        # concurrent lanes (measured: profiles/r03_lanes.md) -- eager launches from one thread per lane do not.
        self.launch = launch
        self.views = [model.lane_view() for _ in range(self.n_lanes)]
        for v in self.views:
            v._lane_eager = launch == "eager"
        self.streams = [self._new_stream(model.device) for _ in range(self.n_lanes)]
        self._build_lock = threading.Lock()      # first clip of a lane: engines / plans / lazily folded weights are built
        self._warm = [False] * self.n_lanes
        self.stats = []

    # the two HIP touch points (the CPU host-logic tests substitute stand-ins; the product uses HIP streams)
    @staticmethod
    def _new_stream(device):
        return torch.cuda.Stream(device=device)

    @staticmethod
    def _stream_ctx(stream):
        return torch.cuda.stream(stream)

    # ------------------------------------------------------------------ one lane
    def _gated_sample(self, view, gate):
        """view.sample_xts_from_x0 with the clip's T draws taken from the global generator in clip order."""
        def sample_xts_from_x0(x0, num_inference_steps=50):
            ed = view.editor(x0.shape[-2], x0.shape[-1])
            x = x0.reshape(1, *x0.shape[-3:])
            with gate.turn(view._clip_index, view._clip_seed):
                noise = torch.stack([torch.randn(x.shape, dtype=torch.float32) for _ in range(num_inference_steps)])
            view._clip_drew = True
            return ed.sample_xts(x, noise=noise)[:, 0]
        return sample_xts_from_x0

    def _lane(self, k, job):
        view, gate = self.views[k], job["gate"]
        view.sample_xts_from_x0 = self._gated_sample(view, gate)
        try:
            with self._stream_ctx(self.streams[k]):
                while True:
                    with job["lock"]:
                        i = job["next"]
                        job["next"] += 1
                    if i >= len(job["items"]) or job["error"] is not None:
                        return
                    view._clip_index, view._clip_seed, view._clip_drew = i, job["seeds"][i], False
                    t0 = time.perf_counter()
                    try:
                        guard = self._build_lock if not self._warm[k] else contextlib.nullcontext()
                        with guard:
                            item = job["items"][i]
                            with torch.inference_mode():
                                x0 = job["prepare"](view, item) if job["prepare"] is not None else item
                            job["out"][i] = edit_clip(view, x0, *job["args"], **job["kwargs"])
                            self._warm[k] = True
                    except BaseException as e:                      # noqa: BLE001 -- reported by edit_clips
                        with job["lock"]:
                            if job["error"] is None:
                                job["error"] = (i, e)
                        gate.done(i, failed=not view._clip_drew)
                        return
                    self.stats.append(dict(clip=i, lane=k, start=t0 - job["t0"], end=time.perf_counter() - job["t0"]))
        finally:
            view.__dict__.pop("sample_xts_from_x0", None)

    def warm_up(self, item, *args, **kwargs):
        """Build every lane's engines, loop plans and hipGraphs by running one clip per lane, one lane at a time (engine
        construction folds shared weights lazily and is not meant to race).  Same arguments as edit_clips, one item."""
        for k in range(self.n_lanes):
            if not self._warm[k]:
                only = ClipPipeline.__new__(ClipPipeline)
                only.__dict__.update(self.__dict__)
                only.n_lanes, only.views, only.streams, only._warm = 1, [self.views[k]], [self.streams[k]], [False]
                only.edit_clips([item], *args, **kwargs)
                self._warm[k] = True

    # ------------------------------------------------------------------ driver
    def edit_clips(self, items, source_prompt, target_prompt, target_neg_prompt, cfg_src, cfg_tar, T, tstart, eta=1.0,
                   schedule="sequential", timestep_group=8, prepare=None, seeds=None, **edit_clip_kwargs):
        """Edit `items` -- mels [1,1,T_mel,64], or anything `prepare(lane_view, item)` turns into one (e.g. waveforms
        through `lane_view.get_fn_STFT()`: the STFT engine then belongs to the lane like every other buffer) -- with
        main_run.edit_clip's arguments; clips are handed to the lanes in order.  seeds[i]: `torch.manual_seed(seeds[i])` right before clip i's noise draws (a serial loop's
        per-clip seeding); None = the global generator simply continues from clip to clip.
        Returns [(edited waveform, original-vocoded waveform, edited latent)] in input order."""
        if schedule != "sequential" and self.n_lanes > 1:
            # the timestep-batched inversion holds ~0.75 GB of activations per U-Net batch row (150 GB at batch 200):
            # one engine per lane does not fit, and lanes at batch 2 already fill the chip
            raise ValueError("lanes run the reference's step order (schedule='sequential'); the timestep-batched "
                             "inversion is the single-clip latency mode of main_run.edit_clip")
        K = len(items)
        seeds = [None] * K if seeds is None else list(seeds)
        if len(seeds) != K:
            raise ValueError("one seed (or None) per clip")
        job = dict(items=list(items), seeds=seeds, prepare=prepare, out=[None] * K, next=0, lock=threading.Lock(),
                   error=None, gate=_DrawGate(), t0=time.perf_counter(),
                   args=(source_prompt, target_prompt, target_neg_prompt, cfg_src, cfg_tar, T, tstart),
                   kwargs=dict(eta=eta, schedule=schedule, timestep_group=timestep_group, **edit_clip_kwargs))
        self.stats = []
        threads = [threading.Thread(target=self._lane, args=(k, job), name=f"aed-lane-{k}", daemon=True)
                   for k in range(min(self.n_lanes, max(K, 1)))]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        if job["error"] is not None:
            i, e = job["error"]
            raise RuntimeError(f"clip {i} failed in the clip pipeline: {e!r}") from e
        return job["out"]

    def lane_report(self):
        """Per-lane clip counts and the clip latency (host wall, start of its lane slot -> waveforms on the host)."""
        if not self.stats:
            return {}
        lat = [s["end"] - s["start"] for s in self.stats]
        per_lane = {}
        for s in self.stats:
            per_lane[s["lane"]] = per_lane.get(s["lane"], 0) + 1
        return dict(lanes=self.n_lanes, clips=len(self.stats), clips_per_lane=[per_lane.get(k, 0) for k in range(self.n_lanes)],
                    clip_latency_ms_avg=1e3 * sum(lat) / len(lat), clip_latency_ms_max=1e3 * max(lat))
