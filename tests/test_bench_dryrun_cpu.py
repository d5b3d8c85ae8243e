"""bench.py's control flow executed on CPU with every GPU-facing piece mocked (model, engines, streams): guards the
driver-facing contract -- ONE JSON line with the required keys, both inversion schedules, the roofline and traffic
legs, --clips-per-gpu -- against run-time errors that would otherwise only show up on the GPU box."""
import contextlib
import io
import json
import os
import sys
from unittest import mock

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                     # noqa: E402
from audioeditingcode_amd import main_run, models               # noqa: E402


class _Op:
    def __init__(self, flags=0, tile=1, variant=0):
        self.flags = flags
        self.i = [0] * 40
        self.i[29], self.i[14] = tile, variant


class _Tape:
    flops = 3.4e11
    exec_flops = 3.0e11
    # one split-bf16 GEMM, one latency-regime fp32 GEMM, one GroupNorm
    meta = [dict(code=1, flops=1e9, exec_flops=8e8, bytes=1e6, name="x.conv1"),
            dict(code=1, flops=1e8, exec_flops=1e8, bytes=1e5, name="x.qkv"),
            dict(code=22, flops=0, exec_flops=0, bytes=4e6, name="gn")]
    ops = [_Op(flags=12, tile=8), _Op(flags=0, tile=10), _Op()]

    def profile(self):
        return [0.01, 0.005, 0.005]

    def capture(self):
        return None

    def replay(self):
        return None


class _Event:
    def __init__(self, *a, **k):
        pass

    def record(self, *a):
        pass

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return 12.0

    def query(self):
        return True


class _Eng:
    tape = _Tape()


class _Ed:
    def __init__(self, batches):
        self._unets = {(b, 8, 16): _Eng() for b in batches}
        self._unets[(batches[0], 8, 9)] = _Eng()            # a second engine of the same batch size (other context length)
        self.state = torch.zeros(4, dtype=torch.int32)
        self._plans = {"p": dict(state=torch.zeros(4, dtype=torch.int32))}

    def edit_latents(self, w0, *a, **k):
        return torch.ones(w0.shape[0], 8, 256, 16)


class _STFT:
    def mel_spectrogram(self, w):
        return torch.zeros(1, 64, 1025), None, None


class _Model:
    weights_source, state_dicts, kind, device = "mock", {}, "audioldm2", "cpu"

    def __init__(self, batches):
        self._ed = _Ed(batches)
        self._editors, self._engines = {}, {}

    def lane_view(self):
        return _Model([2])

    def get_fn_STFT(self):
        return _STFT()

    def editor(self, H, W):
        return self._ed

    def vae_encode(self, x):
        return torch.zeros(1, 8, 256, 16)

    def vae_decode(self, w):
        return torch.zeros(1, 1, 1024, 64)

    def decode_to_mel(self, x):
        return torch.zeros(1, 16)

    def encode_text(self, p, negative=False):
        return torch.zeros(1, 8, 768), torch.zeros(1, 4, 1024), torch.ones(1, 4)


class _Stream:
    def __init__(self, *a, **k):
        pass


@contextlib.contextmanager
def _stream_ctx(s):
    yield


class _W:
    def __init__(self, stage, view):
        self.stage, self.view = stage, view
        self.lane = type("Lane", (), dict(stream=_Stream()))()
        self.full = self.lane


class _Pipe:
    """pipeline.ClipPipeline stand-in: workers with lane views / streams, clips handed back in order."""
    made = []

    def __init__(self, model, plan="partition", edit_cus=None, edit_lanes=1, launch="graph", timestep_group=100,
                 overlap_prep=True, codec_queue="front", mask_prep=True):
        assert launch in ("eager", "graph") and plan == "partition"
        self.plan, self.total, self.edit_cus, self.edit_lanes = plan, 256, edit_cus, edit_lanes
        self.edit_lane_cus, self.codec_stage, self.codec_queue, self.mask_prep = edit_cus, False, codec_queue, mask_prep
        self.workers = [_W("front", _Model([2 * timestep_group]))] + [_W("back", _Model([2])) for _ in range(edit_lanes)]
        self.clips_in_flight = len(self.workers)
        self.calls = []
        _Pipe.made.append(self)

    def warm_up(self, item, *a, **k):
        self.calls.append(("warm_up", 1))

    def edit_clips(self, items, *a, prepare=None, seeds=None, **k):
        assert len(seeds) == len(items) and prepare is not None
        assert prepare(self.workers[0].view, items[0]).shape == (1, 1, 1024, 64)          # prepare(lane view, item) -> mel
        self.calls.append(("edit_clips", len(items)))
        return [(None, None, torch.ones(1, 8, 256, 16)) for _ in items]

    def report(self):
        return dict(plan=self.plan, clips_in_flight=self.clips_in_flight, edit_cus=self.edit_cus)


def _run(argv, batches, extra_patches=()):
    from audioeditingcode_amd import pipeline
    with contextlib.ExitStack() as stack:
        for pt in extra_patches:
            stack.enter_context(pt)
        stack.enter_context(mock.patch.object(pipeline, "ClipPipeline", _Pipe))
        return _run_inner(argv, batches)


def _run_inner(argv, batches):
    with mock.patch.object(sys, "argv", ["bench.py", *argv]), mock.patch("torch.cuda.set_device"), \
            mock.patch("torch.cuda.synchronize"), mock.patch("torch.cuda.Stream", _Stream), \
            mock.patch("torch.cuda.stream", _stream_ctx), mock.patch("torch.cuda.Event", _Event), \
            mock.patch.object(models, "load_model", lambda *a, **k: _Model(batches)), \
            mock.patch.object(bench, "check_fractions", lambda roof: None), \
            mock.patch.object(main_run, "edit_clip", lambda m, x0, *a, **k: (None, None, torch.ones(1, 8, 256, 16))), \
            mock.patch("audioeditingcode_amd.weights.random_state_dict", lambda *a, **k: {}), \
            mock.patch("torch.Tensor.to", lambda self, *a, **k: self), mock.patch("torch.device", lambda *a, **k: "cpu"):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
    lines = [ln for ln in buf.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                    # exactly ONE line on stdout
    return json.loads(lines[0])


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def test_default_line_has_the_contract_keys():
    """Default mode: the two-stage partition pipeline + the two one-clip-at-a-time legs + the roofline measured per stage."""
    _Pipe.made.clear()
    out = _run(["--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-extras", "--group", "20", "--edit-cus", "96",
                "--edit-lanes", "2", "--serial-clips", "2"], [2, 40])
    assert all(k in out for k in REQUIRED)
    assert out["n_gpus"] == 1 and out["steps"] == 4 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["dtype"] == "f32" and out["vs_baseline"] is None and out["higher_is_better"] is True
    assert "workload" in out["config"] and out["config"]["clips_per_gpu_per_step"] == 1
    assert out["config"]["clips_in_flight_per_gpu"] == 3 and "96 CUs" in out["config"]["workload"]
    # workers built, W warm, K timed, then the compared clips one per call through the same engines
    assert _Pipe.made[0].calls == [("warm_up", 1), ("edit_clips", 2), ("edit_clips", 4), ("edit_clips", 1), ("edit_clips", 1)]
    r = out["roofline"]
    # the dominant family: the split-bf16 GEMM of the batch-40 inversion forward, executed MFMA flops over the bf16 MFMA peak
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] <= 1
    assert abs(r["achieved"] - 6 * 8e8 / (0.01e-3) / 1e12) < 1e-9 and abs(r["achieved_fp32_equiv"] * 6 - r["achieved"]) < 1e-9
    assert r["forward"]["unet_batch"] == 40 and set(r["forward"]["families"]) == {"gemm_bf16x6", "gemm_f32", "groupnorm"}
    assert r["forward"]["families"]["gemm_f32"]["instruction_peak_tflops"] == 157.3
    assert abs(r["on_partition"]["cu_fraction_of_chip"] - 160 / 256) < 1e-12
    assert abs(r["on_partition"]["peak"] - 2500.0 * 160 / 256) < 1e-9
    assert set(r["edit_step"]) == {"clips_1"} and r["edit_step"]["clips_1"]["unet_batch"] == 2
    assert r["traffic"] is None or r["traffic"] > 0
    assert 0 < r["path"]["matrix_pipe_frac"] and "fp32_equiv_over_fp32_mfma_peak" in r["path"]
    assert "value_reference_order" in out and "value_single_clip_batched" in out and out["serial_legs_clips"] == 2
    assert out["schedule_deviation_rel_l2"] == 0.0          # the mocked edit returns the same latent for every schedule
    assert out["pipeline_vs_one_clip_at_a_time"] == dict(
        clips_compared=2, bit_identical_to_same_engines_alone=True, max_abs_diff=0.0, rel_l2_vs_plain_serial_leg=0.0,
        plain_serial_schedule="batched")


def test_serial_plan_single_clip_schedules_and_multi_clip_mode():
    _Pipe.made.clear()
    out = _run(["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--plan", "serial", "--schedule",
                "sequential", "--group", "20"], [2, 40])
    assert "value_single_clip_batched" in out and "reference order" in out["config"]["workload"]
    assert out["roofline"]["forward"]["unet_batch"] == 2
    out = _run(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--plan", "serial", "--group", "20"],
               [2, 40])
    assert out["roofline"]["forward"]["unet_batch"] == 40
    out = _run(["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--clips-per-gpu", "2", "--group",
                "20", "--no-batched"], [4, 40])
    assert out["config"]["clips_per_gpu_per_step"] == 2 and out["config"]["gathered_latents"] == [[2, 8, 256, 16]]
    assert out["roofline"]["forward"]["unet_batch"] == 40


def test_extras_are_reported_and_never_fatal():
    """parity + sub-benchmarks: a sub-process that prints a JSON line is embedded, one that fails is recorded."""
    import subprocess

    def fake_run(cmd, **kw):
        if "bench_config4.py" in " ".join(cmd):
            return subprocess.CompletedProcess(cmd, 3, stdout="", stderr="boom")
        return subprocess.CompletedProcess(cmd, 0, stdout='noise\n{"metric": "m", "value": 2.5, "unit": "u", "roofline": '
                                                          '{"frac": 0.5, "by_batch": {}}}\n', stderr="")
    out = _run(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--group", "20"], [2, 40],
               extra_patches=[mock.patch.object(bench, "parity_leg", lambda *a, **k: dict(latent_rel_l2=1e-4)),
                              mock.patch.object(bench, "parity_fixture_leg", lambda *a, **k: dict(latent_rel_l2=2e-5)),
                              mock.patch.object(bench.subprocess, "run", fake_run)])
    assert out["parity"] == dict(latent_rel_l2=1e-4, parity_T200=dict(latent_rel_l2=2e-5))      # live T=8 leg + fixture leg
    assert out["config3_per_rank"]["value"] == 2.5 and out["config3_per_rank"]["roofline"] == {"frac": 0.5}
    assert out["config4_pc_extract_apply"]["failed"] == "rc=3"
    assert out["config5_stable_audio"]["value"] == 2.5


def test_roofline_fraction_check_refuses_fractions_above_their_own_roof():
    """check_fractions (round 5): EVERY `frac` of the roofline object -- at any depth -- is executed work over the peak of the
    instruction that did it and must lie in (0, 1]: round 3's path_frac = 4.23 and round 4's `frac` > 1 under bf16x6 both raise.
    The comparison with an ideal fp32-MFMA implementation is a ratio under another name (`fp32_equiv_over_fp32_mfma_peak`)."""
    import pytest
    ok = dict(frac=0.35, fp32_equiv_over_fp32_mfma_peak=0.93, forward=dict(families=dict(gemm_bf16x6=dict(frac=0.35), groupnorm=dict(frac=0.6))),
              on_partition=dict(frac=0.41, cu_fraction_of_chip=0.625), path=dict(matrix_pipe_frac=0.3, fp32_equiv_over_fp32_mfma_peak=1.2))
    bench.check_fractions(ok)
    with pytest.raises(AssertionError, match="roofline.frac"):
        bench.check_fractions(dict(ok, frac=1.21))
    with pytest.raises(AssertionError, match="gemm_bf16x6.frac"):
        bench.check_fractions(dict(ok, forward=dict(families=dict(gemm_bf16x6=dict(frac=4.23)))))
    with pytest.raises(AssertionError, match="matrix_pipe_frac"):
        bench.check_fractions(dict(ok, path=dict(matrix_pipe_frac=0.0)))


def test_family_table_prices_streaming_ops_against_the_chips_hbm_and_marks_cache_resident_traffic():
    """family_table (round 5): an HBM-bound family is priced against the WHOLE chip's HBM bandwidth whatever the CU partition (a CU
    mask does not partition memory); a producer -> consumer copy served from the 256 MB Infinity Cache moves its algorithmic
    bytes faster than HBM could: no fraction is stated for it (`served_from_cache`) instead of a `frac` above 1; MFMA families
    scale their peak with the partition."""
    from types import SimpleNamespace as NS
    ops = [NS(flags=4, i=[0] * 40), NS(flags=0, i=[0] * 40), NS(flags=0, i=[0] * 40)]
    meta = [dict(code=1, exec_flops=1e12, bytes=1e6), dict(code=22, exec_flops=0.0, bytes=2e9), dict(code=7, exec_flops=0.0, bytes=9e9)]
    eng = NS(tape=NS(ops=ops, meta=meta))
    t = bench.family_table(eng, [10.0, 1.0, 1.0], cu_frac=0.5)
    assert abs(t["gemm_bf16x6"]["achieved_tflops"] - 600.0) < 1e-6 and abs(t["gemm_bf16x6"]["frac"] - 600.0 / 1250.0) < 1e-9
    assert t["groupnorm"]["hbm_peak_gb_per_s"] == bench.PEAK_HBM_GBS and abs(t["groupnorm"]["frac"] - 2000.0 / bench.PEAK_HBM_GBS) < 1e-9
    assert t["elementwise"].get("served_from_cache") is True and "frac" not in t["elementwise"]
    bench.check_fractions(dict(families=t))
