"""HIP streams restricted to a subset of the MI355X's compute units (include/aed.h: aed_stream_create_cu_mask).

Why: one edited clip is two regimes.  The forward inversion (inversion_utils.py:75-133) runs two
U-Net calls at batch 200 that fill every CU (0.39 of the bf16 MFMA peak in the split-bf16 arithmetic); the 100-step edit loop
(:221-315) is a chain of ~570 dependent launches per step at U-Net batch 2 that is latency-bound and leaves most CUs idle.  Two clips in flight
-- clip i in its edit loop, clip i+1 in its inversion -- use the idle CUs, but only if the two kernel classes do not
queue behind each other's workgroups: a 128x128-tile inversion workgroup holds its CU for 0.2-1 ms, a batch-2 kernel
lasts ~10 us.  So each class gets its own hardware queue with a DISJOINT CU mask.

Bit k of the mask is CU k in the driver's enumeration; consecutive bits rotate over the 8 XCDs, so a contiguous range of
8*m bits is m CUs on every XCD (each XCD keeps serving both partitions from its own L2).

Disjoint CUs are not enough (measured in round 4, tools/lane_interference.py): every hardware queue is served by one of the
command processor's few dispatch pipes, and a pipe launches ONE kernel's workgroups at a time.  Two busy queues on the same
pipe take turns -- an edit lane's 10 us kernels then wait for a batch-200 GEMM to finish handing out its ~3000 workgroups
(edit step 15.8 -> 47 ms although the CU sets were disjoint), while the same lanes on different pipes run undisturbed.
Which pipe a queue lands on is the driver's choice at creation, so `separate_queues` measures it: a long dispatch on one
stream, a chain of tiny kernels on the other, and a stream that is delayed is replaced by a fresh queue with the same mask.
"""
import ctypes
import time

import torch

from . import _lib as L


def cu_mask_words(bits, total):
    """32-bit mask words with the given CU bits set, sized for `total` CUs."""
    bits = sorted(set(int(b) for b in bits))
    if not bits or bits[0] < 0 or bits[-1] >= total:
        raise ValueError(f"CU bits {bits[:1]}..{bits[-1:]} outside [0, {total})")
    words = [0] * ((total + 31) // 32)
    for k in bits:
        words[k // 32] |= 1 << (k % 32)
    return words


class PartitionStream:
    """A hipStream_t owned by libaed.so (masked to the CU bits `cus`, or unmasked with a priority when `cus` is None),
    exposed as a torch stream so torch allocations, events and `wait_stream` work on it.

    Lifetime: `acquire()` hands out process-lifetime streams from a cache keyed by (device, CU set, index) and nothing
    destroys them.  torch's caching allocator tags blocks with the stream they were allocated / `record_stream`-ed on and
    touches that stream again when such a block is freed; a hipStreamDestroy in between (from an explicit close or a
    garbage-collected wrapper) crashed the GPU test process once.  A handful of queues per process is all a pipeline
    needs, so they are simply kept."""
    _cache = {}

    @classmethod
    def acquire(cls, device, cus=None, total=None, index=0):
        dev = torch.device(device)
        key = (str(dev), None if cus is None else tuple(sorted(set(int(b) for b in cus))), int(index))
        ps = cls._cache.get(key)
        if ps is None:
            ps = cls._cache[key] = cls(dev, cus=cus, total=total)
            ps.index = int(index)
        return ps

    def __init__(self, device, cus=None, total=None, priority=0):
        self.device = torch.device(device)
        lib = L.lib()
        if total is None:
            n_cu = ctypes.c_int()
            with torch.cuda.device(self.device):
                L.check(lib.aed_device_info(ctypes.byref(n_cu), None, None, 0), "aed_device_info")
            total = n_cu.value
        self.total = total
        self.cus = None if cus is None else sorted(set(int(b) for b in cus))
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            if cus is None:
                L.check(lib.aed_stream_create_cu_mask(ctypes.byref(handle), None, 0, int(priority)),
                        "aed_stream_create_cu_mask")
            else:
                words = cu_mask_words(self.cus, total)
                arr = (ctypes.c_uint32 * len(words))(*words)
                L.check(lib.aed_stream_create_cu_mask(ctypes.byref(handle), arr, len(words), 0),
                        "aed_stream_create_cu_mask")
        self.handle = handle
        self.stream = torch.cuda.ExternalStream(handle.value, device=self.device)

    def respin(self, k):
        """Another hardware queue with this stream's CU mask (the driver may place it on a different dispatch pipe)."""
        # keyed by the ORIGINAL queue's index as well: two lanes that share a CU mask (index 0 and 1) must not respin into the
        # same cached queue (they would silently serialise on one hardware queue)
        return type(self).acquire(self.device, cus=self.cus, total=self.total,
                                  index=1000 + 16 * (getattr(self, "index", 0) % 1000) + int(k))

    def census(self, n_blocks=2048, spin_clocks=200000):
        """Physical CUs this stream's workgroups land on: sorted list of (xcc, se, sh, cu)."""
        out = torch.zeros(2 * n_blocks, dtype=torch.int32, device=self.device)
        L.check(L.lib().aed_cu_census(out.data_ptr(), n_blocks, spin_clocks, ctypes.c_void_p(self.stream.cuda_stream)),
                "aed_cu_census")
        self.stream.synchronize()
        v = out.cpu().view(n_blocks, 2).tolist()
        return sorted({(x & 0xF, (h >> 13) & 0x7, (h >> 12) & 1, (h >> 8) & 0xF) for h, x in v})

    def close(self, destroy=False):
        """Drain the stream.  `destroy=True` also calls aed_stream_destroy -- only safe when no tensor was ever allocated
        or `record_stream`-ed on it (see the class docstring); cached streams are never destroyed."""
        if self.handle is not None:
            self.stream.synchronize()
            if destroy and self not in self._cache.values():
                L.check(L.lib().aed_stream_destroy(self.handle), "aed_stream_destroy")
                self.handle = None


def _census_launch(ps, buf, n_blocks, spin):
    L.check(L.lib().aed_cu_census(buf.data_ptr(), int(n_blocks), int(spin), ctypes.c_void_p(ps.stream.cuda_stream)),
            "aed_cu_census")


def queue_delay(busy, victim, chain=40, rounds=3):
    """How much longer a chain of `chain` tiny kernels on `victim` takes while `busy` dispatches long grids (ratio of wall
    times, >= ~1).  ~1: the two hardware queues are served independently; several: they share a dispatch pipe."""
    dev = victim.device
    nb = len(busy.cus) if busy.cus is not None else busy.total
    nv = len(victim.cus) if victim.cus is not None else victim.total
    big = torch.zeros(2 * 64 * nb, dtype=torch.int32, device=dev)
    small = torch.zeros(2 * nv, dtype=torch.int32, device=dev)

    def tiny():
        for _ in range(chain):
            _census_launch(victim, small, nv, 2000)

    def timed(with_busy):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if with_busy:
            for _ in range(rounds):                      # 64 blocks per CU of ~0.17 ms each: the pipe is held ~0.3 ms per grid
                _census_launch(busy, big, 64 * nb, 400000)
        tiny()
        victim.stream.synchronize()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        return dt
    timed(False)
    alone = min(timed(False) for _ in range(2))
    together = min(timed(True) for _ in range(2))
    return together / max(alone, 1e-6)


def separate_queues(streams, threshold=2.0, attempts=6, log=None):
    """Given PartitionStreams that will be busy AT THE SAME TIME (first = highest priority to keep), return a list in which
    each later stream has been replaced -- by fresh queues with the same CU mask, `PartitionStream.respin` -- until no earlier
    stream's dispatches delay it (`queue_delay` < threshold) or `attempts` run out (then the least-delayed candidate is
    kept).  Streams that are replaced stay alive (process-lifetime cache) but idle."""
    out = []
    for ps in streams:
        best, best_d = ps, None
        cand = ps
        for k in range(attempts):
            d = max([queue_delay(o, cand) for o in out] + [queue_delay(cand, o) for o in out] + [1.0])
            if log is not None:
                log.append(dict(cus=None if cand.cus is None else (cand.cus[0], cand.cus[-1] + 1), attempt=k, delay=round(d, 2)))
            if best_d is None or d < best_d:
                best, best_d = cand, d
            if d < threshold:
                break
            cand = ps.respin(k + 1)
        out.append(best)
    return out
