"""Tape images: compile a model forward ahead of time into one relocatable file (include/aed.h: aed_image_*).

The graph compiler of this package is Python (unet.py, codec.py, stable_audio.py lay out op tapes over torch-allocated
buffers).  A host that has no Python -- the boundary SURVEY 8(b) sketched as aed_create / aed_unet_forward / ... -- does not
need it at run time: `export_image` walks the finished tapes, assigns every buffer they reference (weights, frequency /
index tables, activations, inputs, outputs) an offset in ONE arena, rewrites the ops' raw device pointers as arena offsets
and writes ops + a snapshot of the arena.  `aed_image_load` (C) allocates the arena, uploads the snapshot, relocates the
pointers; `aed_image_run(image, "forward", stream)` is then `aed_tape_run` on those ops.

    eng = UNetEngine(cfg, weights, "cuda:0", 2, 256, 16, ctx_len0=8, ctx_len1=16, timesteps_dev=ts, state_dev=state)
    export_image("unet_b2.aedimg", {"context": eng.ctx_tape, "forward": eng.tape},
                 {"x_in": eng.x_in, "eps": eng.eps, "ehs0": eng.ehs0, "ehs1": eng.ehs1, "bias1": eng.bias1,
                  "timesteps": ts, "state": state})

File layout (little endian): header {magic "AEDIMG1", version 1, n_ops, n_programs, n_names, arena_bytes, snapshot_bytes} |
programs {name[48], first op, op count} | named buffers {name[48], offset, bytes} | ops (struct aed_op, p[k] = arena offset or
~0 for NULL) | arena[0, snapshot_bytes).  Buffers listed in `scratch` (activations) are placed after the snapshot and start
out zeroed, so an image of the AudioLDM2 U-Net at batch 2 is ~1.4 GB of weights, not weights + activations.
"""
import bisect
import ctypes
import struct

import torch

from . import _lib as L

ALIGN = 256
NULL = (1 << 64) - 1


def _storage_key(t):
    st = t.untyped_storage()
    return st.data_ptr(), st.nbytes()


def _storage_bytes(t):
    """Raw bytes of the storage behind `t` (device tensors are fetched to the host)."""
    st = t.untyped_storage()
    flat = torch.empty(0, dtype=torch.uint8, device=t.device).set_(st)
    return flat.cpu().numpy().tobytes()


def export_image(path, programs, names, scratch=()):
    """programs: {name: Tape} (executed in file order by name lookup); names: {name: tensor} buffers a host addresses by
    name; scratch: tensors whose contents need not travel (they are zero after aed_image_load).  Returns a summary dict."""
    tapes = list(programs.items())
    tensors = [t for _, tp in tapes for t in tp.keep if torch.is_tensor(t)]
    tensors += [tp.ws for _, tp in tapes if tp.ws is not None]
    tensors += list(names.values())
    stor = {}
    for t in tensors:
        ptr, nb = _storage_key(t)
        if nb and ptr not in stor:
            stor[ptr] = (nb, t)
    scratch_ptrs = {_storage_key(t)[0] for t in scratch}
    order = sorted(stor, key=lambda p: (p in scratch_ptrs, p))          # snapshot part first, scratch part last
    offset, cur, snapshot_bytes = {}, 0, 0
    for ptr in order:
        cur = -(-cur // ALIGN) * ALIGN
        offset[ptr] = cur
        cur += stor[ptr][0]
        if ptr not in scratch_ptrs:
            snapshot_bytes = cur
    arena_bytes = -(-cur // ALIGN) * ALIGN
    starts = sorted(stor)

    def relocate(p, what):
        if not p:
            return NULL
        k = bisect.bisect_right(starts, p) - 1
        if k < 0 or p >= starts[k] + stor[starts[k]][0]:
            raise ValueError(f"{what}: pointer {p:#x} is not inside any buffer the tape keeps alive")
        return offset[starts[k]] + (p - starts[k])

    ops_blob, prog_rows, n_ops = bytearray(), [], 0
    for name, tp in tapes:
        arr = tp.finalize()
        prog_rows.append((name, n_ops, len(tp.ops)))
        for k in range(len(tp.ops)):
            op = L.aed_op()
            ctypes.memmove(ctypes.byref(op), ctypes.byref(arr[k]), ctypes.sizeof(L.aed_op))
            for j in range(10):
                if j == 7 and not (op.flags & 1):
                    op.p[j] = NULL                      # p7 is the optional in-kernel timeline buffer (flag bit 0)
                    continue
                op.p[j] = relocate(op.p[j] or 0, f"program {name!r} op {k} ({tp.meta[k]['name']}) p{j}")
            ops_blob += bytes(op)
        n_ops += len(tp.ops)
    name_rows = []
    for nm, t in names.items():
        name_rows.append((nm, relocate(t.data_ptr(), f"named buffer {nm!r}"), t.numel() * t.element_size()))
    enc = lambda s_: s_.encode()[:47].ljust(48, b"\0")              # noqa: E731
    with open(path, "wb") as f:
        f.write(struct.pack("<8sIIIIQQ", b"AEDIMG1\0", 1, n_ops, len(prog_rows), len(name_rows), arena_bytes, snapshot_bytes))
        for nm, first, cnt in prog_rows:
            f.write(enc(nm) + struct.pack("<II", first, cnt))
        for nm, off, nb in name_rows:
            f.write(enc(nm) + struct.pack("<QQ", off, nb))
        f.write(ops_blob)
        pos = 0
        for ptr in order:
            if ptr in scratch_ptrs:
                break
            f.write(b"\0" * (offset[ptr] - pos))
            data = _storage_bytes(stor[ptr][1])
            f.write(data)
            pos = offset[ptr] + len(data)
        f.write(b"\0" * (snapshot_bytes - pos))
    return dict(path=path, n_ops=n_ops, programs=[r[0] for r in prog_rows], names=[r[0] for r in name_rows],
                arena_bytes=arena_bytes, snapshot_bytes=snapshot_bytes, buffers=len(order))


class Image:
    """ctypes handle on a loaded tape image (what a C / C++ host does with the same seven entry points)."""

    def __init__(self, path, host=False):
        self.h = ctypes.c_void_p()
        L.check(L.lib().aed_image_load(str(path).encode(), 1 if host else 0, ctypes.byref(self.h)), "aed_image_load")

    def run(self, program, stream_ptr=None):
        L.check(L.lib().aed_image_run(self.h, program.encode(), stream_ptr), "aed_image_run")

    def buffer(self, name):
        p, n = ctypes.c_void_p(), ctypes.c_uint64()
        L.check(L.lib().aed_image_buffer(self.h, name.encode(), ctypes.byref(p), ctypes.byref(n)), "aed_image_buffer")
        return p.value, n.value

    def program(self, name):
        ops, n = ctypes.POINTER(L.aed_op)(), ctypes.c_int()
        L.check(L.lib().aed_image_program(self.h, name.encode(), ctypes.byref(ops), ctypes.byref(n)), "aed_image_program")
        return ops, n.value

    def copy_in(self, name, host_tensor, stream_ptr=None):
        t = host_tensor.contiguous()
        L.check(L.lib().aed_image_copy_in(self.h, name.encode(), ctypes.c_void_p(t.data_ptr()),
                                          t.numel() * t.element_size(), stream_ptr), "aed_image_copy_in")

    def copy_out(self, name, host_tensor, stream_ptr=None):
        assert host_tensor.is_contiguous()
        L.check(L.lib().aed_image_copy_out(self.h, name.encode(), ctypes.c_void_p(host_tensor.data_ptr()),
                                           host_tensor.numel() * host_tensor.element_size(), stream_ptr),
                "aed_image_copy_out")
        return host_tensor

    def close(self):
        if self.h:
            L.check(L.lib().aed_image_free(self.h), "aed_image_free")
            self.h = ctypes.c_void_p()
