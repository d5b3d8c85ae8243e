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


# ------------------------------------------------------------------------------------------------ whole engines / models
def engine_image_spec(eng):
    """(programs, names, scratch) of one engine of this package, by its class: the buffers a host fills and reads."""
    kind = type(eng).__name__
    if kind == "UNetEngine":
        names = {k: getattr(eng, k) for k in ("x_in", "eps", "h_space", "ehs0", "ehs1", "bias0", "bias1", "class_labels")
                 if getattr(eng, k, None) is not None}
        if eng.timesteps_dev is not None:
            names["timesteps"] = eng.timesteps_dev          # int64 table read by the time-embedding op at index state[0]
        if eng.state_dev is not None:
            names["state"] = eng.state_dev
        return {"context": eng.ctx_tape, "forward": eng.tape}, names, [eng.eps, eng.h_space]
    if kind == "STFTEngine":
        return {"forward": eng.tape}, {"wav": eng.wav, "mel": eng.mel}, [eng.mel, eng.mag]
    if kind == "VAEEncoder":
        return {"forward": eng.tape}, {"x_in": eng.x_in, "latent": eng.latent}, [eng.latent]
    if kind == "VAEDecoder":
        return {"forward": eng.tape}, {"z_in": eng.z_in, "mel": eng.mel}, [eng.mel]
    if kind == "VocoderEngine":
        return {"forward": eng.tape}, {"mel_in": eng.mel_in, "wav": eng.wav}, [eng.wav]
    raise TypeError(f"no image spec for {kind}")


def export_engine(path, eng):
    programs, names, scratch = engine_image_spec(eng)
    return export_image(path, programs, names, scratch)


def export_model_images(model, out_dir, n_samples=163840, unet_batch=2, ctx_len1=16):
    """The five engines of one mel-latent wrapper (AudioLDM / AudioLDM2 / TANGO) for a clip of `n_samples` at the model's
    rate, as tape images: stft.aedimg, vae_encode.aedimg, unet_b<B>.aedimg, vae_decode.aedimg, vocoder.aedimg -- what a
    host needs besides the loop-level step entry points (aed_get_zs_from_xts, aed_reverse_step_with_custom_noise) to run
    the whole path of main_run.py:104-185 without Python.  Returns {file: summary}."""
    import os
    os.makedirs(out_dir, exist_ok=True)
    stft = model._stft(1, n_samples)
    T = stft.frames - 1                                 # utils.pad_spec cuts the STFT's surplus last frame
    if T % 4:
        raise ValueError(f"n_samples={n_samples}: {T} mel frames is not a multiple of the VAE's time stride (4)")
    F = stft.n_mels
    enc = model._vae_enc(1, T, F)
    dec = model._vae_dec(1, enc.h, enc.w)
    voc = model._vocoder(1, T)
    ed = model.editor(enc.h, enc.w)
    fam = model.family["ctx"]
    L0 = fam.get("gpt2_len", 0) if model.kind == "audioldm2" else (ctx_len1 if model.kind == "tango" else 0)
    unet = ed.unet(unet_batch, L0, ctx_len1 if model.kind == "audioldm2" else 0)
    out = {}
    for name, eng in (("stft", stft), ("vae_encode", enc), (f"unet_b{unet_batch}", unet), ("vae_decode", dec),
                      ("vocoder", voc)):
        out[name + ".aedimg"] = export_engine(os.path.join(out_dir, name + ".aedimg"), eng)
    return out


def main(argv=None):
    import argparse
    p = argparse.ArgumentParser(description="Export the engines of one model as tape images (include/aed.h: aed_image_*)")
    p.add_argument("--model_id", default="cvssp/audioldm2")
    p.add_argument("--out", default="images")
    p.add_argument("--num_diffusion_steps", type=int, default=200)
    p.add_argument("--seconds", type=float, default=10.24)
    p.add_argument("--device_num", type=int, default=0)
    p.add_argument("--allow_synthetic", action="store_true")
    a = p.parse_args(argv)
    from .models import load_model
    m = load_model(a.model_id, f"cuda:{a.device_num}", a.num_diffusion_steps, allow_synthetic=a.allow_synthetic or None)
    for f, info in export_model_images(m, a.out, n_samples=int(a.seconds * m.get_sr())).items():
        print(f"{f}: {info['n_ops']} ops, programs {info['programs']}, buffers {info['names']}, "
              f"{info['snapshot_bytes'] / 1e6:.1f} MB snapshot + {(info['arena_bytes'] - info['snapshot_bytes']) / 1e6:.1f} MB scratch")


if __name__ == "__main__":
    main()
