"""ctypes binding of libaed.so (include/aed.h).  The product path fails loudly when the HIP
library is missing -- there is no CPU fallback anywhere in this package."""
import ctypes
import os

import torch  # noqa: F401  -- MUST be imported before libaed.so is dlopen'ed: the library has to bind to the
#                               HIP runtime torch already loaded (a second libamdhip64 in the process sees no device)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaed.so")


class aed_op(ctypes.Structure):
    _fields_ = [("code", ctypes.c_int32), ("flags", ctypes.c_int32), ("i", ctypes.c_int32 * 40),
                ("f", ctypes.c_float * 8), ("p", ctypes.c_void_p * 10)]


# opcodes (include/aed.h enum aed_opcode)
OP_NOP, OP_CONV_GEMM, OP_GN_STATS, OP_GN_APPLY, OP_LAYERNORM, OP_ATTENTION, OP_GEGLU, OP_COPY2D, \
    OP_TIME_EMBED, OP_SOFTMAX_ROWS, OP_TRANSPOSE, OP_AXPBY, OP_INVERT_STEP, OP_REVERSE_STEP, OP_DDIM_STEP, \
    OP_ADVANCE, OP_REFLECT_PAD, OP_MAGNITUDE, OP_NCHW_TO_NHWC, OP_NHWC_TO_NCHW, OP_SPLITK_REDUCE, \
    OP_GN_SCALE_SHIFT, OP_GN_SMALL, OP_XATTN_FOLD, OP_ROTARY, OP_SNAKE, OP_SA_STEP, OP_GAUSS_SAMPLE = range(28)
OP_NAMES = ["nop", "conv_gemm", "gn_stats", "gn_apply", "layernorm", "attention", "geglu", "copy2d", "time_embed",
            "softmax_rows", "transpose", "axpby", "invert_step", "reverse_step", "ddim_step", "advance",
            "reflect_pad", "magnitude", "nchw_to_nhwc", "nhwc_to_nchw", "splitk_reduce", "gn_scale_shift", "gn_small",
            "xattn_fold", "rotary", "snake", "sa_step", "gauss_sample"]
ACT_NONE, ACT_SILU, ACT_LEAKY, ACT_TANH, ACT_LOGCLAMP = range(5)
COEF_STRIDE = 8
SA_COEF_STRIDE = 12

_lib = None


class AedError(RuntimeError):
    pass


def lib():
    """Load libaed.so once; raise if it is not built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AedError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                           f"(audioeditingcode_amd/csrc/build.sh). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        L.aed_version.restype = ci
        L.aed_last_error.restype = ctypes.c_char_p
        L.aed_device_info.argtypes = [ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.c_char_p, ci]
        L.aed_launch.argtypes = [ctypes.POINTER(aed_op), vp]
        L.aed_tape_run.argtypes = [ctypes.POINTER(aed_op), ci, vp]
        L.aed_tape_profile.argtypes = [ctypes.POINTER(aed_op), ci, vp, ctypes.POINTER(cf)]
        L.aed_graph_begin.argtypes = [vp]
        L.aed_graph_end.argtypes = [vp, ctypes.POINTER(vp)]
        L.aed_graph_launch.argtypes = [vp, vp]
        L.aed_graph_destroy.argtypes = [vp]
        L.aed_stream_create_cu_mask.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_uint32), ci, ci]
        L.aed_stream_destroy.argtypes = [vp]
        L.aed_cu_census.argtypes = [vp, ci, ci, vp]
        u64 = ctypes.c_uint64
        L.aed_image_load.argtypes = [ctypes.c_char_p, ci, ctypes.POINTER(vp)]
        L.aed_image_free.argtypes = [vp]
        L.aed_image_run.argtypes = [vp, ctypes.c_char_p, vp]
        L.aed_image_program.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.POINTER(aed_op)), ctypes.POINTER(ci)]
        L.aed_image_buffer.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(vp), ctypes.POINTER(u64)]
        L.aed_image_copy_in.argtypes = [vp, ctypes.c_char_p, vp, u64, vp]
        L.aed_image_copy_out.argtypes = [vp, ctypes.c_char_p, vp, u64, vp]
        L.aed_event_create.argtypes = [ctypes.POINTER(vp)]
        L.aed_event_record.argtypes = [vp, vp]
        L.aed_event_elapsed_ms.argtypes = [vp, vp, ctypes.POINTER(cf)]
        L.aed_event_destroy.argtypes = [vp]
        fp = ctypes.POINTER(cf)
        L.aed_get_zs_from_xts.argtypes = [vp, vp, vp, vp, vp, cf, ci, fp, ci, ci, vp, vp, ctypes.c_int64, vp]
        L.aed_reverse_step_with_custom_noise.argtypes = [vp, vp, vp, vp, cf, ci, fp, ci, vp, vp, ctypes.c_int64, vp]
        L.aed_sample_xts_from_x0.argtypes = [vp, vp, vp, vp, vp, ci, ctypes.c_int64, vp]
        L.aed_mx_quantize_rows.argtypes = [vp, vp, vp, ctypes.c_longlong, ci, vp]
        L.aed_sa_get_zs_from_xts.argtypes = [vp, vp, vp, vp, cf, fp, vp, ci, vp, vp, ctypes.c_int64, vp]
        L.aed_sa_reverse_step_with_custom_noise.argtypes = [vp, vp, vp, cf, fp, vp, vp, vp, ctypes.c_int64, vp]
        for name in ("aed_launch", "aed_tape_run", "aed_tape_profile", "aed_graph_begin", "aed_graph_end",
                     "aed_graph_launch", "aed_graph_destroy", "aed_stream_create_cu_mask", "aed_stream_destroy",
                     "aed_cu_census", "aed_image_load", "aed_image_free", "aed_image_run", "aed_image_program",
                     "aed_image_buffer", "aed_image_copy_in", "aed_image_copy_out", "aed_event_create", "aed_event_record", "aed_event_elapsed_ms", "aed_event_destroy", "aed_get_zs_from_xts",
                     "aed_reverse_step_with_custom_noise", "aed_sample_xts_from_x0", "aed_device_info",
                     "aed_sa_get_zs_from_xts", "aed_sa_reverse_step_with_custom_noise", "aed_mx_quantize_rows"):
            getattr(L, name).restype = ci
        if L.aed_version() != 4:
            raise AedError("libaed.so ABI version mismatch")
        _lib = L
    return _lib


EXPORTS = ["aed_version", "aed_last_error", "aed_device_info", "aed_launch", "aed_tape_run", "aed_tape_profile",
           "aed_graph_begin", "aed_graph_end", "aed_graph_launch", "aed_graph_destroy", "aed_stream_create_cu_mask",
           "aed_stream_destroy", "aed_cu_census", "aed_image_load", "aed_image_free", "aed_image_run", "aed_image_program",
           "aed_image_buffer", "aed_image_copy_in", "aed_image_copy_out", "aed_event_create",
           "aed_event_record", "aed_event_elapsed_ms", "aed_event_destroy", "aed_get_zs_from_xts",
           "aed_reverse_step_with_custom_noise", "aed_sample_xts_from_x0", "aed_sa_get_zs_from_xts",
           "aed_sa_reverse_step_with_custom_noise", "aed_mx_quantize_rows"]


def check(rc, what=""):
    if rc != 0:
        raise AedError(f"{what} failed (rc={rc}): {lib().aed_last_error().decode()}")


def current_stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
