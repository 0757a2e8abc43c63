"""ctypes binding of libposegan_hip.so (include/posegan_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails, a RuntimeError is
raised.  PyTorch tensors are used only as device storage: every call passes raw ``data_ptr()``s and
the current HIP stream.
"""
import ctypes as C
import os

import torch

from . import build as _build

PG_MAX_SRC = 4
STAT_SLOTS = 16          # include/posegan_hip.h PG_STAT_SLOTS
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
OUT_NONE, OUT_TANH = 0, 1

_f32p = C.c_void_p


class Src(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int32), ("_pad", C.c_int32), ("aff", C.c_void_p),
                ("mask", C.c_void_p), ("sN", C.c_int64), ("sC", C.c_int64), ("sH", C.c_int64), ("sW", C.c_int64)]


class Dst(C.Structure):
    _fields_ = [("grad", C.c_void_p), ("fwd", C.c_void_p), ("aff", C.c_void_p), ("mask", C.c_void_p),
                ("C", C.c_int32), ("act", C.c_int32), ("accumulate", C.c_int32), ("flags", C.c_int32),
                ("bsums", C.c_void_p)]


class ConvDesc(C.Structure):
    _fields_ = [("src", Src * PG_MAX_SRC),
                ("nsrc", C.c_int32), ("N", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32),
                ("act", C.c_int32), ("scalar_in", C.c_int32),
                ("mode", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32),
                ("pad", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
                ("w_transposed", C.c_int32),
                ("W", C.c_void_p), ("wCout", C.c_int32), ("wCin", C.c_int32),
                ("n_off", C.c_int32), ("n_cnt", C.c_int32),
                ("epilogue", C.c_int32), ("out_act", C.c_int32),
                ("out", C.c_void_p), ("bias", C.c_void_p),
                ("oN", C.c_int64), ("oC", C.c_int64), ("oH", C.c_int64), ("oW", C.c_int64),
                ("dst", Dst * PG_MAX_SRC),
                ("ndst", C.c_int32), ("ksplit", C.c_int32), ("precision", C.c_int32), ("out_bf16", C.c_int32),
                ("stats", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64)]


class WgradDesc(C.Structure):
    _fields_ = [("src", Src * PG_MAX_SRC),
                ("nsrc", C.c_int32), ("N", C.c_int32), ("act", C.c_int32), ("scalar_x", C.c_int32),
                ("dY", C.c_void_p),
                ("yN", C.c_int64), ("yC", C.c_int64), ("yH", C.c_int64), ("yW", C.c_int64),
                ("scalar_y", C.c_int32), ("x_is_large", C.c_int32),
                ("Hs", C.c_int32), ("Ws", C.c_int32), ("Hl", C.c_int32), ("Wl", C.c_int32),
                ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("dW", C.c_void_p), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("ksplit", C.c_int32), ("cout_store", C.c_int32)]


_i32, _i64, _u64, _f32, _vp = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p
_PROTOS = {
    "pg_conv": [C.POINTER(ConvDesc), _vp],
    "pg_conv_wgrad": [C.POINTER(WgradDesc), _vp],
    "pg_tap_gather": [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i64, _i64, _i64, _i64, _vp],
    "pg_im2col_taps": [_vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "pg_small_cout_dgrad": [_vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, C.POINTER(Dst), _i32, _vp],
    "pg_out_conv_dgrad": [_vp, _vp, _i32, _i32, _i32, C.POINTER(Dst), _i32, _vp],
    "pg_repack_small_cin": [_vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "pg_small_cin_conv": [C.POINTER(Src), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "pg_small_cin_wgrad": [C.POINTER(Src), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp],
    "pg_add2": [_vp, _vp, _vp, _i64, _vp],
    "pg_counter_add": [_vp, _u64, _vp],
    "pg_dropout_mask_ctr": [_vp, _i64, _u64, _f32, _vp, _vp],
    "pg_adam_ctr": [_vp, _vp, _vp, _vp, _vp, _i64, C.c_double, C.c_double, _f32, _f32, _i64, _vp, _f32, _vp, _vp],
    "pg_small_cin_dgrad": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _i64, _i64, _i64, _vp],
    "pg_small_cin_dgrad_io": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _i64, _i64, _i64, _i32, _vp],
    "pg_stem_pack_elems": [_i32, _i32],
    "pg_stem_pack_bf16": [_vp, _i32, _i32, _vp, _vp],
    "pg_stem_conv_bf16": [C.POINTER(Src), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "pg_stem_conv_bf16_ex": [C.POINTER(Src), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp],
    "pg_stem_wgrad_bf16": [C.POINTER(Src), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp],
    "pg_bias_grad": [_vp, _i64, _i64, _i32, _i64, _i64, _i64, _vp, _vp],
    "pg_cords_to_map": [_vp, _i32, _i32, _i32, _i32, C.c_float, _vp, _i64, _i64, _i64, _i64, _vp],
    "pg_affine_transforms": [_vp, _vp, _i32, _i32, _vp, _vp],
    "pg_uniform_transform": [_vp, _vp, _i32, _i32, _vp, _vp],
    "pg_pose_masks": [_vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "pg_preprocess_image": [_vp, _i32, _i32, _i32, _vp, _i64, _i64, _i64, _i64, _vp],
    "pg_materialise_bf16": [_vp, _vp, _vp, _i32, _i32, _i64, _i32, _vp, _vp],
    "pg_weights_to_bf16": [_vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "pg_channel_major_bf16": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _vp, _vp],
    "pg_wgrad_bf16": [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp],
    "pg_wgrad_bf16_ex": [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp],
    "pg_gemm_taps_bf16": [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "pg_norm_stats": [_vp, _i32, _i64, _vp, _vp],
    "pg_norm_finalize": [_vp, _vp, _vp, _i32, _i64, _f32, _vp, _vp, _vp],
    "pg_norm_bwd_reduce": [_vp, _vp, _vp, _i32, _i64, _vp, _vp],
    "pg_norm_bwd_apply": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp],
    "pg_norm_bwd_apply_ex": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp],
    "pg_mask_pyramid": [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "pg_warp_mask_max_fwd": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "pg_warp_mask_max_bwd": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "pg_gan_logloss": [_vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp],
    "pg_l1_loss": [_vp, _vp, _i64, _f32, _vp, _vp, _i32, _vp],
    "pg_tanh_bwd": [_vp, _vp, _i64, _vp],
    "pg_vgg_conv1_relu_fwd": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp],
    "pg_vgg_conv1_dgrad": [_vp, _vp, _i32, _i32, _i32, _vp, _vp],
    "pg_nn_loss": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp],
    "pg_adam": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _vp],
    "pg_adam_ex": [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _vp],
    "pg_comm_unique_id": [_vp],
    "pg_comm_init": [_vp, _i32, _i32, C.POINTER(_vp)],
    "pg_comm_allreduce_bucket": [_vp, _vp, _i64, _i32, _vp],
    "pg_comm_destroy": [_vp],
    "pg_comm_ranks": [_vp, C.POINTER(_i32), C.POINTER(_i32)],
    "pg_pack_bf16": [_vp, _vp, _i64, _vp],
    "pg_dropout_mask": [_vp, _i64, _u64, _f32, _vp],
    "pg_nchw_to_nhwc": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "pg_nhwc_to_nchw": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "pg_apply_affine_act": [_vp, _vp, _vp, _i32, _i32, _i64, _i32, _vp, _vp],
    "pg_event_create": [C.POINTER(_vp)],
    "pg_event_record": [_vp, _vp],
    "pg_event_elapsed_ms": [_vp, _vp, C.POINTER(_f32)],
    "pg_event_destroy": [_vp],
    "pg_debug_spin": [_i32, _vp],
    "pg_weights_to_bf16_batch": [_vp, _vp, _i32, _i32, _vp, _vp],
    "pg_tape_begin": [],
    "pg_tape_end": [C.POINTER(_vp), C.POINTER(_i64)],
    "pg_tape_replay": [_vp],
    "pg_tape_destroy": [_vp],
    "pg_stream_wait": [_vp, _vp],
    "pg_zero": [_vp, _i64, _vp],
    "pg_copy": [_vp, _vp, _i64, _vp],
    "pg_transpose_f32": [_vp, _i32, _i32, _vp, _i32, _vp],
    "pg_materialise_bf16_ex": [_vp, _i32, _vp, _vp, _i32, _i32, _i64, _i32, _vp, _vp, _i32, _vp],
    "pg_norm_bwd_reduce_ex": [_vp, _vp, _vp, _i32, _i64, _vp, _i32, _vp],
    "pg_materialise_bf16_norm": [_vp, _i32, _vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _vp, _vp, _i32, _vp],
    "pg_norm_bwd_apply_io": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _i32, _vp],
    "pg_norm_bwd_apply_v2": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _i32, _i32, _vp],
    "pg_norm_bwd_reduce_guard": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _i32, _vp],
    "pg_norm_bwd_apply_v3": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp],
    "pg_warp_mask_max_fwd_io": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp],
    "pg_warp_mask_max_bwd_io": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp],
    "pg_debug_conv_timeline": [_vp, _i32],
    "pg_debug_warp_gather_overflows": [_vp],
    "pg_mask_bbox": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "pg_warp_mask_max_bwd_bbox": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp],
    "pg_stem_conv_bf16_v3": [C.POINTER(Src), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp],
    "pg_stem_wgrad_bf16_ex": [C.POINTER(Src), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i64, _vp],
    "pg_stem_wgrad_bf16_v2": [C.POINTER(Src), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _i64, _vp],
    "pg_bias_grad_bf16": [_vp, _i64, _i32, _vp, _vp],
    "pg_out_conv_fwd_fused": [_vp, _i32, _vp, _vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "pg_tap_gather_pitch": [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i64, _i64, _i64, _i64, _vp],
    "pg_im2col_taps_bf16": [_vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "pg_out_conv_wgrad_bf16": [_vp, _i32, _i32, _i32, _i32, C.POINTER(Dst), _i32, _vp, _vp, _i64, _vp],
    "pg_out_conv_bwd_direct": [_vp, _i32, _vp, _i32, _i32, _i32, C.POINTER(Dst), _i32, _vp, _vp, _i64, _vp, _vp],
    "pg_out_conv_dgrad_wgrad": [_vp, _vp, _i32, _i32, _i32, C.POINTER(Dst), _i32, _vp, _vp, _i64, _vp],
    "pg_version": [],
    "pg_set_deterministic": [_i32],
    "pg_get_deterministic": [],
    "pg_last_launch_info": [],
}
EXPORTS = sorted(list(_PROTOS) + ["pg_last_error"])

_lib = None


def lib_path():
    return _build.LIB


def load():
    """dlopen the in-tree library; fail loudly if it is not there (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError("libposegan_hip.so not found at %s — run `python __graft_entry__.py` "
                           "(build()) first; this package has no CPU fallback" % path)
    lib = C.CDLL(path)
    for name, args in _PROTOS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.pg_stem_pack_elems.restype = C.c_int64
    lib.pg_last_error.argtypes = []
    lib.pg_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, load().pg_last_error().decode()))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """torch's CURRENT stream of the current device as a raw hipStream_t (every kernel of the library is enqueued on it,
    so torch ops, RCCL collectives and torch.cuda events order against them).  torch.cuda.current_stream() builds a
    Python Stream object through several layers (8 us; 240 calls per iteration = 2 ms of host time), the raw getter is
    one C call."""
    if _raw_stream is not None and _cur_device is not None:
        return C.c_void_p(_raw_stream(_cur_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


CALL_HOOK = None      # bench.py's per-kernel timing leg: callable(name, args, launch) wrapping every named entry-point call


def call(name, *args):
    if CALL_HOOK is not None:
        return CALL_HOOK(name, args, lambda: check(getattr(load(), name)(*args), name))
    check(getattr(load(), name)(*args), name)


def make_src(t, C_, aff=None, mask=None, strides=None):
    s = Src()
    s.ptr = ptr(t)
    s.C = C_
    s.aff = ptr(aff)
    s.mask = ptr(mask)
    if strides is not None:
        s.sN, s.sC, s.sH, s.sW = strides
    return s


DST_GRAD_BF16, DST_FWD_BF16 = 1, 2


INFO_BSUMS = 1 << 14     # include/posegan_hip.h PG_INFO_BSUMS
INFO_STEM_BIAS = 1 << 15  # PG_INFO_STEM_BIAS: pg_stem_wgrad_bf16_v2 also produced the bias gradient


def make_dst(grad, C_, fwd=None, aff=None, mask=None, act=ACT_NONE, accumulate=False, bsums=None):
    """grad / fwd may be fp32 or bf16 tensors (bf16 STORAGE on the bf16 data path): the dtype travels in Dst.flags.
    bsums: [N][STAT_SLOTS][2] double tensor (zeroed) that the epilogue may fill with the following norm backward's sums."""
    d = Dst()
    d.bsums = ptr(bsums)
    if grad is not None and torch.is_tensor(grad) and grad.dtype == torch.bfloat16:
        d.flags |= DST_GRAD_BF16
    if fwd is not None and torch.is_tensor(fwd) and fwd.dtype == torch.bfloat16:
        d.flags |= DST_FWD_BF16
    d.grad = ptr(grad)
    d.fwd = ptr(fwd)
    d.aff = ptr(aff)
    d.mask = ptr(mask)
    d.C = C_
    d.act = act
    d.accumulate = 1 if accumulate else 0
    return d
