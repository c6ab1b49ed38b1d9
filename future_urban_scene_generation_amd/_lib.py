"""ctypes binding of libfusg.so (include/fusg.h).  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C future_urban_scene_generation_amd/csrc``.

There is no fallback: if the shared object is missing or fails to load, every product entry point
raises ``FusgUnavailable``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FUSG_LIB", os.path.join(_HERE, "libfusg.so"))   # FUSG_LIB: kernel-development override


class FusgUnavailable(RuntimeError):
    pass


class FusgError(RuntimeError):
    pass


class Tensor(C.Structure):              # fusg_tensor
    _fields_ = [("data", C.c_void_p),
                ("n", C.c_int64), ("c", C.c_int64), ("h", C.c_int64), ("w", C.c_int64),
                ("sn", C.c_int64), ("sc", C.c_int64), ("sh", C.c_int64), ("sw", C.c_int64),
                ("dtype", C.c_int32), ("_pad", C.c_int32)]


class ConvDesc(C.Structure):            # fusg_conv_desc
    _fields_ = [("src0", Tensor), ("src1", Tensor), ("dst", Tensor), ("res0", Tensor), ("res1", Tensor),
                ("wpack", C.c_void_p), ("bias", C.c_void_p), ("ktab", C.c_void_p),
                ("pre_scale", C.c_void_p), ("pre_shift", C.c_void_p), ("pre_bstride", C.c_int64),
                ("workspace", C.c_void_p),
                ("k_pad", C.c_int32), ("c0k", C.c_int32), ("cout", C.c_int32), ("cout_pad", C.c_int32),
                ("stride", C.c_int32), ("upsample", C.c_int32), ("pad_mode", C.c_int32), ("pre_op", C.c_int32),
                ("act", C.c_int32), ("store_mode", C.c_int32), ("qh", C.c_int32), ("qw", C.c_int32),
                ("nphase", C.c_int32), ("out_sy", C.c_int32), ("out_sx", C.c_int32),
                ("out_oy", C.c_int32 * 4), ("out_ox", C.c_int32 * 4),
                ("dst_c_off", C.c_int32), ("tile", C.c_int32), ("ksplit", C.c_int32), ("precision", C.c_int32),
                ("wpack_h", C.c_void_p),
                ("kh", C.c_int32), ("kw", C.c_int32), ("dil", C.c_int32), ("pad_h", C.c_int32), ("pad_w", C.c_int32),
                ("wfrag_order", C.c_int32), ("wfrag", C.c_void_p), ("stats_out", C.c_void_p),
                ("tile_list", C.c_void_p), ("tile_count", C.c_int32), ("q_oy", C.c_int32), ("q_ox", C.c_int32),
                ("stats_slots", C.c_int32),
                ("wscale", C.c_void_p), ("status", C.c_void_p), ("wfrag_bf16", C.c_void_p), ("wfrag_f32", C.c_void_p),
                ("splitk_counters", C.c_void_p), ("splitk_counters_len", C.c_int32), ("_pad2", C.c_int32)]


class BneckDesc(C.Structure):           # fusg_bneck_desc
    _fields_ = [("x", Tensor), ("res", Tensor), ("dst", Tensor),
                ("pre_scale", C.c_void_p), ("pre_shift", C.c_void_p),
                ("w1frag", C.c_void_p), ("bias1", C.c_void_p), ("wscale1", C.c_void_p),
                ("w2frag", C.c_void_p), ("bias2", C.c_void_p), ("wscale2", C.c_void_p),
                ("w3frag", C.c_void_p), ("bias3", C.c_void_p), ("wscale3", C.c_void_p),
                ("status", C.c_void_p), ("planes", C.c_int32), ("exact_f32", C.c_int32)]


class PackSpec(C.Structure):            # fusg_pack_spec
    _fields_ = [(n, C.c_int32) for n in ("cout", "cin", "kh", "kw", "c0", "stride", "pad", "dil", "upsample", "cin_pad")]


class PackSizes(C.Structure):           # fusg_pack_sizes
    _fields_ = [("cout_pad", C.c_int32), ("k_pad", C.c_int32), ("c0k", C.c_int32), ("c1k", C.c_int32),
                ("wfrag_order", C.c_int32), ("_pad", C.c_int32), ("wpack_floats", C.c_int64), ("ktab_ints", C.c_int64),
                ("wpack_h_halves", C.c_int64), ("wfrag_halves", C.c_int64)]


# enums (include/fusg.h)
F32, U8, I32 = 0, 1, 2
PAD_ZERO, PAD_REFLECT, PAD_REPLICATE = 0, 1, 2
PRE_NONE, PRE_RELU, PRE_ELU, PRE_AFFINE_RELU, PRE_AFFINE = 0, 1, 2, 3, 4
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID, ACT_TANH01 = 0, 1, 2, 3, 4
STORE_NORMAL, STORE_D2S, STORE_S2D = 0, 1, 2
PREC_F32, PREC_F16X3, PREC_EMU_BF16, PREC_EMU_BF16X2, PREC_BF16 = 0, 1, 2, 3, 4
TILE_AUTO, TILE_128x128, TILE_128x64, TILE_128x32, TILE_64x64, TILE_64x128 = 0, 1, 2, 3, 4, 5

_TP = C.POINTER(Tensor)
_SIGS = {
    "fusg_conv2d": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "fusg_conv2d_plan": (C.c_int64, [C.POINTER(ConvDesc)]),
    "fusg_hg_bottleneck": (C.c_int, [C.POINTER(BneckDesc), C.c_void_p]),
    "fusg_chan_stats": (C.c_int, [_TP, C.c_void_p, C.c_int32, C.c_void_p]),
    "fusg_in_finalize": (C.c_int, [_TP, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fusg_ln_finalize": (C.c_int, [_TP, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "fusg_in_finalize_slots": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fusg_ln_finalize_slots": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "fusg_affine_act": (C.c_int, [_TP, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, _TP, _TP, C.c_void_p]),
    "fusg_maxpool2": (C.c_int, [_TP, _TP, C.c_void_p]),
    "fusg_upsample2_add": (C.c_int, [_TP, _TP, _TP, C.c_void_p]),
    "fusg_copy4d": (C.c_int, [_TP, _TP, C.c_int32, C.c_void_p]),
    "fusg_add4d": (C.c_int, [_TP, _TP, _TP, C.c_void_p]),
    "fusg_space_to_depth2": (C.c_int, [_TP, _TP, C.c_void_p]),
    "fusg_depth_to_space2": (C.c_int, [_TP, _TP, C.c_void_p]),
    "fusg_ec_inputs": (C.c_int, [_TP, _TP, _TP, _TP, C.c_int32, C.c_void_p]),
    "fusg_hshift_sum": (C.c_int, [_TP, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _TP, C.c_void_p]),
    "fusg_argmax_hw": (C.c_int, [_TP, C.c_void_p, C.c_void_p]),
    "fusg_to_image_u8": (C.c_int, [_TP, _TP, C.c_void_p]),
    "fusg_merge_u8": (C.c_int, [_TP, _TP, _TP, _TP, C.c_void_p]),
    "fusg_warp_perspective_u8": (C.c_int, [_TP, C.c_void_p, _TP, C.c_void_p]),
    "fusg_warp_perspective_indexed_u8": (C.c_int, [_TP, C.c_void_p, C.c_void_p, C.c_int32, _TP, C.c_void_p]),
    "fusg_fill_poly_planes_u8": (C.c_int, [_TP, C.c_void_p, C.c_void_p, C.c_int32, _TP, C.c_void_p]),
    "fusg_icn_inputs": (C.c_int, [_TP, _TP, _TP, C.c_void_p, _TP, C.c_void_p]),
    "fusg_lab2bgr_u8": (C.c_int, [_TP, _TP, C.c_void_p]),
    "fusg_paste_back_u8": (C.c_int, [_TP, _TP, C.c_void_p, _TP, C.c_void_p]),
    "fusg_paste_layers_u8": (C.c_int, [_TP, _TP, C.c_void_p, _TP, C.c_void_p, _TP, C.c_void_p]),
    "fusg_crop_resize_u8": (C.c_int, [_TP, C.c_void_p, _TP, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fusg_vunet_inputs": (C.c_int, [_TP, _TP, _TP, _TP, C.c_void_p, _TP, _TP, C.c_void_p]),
    "fusg_mask_bbox_geom": (C.c_int, [_TP, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fusg_keypoints_to_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "fusg_pnp_cpc": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 4 + [C.c_void_p] * 4),
    "fusg_plan_create": (C.c_void_p, []),
    "fusg_plan_destroy": (None, [C.c_void_p]),
    "fusg_plan_begin": (C.c_int, [C.c_void_p]),
    "fusg_plan_end": (C.c_int, [C.c_void_p]),
    "fusg_plan_add_dependency": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "fusg_plan_add_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p]),
    "fusg_plan_next_slot": (C.c_int, [C.c_void_p]),
    "fusg_plan_size": (C.c_int64, [C.c_void_p]),
    "fusg_plan_run": (C.c_int, [C.c_void_p]),
    "fusg_plan_streams": (C.c_int32, [C.c_void_p]),
    "fusg_plan_run_mt": (C.c_int, [C.c_void_p]),
    "fusg_plan_graph_capture": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "fusg_plan_graph_nodes": (C.c_int64, [C.c_void_p]),
    "fusg_plan_graph_slot": (C.c_int, [C.c_void_p]),
    "fusg_plan_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fusg_pack_conv_sizes": (C.c_int, [C.POINTER(PackSpec), C.POINTER(PackSizes)]),
    "fusg_pack_conv_weights": (C.c_int, [C.POINTER(PackSpec)] + [C.c_void_p] * 8),
    "fusg_version": (C.c_int, []),
    "fusg_last_error": (C.c_char_p, []),
    "fusg_last_conv_kernel": (C.c_int, []),
    "fusg_arch": (C.c_char_p, []),
    "fusg_sizeof_tensor": (C.c_int, []),
    "fusg_sizeof_conv_desc": (C.c_int, []),
    "fusg_sizeof_bneck_desc": (C.c_int, []),
    "fusg_prof_enable": (None, [C.c_int]),
    "fusg_prof_reset": (None, []),
    "fusg_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
}
EXPORTS = tuple(_SIGS.keys())

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raise FusgUnavailable if the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FusgUnavailable(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and libfusg.so is linked against the
        # system one (same SONAME).  Whichever is mapped first serves both, so torch goes first - loaded the other
        # way round, torch would run on a runtime it was not built for and the library's launches fail with
        # "no ROCm-capable device is detected".
        import torch  # noqa: F401
        try:
            h = C.CDLL(LIB_PATH)
        except OSError as e:                         # pragma: no cover
            raise FusgUnavailable(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if (h.fusg_sizeof_tensor() != C.sizeof(Tensor) or h.fusg_sizeof_conv_desc() != C.sizeof(ConvDesc)
                or h.fusg_sizeof_bneck_desc() != C.sizeof(BneckDesc)):
            raise FusgUnavailable("libfusg.so struct layout differs from the ctypes mirror (stale build?)")
        _lib = h
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().fusg_last_error()
        raise FusgError(f"{what}: rc={rc}: {msg.decode() if msg else ''}")
