"""Thin Python wrappers over the libfusg C ABI (include/fusg.h).

PyTorch is used here only as plumbing: device allocations (caching allocator), the current HIP
stream, and host<->device copies.  All arithmetic on activations happens inside libfusg kernels.
Activations are torch tensors of *logical* NCHW shape whose memory is NHWC ("NHWC-physical":
channels contiguous, channel pitch a multiple of 4) so that callers can still `.cpu().numpy()`
them like any other tensor.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib as L
from .pack import ConvPlan

_DT = {torch.float32: L.F32, torch.uint8: L.U8, torch.int32: L.I32}

# Arithmetic of the conv contraction (include/fusg.h fusg_precision): "f32" = exact fp32 MFMA,
# "f16x3" = split-fp16 (fp32-class accuracy, 3 passes at the fp16 matrix rate).  Process-wide default,
# overridable per call; FUSG_PRECISION in the environment sets the initial value.
import os as _os
import numpy as _np


def _env_set(name: str) -> bool:
    """Development switch read from the environment (per call: tests and bench flip some of them at run time)."""
    return name in _os.environ


_PREC = {"f32": L.PREC_F32, "f16x3": L.PREC_F16X3,
         # BASELINE configs[4]: single-pass bf16 MFMA on the halo-kernel layers, f16x3 elsewhere (include/fusg.h)
         "bf16": L.PREC_BF16,
         # evidence paths (include/fusg.h): operands rounded to 8 / 16 significant bits on the exact-fp32 kernel
         "emu_bf16": L.PREC_EMU_BF16, "emu_bf16x2": L.PREC_EMU_BF16X2}
_EMU_BITS = {"emu_bf16": 8, "emu_bf16x2": 16}


def _round_sig_bits(w: torch.Tensor, bits: int) -> torch.Tensor:
    """fp32 -> nearest (ties to even) value with `bits` significant bits, as fp32."""
    drop = 24 - bits
    u = w.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    u = (u + ((1 << (drop - 1)) - 1) + ((u >> drop) & 1)) & ~((1 << drop) - 1) & 0xFFFFFFFF
    u = torch.where(u >= (1 << 31), u - (1 << 32), u).to(torch.int32)
    return u.view(torch.float32)
PRECISION = _os.environ.get("FUSG_PRECISION", "f16x3")


def set_precision(name: str) -> None:
    global PRECISION
    if name not in _PREC:
        raise ValueError(f"precision must be one of {list(_PREC)}")
    PRECISION = name


class precision:
    """`with ops.precision("f32"):` - temporary override of the process-wide contraction arithmetic."""

    def __init__(self, name: str):
        if name not in _PREC:
            raise ValueError(f"precision must be one of {list(_PREC)}")
        self.name = name

    def __enter__(self):
        global PRECISION
        self.prev, PRECISION = PRECISION, self.name
        return self

    def __exit__(self, *exc):
        global PRECISION
        PRECISION = self.prev
        return False


# ---- range guard of the split-fp16 contraction --------------------------------------------------------------
# An f16x3 launch that stages an operand with |x| >= 2^15 (or a non-finite one) cannot represent it and raises a
# device-side status word instead of clamping (include/fusg.h, fusg_precision).  The word is caller-owned: one int32
# per device, here.  The module entry points (nn_base.range_guarded) and VehiclePipeline.run read it after their
# launches and redo the call in exact fp32 when it is set, so a result computed from a saturated operand never
# reaches the caller.
# Single-threaded by contract, like the reference's callers (Threading): `_GUARD`, `_STATUS_SCOPE`,
# `PRECISION` and `RECORDER` are process-wide, not per-thread.
_STATUS = {}
_STATUS_SCOPE = []                       # innermost `status_scope` word; empty: the device's default word
_GUARD = {"depth": 0, "deferred": 0}


def range_guarded() -> bool:
    """Do launches of the current precision use the fp16 split (and hence its range status)?  f16x3 does everywhere;
    bf16 does on the layers that do not run on the halo kernel (and the hourglass keeps f16x3 under it)."""
    return PRECISION in ("f16x3", "bf16")


def halo_precision() -> bool:
    """Does the current precision route qualifying layers to the halo kernel?  (f32: its exact-fp32 mode, round 4.)"""
    return PRECISION in ("f16x3", "bf16") or (PRECISION == "f32" and not _env_set("FUSG_NO_F32_HALO"))


def new_status_word(device) -> torch.Tensor:
    return torch.zeros(4, dtype=torch.int32, device=torch.device(device))


def status_word(device) -> torch.Tensor:
    """The word the launches issued NOW report to: the innermost `status_scope`'s, else the device's default one
    (what the module entry points read)."""
    if _STATUS_SCOPE:
        return _STATUS_SCOPE[-1]
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    w = _STATUS.get(key)
    if w is None:
        w = _STATUS[key] = new_status_word(torch.device("cuda", key))
    return w


class status_scope:
    """Launches issued inside report their range status to `word` instead of the device's default word.  A
    VehiclePipeline owns one: passes issued with check="async" then cannot be cleared (or mistaken for its own) by a
    module entry point called between them and `finish()` - each reader only ever sees the launches it issued."""

    def __init__(self, word: torch.Tensor):
        self.word = word

    def __enter__(self):
        _STATUS_SCOPE.append(self.word)
        return self

    def __exit__(self, *exc):
        _STATUS_SCOPE.pop()
        return False


def range_exceeded(device, clear: bool = True, word: Optional[torch.Tensor] = None) -> bool:
    """Has any f16x3 launch reporting to `word` (default: the current scope's / the device's word) since the last
    clear seen an operand outside the split's range?  Synchronises the current stream (a 4-byte read)."""
    w = status_word(device) if word is None else word
    hit = bool(w[0].item())
    if hit and clear:
        w.zero_()
    return hit


class defer_range_check:
    """Inside this context the module entry points do not read the status word (no host synchronisation per
    network): the caller checks once, after all its launches (VehiclePipeline.run / finish)."""

    def __enter__(self):
        _GUARD["deferred"] += 1
        return self

    def __exit__(self, *exc):
        _GUARD["deferred"] -= 1
        return False


def _require_gpu(t: torch.Tensor, what: str = "input") -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what} is on {t.device}: the MI355X-native modules run on a HIP device only "
                           "(there is no CPU fallback; the CPU oracle lives in oracle/ for tests)")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr() -> int:
    """Raw hipStream_t of torch's current stream on the current device (the fast C accessor when this torch has
    it: `torch.cuda.current_stream()` costs ~4 us of Python per call, and there are ~370 launches per pass)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_NO_TENSOR = L.Tensor()

# ---- recorded passes (include/fusg.h, fusg_plan) -----------------------------------------------------------------
# While `RECORDER` is set, the calling thread is recording a pass into a fusg_plan (pipeline.CompiledPass): libfusg
# remembers every launch by itself; the Python side only has to route the two things that are not libfusg launches
# through the recorder - cross-stream dependencies and the host-to-device copies of per-pass host data.
RECORDER = None


class PlanRecorder:
    def __init__(self):
        self.handle = L.lib().fusg_plan_create()
        self.noise_slots = []                  # [(pinned host buffer, [shape, ...])] in draw order
        self.keep = []                         # tensors the recorded launches point into beyond the memory pool

    def dependency(self, waiter: int, signaller: int) -> None:
        L.check(L.lib().fusg_plan_add_dependency(self.handle, waiter, signaller), "plan_add_dependency")

    NSLOTS = 4                                 # ring depth of the pinned sources: the host may run this many passes ahead

    def h2d(self, dst: torch.Tensor, src_ring: torch.Tensor) -> None:
        """src_ring: pinned [NSLOTS, n] - slot 0 holds the data of the recording pass."""
        assert src_ring.is_pinned() and dst.is_cuda and src_ring.dim() == 2 and src_ring.shape[0] == self.NSLOTS
        nbytes = dst.numel() * dst.element_size()
        assert nbytes == src_ring.shape[1] * src_ring.element_size()
        L.check(L.lib().fusg_plan_add_h2d(self.handle, dst.data_ptr(), src_ring.data_ptr(), nbytes, self.NSLOTS,
                                          src_ring.stride(0) * src_ring.element_size(), stream_ptr()), "plan_add_h2d")


# ---- side streams ---------------------------------------------------------------------------------------------------
# Launches that depend on little else (the VUnet's shape encoder, its 1x1 skip projections, the conditioning of its
# autoregressive blocks) are issued on side streams so that they do not lengthen the chain of dependent launches on
# the main one - what bounds a pass at small batches.  Both directions of every hand-over are explicit
# (fork_to / join_from; through the plan recorder while a pass is being recorded), and a tensor that a side stream
# reads must be kept referenced by the caller until the join: the caching allocator only knows the stream a tensor was
# allocated on, and a recorded pass replays the addresses of the recording with no allocator in the loop.
# Fork each branch ONCE and join it once: every additional event hand-over between streams costs ~0.2 ms of latency
# on this stack (measured: the VUnet's 14 skip projections forked level by level onto a fifth stream made a B=1 replay
# 2.2 -> 4.1 ms and B=8 1223 -> 966 crops/s; GPU_MAX_HW_QUEUES made no difference) - DESIGN.md §9.
_SIDE = {}


def side_streams_enabled() -> bool:
    return _os.environ.get("FUSG_STREAMS", "1") != "0" and _os.environ.get("FUSG_VUNET_SPLIT", "1") != "0"


def side_stream(name: str, device) -> "torch.cuda.Stream":
    dev = torch.device(device)
    key = (name, dev.index if dev.index is not None else torch.cuda.current_device())
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=dev, priority=-1)
    return st


def fork_to(st: "torch.cuda.Stream") -> None:
    """`st` may not start anything issued from now on before what is queued on the current stream has finished."""
    cur = torch.cuda.current_stream(st.device)
    if RECORDER is None:
        st.wait_stream(cur)
    else:
        RECORDER.dependency(st.cuda_stream, cur.cuda_stream)


def join_from(st: "torch.cuda.Stream", tensors=()) -> None:
    """The current stream waits for `st`; `tensors` (allocated under `st`) will be used on the current stream."""
    cur = torch.cuda.current_stream(st.device)
    if RECORDER is None:
        cur.wait_stream(st)
    else:
        RECORDER.dependency(cur.cuda_stream, st.cuda_stream)
    for t in tensors:
        t.record_stream(cur)


def desc(t: Optional[torch.Tensor]) -> L.Tensor:
    """fusg_tensor for a 4-D torch tensor (or an absent tensor: a shared all-zero struct, never written to)."""
    if t is None:
        return _NO_TENSOR
    d = L.Tensor()
    assert t.dim() == 4, t.shape
    d.data = t.data_ptr()
    d.n, d.c, d.h, d.w = t.shape
    d.sn, d.sc, d.sh, d.sw = t.stride()
    d.dtype = _DT[t.dtype]
    return d


def h2d(a, device, dtype=None) -> torch.Tensor:
    """Small host data (a list, a numpy array, a CPU tensor) to the device WITHOUT synchronising the stream: staged in
    pinned memory (torch's caching host allocator holds the block until the copy has run) and copied asynchronously.
    `torch.tensor(..., device=...)` / `.to(device)` from pageable memory wait for everything queued on the stream - one
    such call per frame is enough to serialise `VehiclePipeline.run_frames`' host against its GPU."""
    if isinstance(a, _np.ndarray) and not a.flags.writeable:
        a = a.copy()                       # (a broadcast view: torch refuses to wrap read-only memory silently)
    t = a if isinstance(a, torch.Tensor) else torch.as_tensor(a)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous().pin_memory().to(device, non_blocking=True)


def nhwc_empty(b: int, c: int, h: int, w: int, device, dtype=torch.float32, zero: bool = False) -> torch.Tensor:
    """Logical [b, c, h, w] view over a fresh NHWC buffer whose channel pitch is c rounded up to 4."""
    cp = (c + 3) // 4 * 4
    buf = (torch.zeros if zero else torch.empty)((b, h, w, cp), device=device, dtype=dtype)
    t = buf.permute(0, 3, 1, 2)
    t = t if cp == c else t[:, :c]
    if zero:
        t._fusg_zero_pad = True         # this very object: its padding channels hold zeros (see as_nhwc)
    return t


def is_nhwc(t: torch.Tensor) -> bool:
    if t.dim() != 4 or t.dtype != torch.float32 or not t.is_cuda:
        return False
    b, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    if c > 1 and sc != 1:
        return False
    if sw < c or sw % 4:
        return False
    if h > 1 and sh != w * sw:
        return False
    if b > 1 and sn != h * w * sw:
        return False
    return t.data_ptr() % 16 == 0


def copy4d(src: torch.Tensor, dst: torch.Tensor, c_fill: int = 0) -> torch.Tensor:
    L.check(L.lib().fusg_copy4d(C.byref(desc(src)), C.byref(desc(dst)), int(c_fill), stream_ptr()), "copy4d")
    return dst


def as_nhwc(t: torch.Tensor, cpad: int = 4) -> torch.Tensor:
    """Return `t` itself if it already is NHWC-physical f32, else a converted copy (channel padding
    zero-filled).  Used at the module boundary for caller-provided NCHW tensors.  `cpad` > 4 forces a
    copy whose channel pitch is a multiple of cpad (zero-filled), for layers packed with cin_pad."""
    _require_gpu(t)
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    if cpad == 4 and is_nhwc(t):
        return t
    b, c, h, w = t.shape
    cp = (c + cpad - 1) // cpad * cpad
    # a buffer of ours whose pixel pitch already is the layer's and whose padding channels are known to be zero (the
    # glue kernels' outputs, a recorded pass's inputs): read in place, e.g. the ICN stem on fusg_icn_inputs' 24-pitch rows
    if getattr(t, "_fusg_zero_pad", False) and is_nhwc(t) and t.stride(3) == cp:
        return t
    buf = torch.empty((b, h, w, cp), device=t.device, dtype=torch.float32)
    full = buf.permute(0, 3, 1, 2)
    copy4d(t, full, cp)
    out = full if cp == c else full[:, :c]
    out._fusg_zero_pad = True              # (this very object: its padding channels were zero-filled just now)
    return out


def to_nchw(t: torch.Tensor) -> torch.Tensor:
    """Standard-contiguous NCHW copy (module outputs handed back to the caller)."""
    out = torch.empty(t.shape, device=t.device, dtype=torch.float32)
    return copy4d(t, out, 0)


# ---------------------------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------------------------
_WS = {}


_COUNTERS = {}
N_COUNTERS = 4096


def _splitk_counters(device) -> torch.Tensor:
    """Zeroed arrival counters of the in-launch split-K combine (fusg_conv_desc.splitk_counters), one array per device
    and stream: launches on one stream run in order and each leaves its counters zero."""
    key = (str(device), stream_ptr())
    c = _COUNTERS.get(key)
    if c is None:
        c = _COUNTERS[key] = torch.zeros(N_COUNTERS, device=device, dtype=torch.int32)
    if RECORDER is not None:
        RECORDER.keep.append(c)
    return c


def _workspace(device, nbytes: int) -> torch.Tensor:
    """Stream-ordered split-K scratch, grown on demand (one per device and stream: branches of the crop pass
    run concurrently on their own streams)."""
    key = (str(device), stream_ptr())
    ws = _WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty((max(nbytes, 1 << 22) + 3) // 4, device=device, dtype=torch.float32)
        _WS[key] = ws
    if RECORDER is not None:
        # a recorded pass replays this address: it must outlive a later, larger eager pass that replaces _WS[key]
        # (the old block would go back to the caching allocator and be handed to someone else while the plan still
        # writes split-K partial sums into it)
        RECORDER.keep.append(ws)
    return ws


def conv(plan: ConvPlan, x0: torch.Tensor, x1: Optional[torch.Tensor] = None, *, out: Optional[torch.Tensor] = None,
         out_c_off: int = 0, res0: Optional[torch.Tensor] = None, res1: Optional[torch.Tensor] = None,
         pre_op: int = L.PRE_NONE, pre: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, pre_bstride: int = 0,
         act: int = L.ACT_NONE, store: int = L.STORE_NORMAL, nchw_out: bool = False, tile: int = L.TILE_AUTO,
         ksplit: int = 0, precision: Optional[str] = None, want_stats: bool = False,
         out_stride: int = 1, out_off: Tuple[int, int] = (0, 0), tiles: Optional[torch.Tensor] = None,
         q_window: Optional[Tuple[int, int, int, int]] = None, q_size: Optional[Tuple[int, int]] = None,
         stats_into: Optional[Tuple[torch.Tensor, int]] = None):
    """One fused convolution launch (fusg_conv2d).  Returns the output tensor (allocated NHWC-physical
    unless `out` is given or `nchw_out` asks for a standard-contiguous NCHW result).
    out_stride / out_off: write output pixel (qy, qx) at (qy*out_stride + out_off[0], qx*out_stride + out_off[1])
    of `out` (phase launches).  tiles: int32 device tensor of 8x16-pixel patch indices - compute only those
    (halo-kernel launches only).  q_window = (oy, ox, h, w): compute only that window of the output grid (into the
    same positions of `out`, which must be given).  q_size = (qh, qw): output grid of a launch whose padding is not
    symmetric (transposed-conv phases; zero padding only).  stats_into = (buffer [B, slots, cout, 2], first_slot): write
    this launch's fused statistics slots there (the launch must qualify)."""
    plan.to(x0.device)
    if RECORDER is not None:
        RECORDER.keep.append(plan.dev)             # packed weights / tables the recorded launch points into
        if pre is not None:
            RECORDER.keep.append(pre)
    b, c0, h, w = x0.shape
    assert c0 == plan.c_split[0], (x0.shape, plan.c_split)
    if len(plan.c_split) > 1:
        assert x1 is not None and x1.shape[1] == plan.c_split[1] and x1.shape[0] == b and x1.shape[2:] == x0.shape[2:], \
            (x0.shape, None if x1 is None else x1.shape, plan.c_split)
    else:
        assert x1 is None
    qh, qw = plan.out_hw(h, w)
    if q_size is not None:
        assert out is not None and plan.nphase == 1 and plan.pad_mode == L.PAD_ZERO and q_window is None
        qh, qw = int(q_size[0]), int(q_size[1])
    if q_window is not None:
        assert out is not None and plan.nphase == 1 and store == L.STORE_NORMAL and not want_stats
        assert 0 <= q_window[0] and q_window[0] + q_window[2] <= qh and 0 <= q_window[1] and q_window[1] + q_window[3] <= qw
    if plan.nphase == 4:
        oc, oh, ow = plan.cout, 2 * qh, 2 * qw
    elif store == L.STORE_D2S:
        oc, oh, ow = plan.cout // 4, 2 * qh, 2 * qw
    elif store == L.STORE_S2D:
        oc, oh, ow = plan.cout * 4, qh // 2, qw // 2
    else:
        oc, oh, ow = plan.cout, qh, qw
    if out is None:
        out = torch.empty((b, oc, oh, ow), device=x0.device, dtype=torch.float32) if nchw_out \
            else nhwc_empty(b, oc, oh, ow, x0.device)
    d = L.ConvDesc()
    d.src0 = desc(x0)
    d.src1 = desc(x1)
    d.dst = desc(out)
    d.res0 = desc(res0)
    d.res1 = desc(res1)
    dev = plan.dev
    prec_name = precision or PRECISION
    if prec_name in _EMU_BITS:                # reduced-precision evidence path: weights rounded like the activations
        key = "wpack_r%d" % _EMU_BITS[prec_name]
        if key not in dev:
            dev[key] = _round_sig_bits(dev["wpack"], _EMU_BITS[prec_name])
        d.wpack = dev[key].data_ptr()
    else:
        d.wpack = dev["wpack"].data_ptr()
    d.bias = dev["bias"].data_ptr()
    d.ktab = dev["ktab"].data_ptr()
    if pre is not None:
        d.pre_scale = pre[0].data_ptr()
        d.pre_shift = pre[1].data_ptr()
        d.pre_bstride = int(pre_bstride)
    d.k_pad, d.c0k, d.cout, d.cout_pad = plan.k_pad, plan.c0k, plan.cout, plan.cout_pad
    d.stride, d.upsample, d.pad_mode = plan.stride, plan.upsample, plan.pad_mode
    d.pre_op, d.act, d.store_mode = int(pre_op), int(act), int(store)
    if q_window is not None:
        d.q_oy, d.q_ox, qh, qw = (int(v) for v in q_window)
    d.qh, d.qw, d.nphase = qh, qw, plan.nphase
    if plan.nphase == 4:
        d.out_sy = d.out_sx = 2
        for ph in range(4):
            d.out_oy[ph] = ph >> 1
            d.out_ox[ph] = ph & 1
    else:
        assert out_stride == 1 or (out is not None and store == L.STORE_NORMAL)
        d.out_sy = d.out_sx = int(out_stride)
        d.out_oy[0], d.out_ox[0] = int(out_off[0]), int(out_off[1])
    if tiles is not None:
        assert tiles.dtype == torch.int32 and tiles.is_contiguous() and tiles.device == x0.device
        d.tile_list, d.tile_count = tiles.data_ptr(), int(tiles.numel())
    d.dst_c_off = int(out_c_off)
    d.tile, d.ksplit = int(tile), int(ksplit)
    d.precision = _PREC[precision or PRECISION]
    d.wpack_h = dev["wpack_h"].data_ptr()
    d.wscale = dev["wscale"].data_ptr()
    d.status = status_word(x0.device).data_ptr()
    if plan.nphase == 1:                      # dense kh x kw tap grid: lets the library pick the halo-tiled kernel
        d.kh, d.kw, d.dil = plan.kh, plan.kw, plan.dil
        d.pad_h, d.pad_w = plan.pad, (plan.pad if plan.pad_w < 0 else plan.pad_w)
        if dev.get("wfrag") is not None:
            d.wfrag = dev["wfrag"].data_ptr()
            d.wfrag_order = dev["wfrag_order"]
            if prec_name == "bf16" and dev.get("wfrag_bf16") is not None:
                d.wfrag_bf16 = dev["wfrag_bf16"].data_ptr()
            if prec_name == "f32" and dev["wfrag_order"] in (0, 1, 2) and not _env_set("FUSG_NO_F32_HALO"):
                ff = plan.frag_f32_dev()                  # exact-fp32 halo kernel (built on first use)
                if ff is not None:
                    d.wfrag_f32 = ff.data_ptr()
                    if RECORDER is not None:
                        RECORDER.keep.append(ff)
    lib = L.lib()
    nbytes = lib.fusg_conv2d_plan(C.byref(d))
    ws = None
    if nbytes > 0:
        ws = _workspace(x0.device, nbytes)
        d.workspace = ws.data_ptr()
        # In-launch combine by the last-arriving workgroup (fusg_conv_desc.splitk_counters): correct and bit-identical,
        # but measured SLOWER than the separate reduce kernel on this chip (B=1 replay 2.6 -> 3.3 ms/pass, B=32 conv time
        # 26.4 -> 27.1 ms/step): every split workgroup pays an agent-scope release (an L2 write-back) that costs more
        # than the kernel boundary it saves - as cdna_hip_programming.md §5.6 predicts for seams of this size.  Opt-in.
        if _env_set("FUSG_SPLITK_IN_LAUNCH"):
            d.splitk_counters = _splitk_counters(x0.device).data_ptr()
            d.splitk_counters_len = N_COUNTERS
    stats = None
    if want_stats:
        # fused norm statistics when the launch qualifies (include/fusg.h, stats_out); else the caller falls
        # back to the separate streaming pass
        if (d.ksplit <= 1 and plan.nphase == 1 and store == L.STORE_NORMAL and act == L.ACT_NONE and res0 is None
                and res1 is None and (qh * qw) % 32 == 0 and plan.cout % 4 == 0 and out_c_off % 4 == 0 and is_nhwc(out)
                and out.stride(3) % 4 == 0 and not _env_set("FUSG_NO_VEC_EPI")):
            stats = torch.empty((b, qh * qw // 32, plan.cout, 2), device=x0.device, dtype=torch.float32)
            d.stats_out = stats.data_ptr()
    if stats_into is not None:
        buf, first = stats_into
        assert buf.dtype == torch.float32 and buf.is_contiguous() and buf.shape[0] == b and buf.shape[2] == plan.cout
        assert (qh * qw) % 32 == 0 and first + qh * qw // 32 <= buf.shape[1]
        d.stats_out = buf.data_ptr() + first * plan.cout * 2 * 4
        d.stats_slots = int(buf.shape[1])
    L.check(lib.fusg_conv2d(C.byref(d), stream_ptr()), "conv2d")
    return (out, stats) if want_stats else out


def bottleneck_ok(p: dict, x: torch.Tensor, precision: Optional[str] = None) -> bool:
    """Can this hourglass Bottleneck (packed plans c1, c2, c3 + bn1 affine) run as ONE launch (fusg_hg_bottleneck)?
    planes 64 or 128, split-fp16 or (round 4) exact-fp32 arithmetic, channels of x a multiple of 32.  FUSG_NO_BNECK=1 keeps the
    three launches (FUSG_NO_BNECK_F32=1: only for f32); FUSG_BNECK_MAXHW=n fuses only levels of at most n x n pixels."""
    prec = precision or PRECISION
    if prec not in ("f16x3", "f32") or _env_set("FUSG_NO_BNECK") or (prec == "f32" and (_env_set("FUSG_NO_BNECK_F32") or _env_set("FUSG_NO_F32_HALO"))):
        return False
    c1, c2, c3 = p["c1"], p["c2"], p["c3"]
    if not (c1.cout in (64, 128) and c2.cout == c1.cout and c3.cout == 2 * c1.cout and c1.kh == 1 and c2.kh == 3 and c3.kh == 1
            and c1.c0k == x.shape[1] and x.shape[1] % 32 == 0 and c1.c1k == 0 and c2.pad == 1 and c2.stride == 1):
        return False
    lim = _os.environ.get("FUSG_BNECK_MAXHW")
    if lim is not None and max(x.shape[2], x.shape[3]) > int(lim):
        return False
    # Levels below a minimum size run as three launches (small-image kernel at 4 x 4 / 8 x 8, halo kernel at 16 x 16) instead of the
    # fused block, whose grid there is one workgroup per image walking 48 K-steps in series.  Measured (profiles/
    # r04_ab_experiments.txt [r04j], recorded-plan replay): with at most 8 images the three launches win - B = 8 1266 -> 1305
    # crops/s, B = 1 2.10 -> 2.03 ms per pass with levels < 32 unfused - from B = 16 up the fused block wins (B = 32: conv 23.5 ->
    # 23.8 ms).  FUSG_BNECK_MINHW=n overrides (0 = always fused).
    minhw = BNECK_MINHW if BNECK_MINHW is not None else (32 if x.shape[0] <= 8 and prec == "f16x3" else 0)
    return max(x.shape[2], x.shape[3]) >= minhw or _env_set("FUSG_NO_SMALL")


BNECK_MINHW = int(_os.environ["FUSG_BNECK_MINHW"]) if "FUSG_BNECK_MINHW" in _os.environ else None


def bottleneck(p: dict, x: torch.Tensor, res: Optional[torch.Tensor] = None, precision: Optional[str] = None) -> torch.Tensor:
    """res + conv3(relu(conv2(relu(conv1(relu(bn1(x))))))) in one launch (fusg_hg_bottleneck; stacked_hourglass/
    models.py:22-42).  `p`: the packed Bottleneck (pre = bn1 scale / shift, c1 / c2 with bn2 / bn3 folded, c3);
    `res` defaults to x (no downsample conv).  precision "f32": the exact-fp32 form of the block (fp32 fragment copies)."""
    f32 = (precision or PRECISION) == "f32"
    res = x if res is None else res
    b, _, h, w = x.shape
    planes = p["c1"].cout
    out = nhwc_empty(b, 2 * planes, h, w, x.device)
    d = L.BneckDesc()
    d.x, d.res, d.dst = desc(x), desc(res), desc(out)
    d.pre_scale, d.pre_shift = p["pre"][0].data_ptr(), p["pre"][1].data_ptr()
    if RECORDER is not None:
        RECORDER.keep.append(p["pre"])
    for i, key in ((1, "c1"), (2, "c2"), (3, "c3")):
        plan = p[key].to(x.device)
        dev = plan.dev
        if RECORDER is not None:
            RECORDER.keep.append(dev)
        assert dev.get("wfrag") is not None and dev["wfrag_order"] == 0, key
        frag = plan.frag_f32_dev() if f32 else dev["wfrag"]
        assert frag is not None, key
        if RECORDER is not None:
            RECORDER.keep.append(frag)
        setattr(d, "w%dfrag" % i, frag.data_ptr())
        setattr(d, "bias%d" % i, dev["bias"].data_ptr())
        setattr(d, "wscale%d" % i, dev["wscale"].data_ptr())
    d.status = status_word(x.device).data_ptr()
    d.planes = planes
    d.exact_f32 = 1 if f32 else 0
    L.check(L.lib().fusg_hg_bottleneck(C.byref(d), stream_ptr()), "hg_bottleneck")
    return out


def last_conv_kernel() -> int:
    """Kernel family of the last conv launch issued by this thread (fusg_last_conv_kernel): 0 generic fp32,
    1 generic split-fp16, 2 halo, 3 halo in parity-quadrant (stride-2) form, 4 tap-unit kernel (few-channel stems),
    5 halo in bf16 mode, 6 fused hourglass Bottleneck, 7 pointwise-from-few-channels streaming kernel, 8 small-image kernel,
    9 halo in exact fp32, 10 tap-unit in exact fp32, 11 tap-unit in bf16 mode."""
    return int(L.lib().fusg_last_conv_kernel())


_BORDER_TILES = {}


def border_tiles(h: int, w: int, device) -> torch.Tensor:
    """Indices of the 8x16-pixel patches of an h x w image that touch its border (row-major patch grid)."""
    key = (h, w, str(device))
    t = _BORDER_TILES.get(key)
    if t is None:
        ny, nx = h // 8, w // 16
        idx = [y * nx + x for y in range(ny) for x in range(nx) if y in (0, ny - 1) or x in (0, nx - 1)]
        t = torch.tensor(idx, dtype=torch.int32, device=device)
        _BORDER_TILES[key] = t
    return t


def up2_phases_ok(x: torch.Tensor, precision: Optional[str] = None) -> bool:
    """Does `conv_up2` take the 4-phase route for this input?  (Any shape the reflect padding itself allows; the
    launches use the halo kernel where they qualify and the generic gather elsewhere.)"""
    b, c, h, w = x.shape
    return h >= 2 and w >= 2 and _os.environ.get("FUSG_NO_UP2_PHASES") is None


def conv_up2(exact: ConvPlan, phases, x: torch.Tensor, *, pre_op: int = L.PRE_NONE, pre=None, pre_bstride: int = 0,
             precision: Optional[str] = None, ring: Optional[dict] = None) -> torch.Tensor:
    """nn.Upsample(2) -> ReflectionPad2d(2) -> 5x5 conv.  `exact` = the 25-tap plan with the upsample fused into its
    gather (pack_conv(..., upsample=1)); `phases` = pack_conv_up2_d2s (one launch) or pack_conv_up2_phases (four) of the same filter.  When the shapes qualify,
    four 3x3 phase launches on the low-res input (9 MACs per output instead of 25) write the interleaved output; the
    outermost ring of pixels, the only place where the phase form and the reflect-padded 25-tap form differ
    (pack.up2_phase_weights), is then rewritten: by four one-pixel-wide windows of the 25-tap form, or - with `ring`
    (pack.pack_conv_up2_ring) - by twelve small 3x3 launches whose weights are regrouped for the border (9 MACs there
    too).  Measured (round 3, same card): the twelve launches, eight of them 32 - 4000 output pixels small, cost as much as the
    four efficient 25-tap windows (conv 23.2 vs 23.3 - 23.7 ms per step, 1437 vs 1426 crops/s), so the networks do not
    pass `ring` unless FUSG_UP2_RING9 is set."""
    if phases is None or not up2_phases_ok(x, precision):
        return conv(exact, x, pre_op=pre_op, pre=pre, pre_bstride=pre_bstride, precision=precision)
    b, c, h, w = x.shape
    out = nhwc_empty(b, exact.cout, 2 * h, 2 * w, x.device)
    # split-K launches do not use the halo kernel: keep K whole where the phase launches qualify for it
    halo = (((precision or PRECISION) in ("f16x3", "bf16") or ((precision or PRECISION) == "f32" and not _env_set("FUSG_NO_F32_HALO")))
            and c % 32 == 0 and h % 8 == 0 and w % 16 == 0 and _os.environ.get("FUSG_NO_HALO") is None)
    if isinstance(phases, ConvPlan):                               # all four phases in one launch, DepthToSpace store
        conv(phases, x, out=out, store=L.STORE_D2S, pre_op=pre_op, pre=pre, pre_bstride=pre_bstride,
             precision=precision, ksplit=1 if halo else 0)
    else:
        for ph, plan in enumerate(phases):
            conv(plan, x, out=out, out_stride=2, out_off=(ph >> 1, ph & 1), pre_op=pre_op, pre=pre,
                 pre_bstride=pre_bstride, precision=precision, ksplit=1 if halo else 0)
    if ring is not None:
        from .pack import up2_ring_launches
        for ry, rx, win, off in up2_ring_launches(h, w):
            if win[2] > 0 and win[3] > 0:
                conv(ring[(ry, rx)], x, out=out, q_window=win, out_stride=2, out_off=off, pre_op=pre_op, pre=pre,
                     pre_bstride=pre_bstride, precision=precision)
        return out
    # the outermost ring of output pixels, with the 25-tap form: four one-pixel-wide windows of the full convolution
    for win in ((0, 0, 1, 2 * w), (2 * h - 1, 0, 1, 2 * w), (0, 0, 2 * h, 1), (0, 2 * w - 1, 2 * h, 1)):
        conv(exact, x, out=out, q_window=win, pre_op=pre_op, pre=pre, pre_bstride=pre_bstride, precision=precision)
    return out


def conv_transpose_phases(phases, x: torch.Tensor, *, pre_op: int = L.PRE_NONE, pre=None, pre_bstride: int = 0,
                          act: int = L.ACT_NONE, want_stats: bool = False, precision: Optional[str] = None):
    """nn.ConvTranspose2d(k4, s2, p1) as four dense 2x2 stride-1 launches with strided stores
    (pack.pack_conv_transpose_k4s2p1_phases): each one qualifies for the halo kernel when the input does
    (split-fp16, channels % 32 == 0, H % 8 == 0, W % 16 == 0).  With want_stats the four launches write disjoint
    slot ranges of one statistics buffer: returns (out, stats [B, 4*h*w/32, cout, 2] or None)."""
    b, c, h, w = x.shape
    cout = phases[0].cout
    out = nhwc_empty(b, cout, 2 * h, 2 * w, x.device)
    n = h * w // 32
    fuse = (want_stats and act == L.ACT_NONE and (h * w) % 32 == 0 and cout % 4 == 0
            and _os.environ.get("FUSG_NO_VEC_EPI") is None)
    stats = torch.empty((b, 4 * n, cout, 2), device=x.device, dtype=torch.float32) if fuse else None
    for ph, plan in enumerate(phases):
        conv(plan, x, out=out, out_stride=2, out_off=(ph >> 1, ph & 1), q_size=(h, w), pre_op=pre_op, pre=pre,
             pre_bstride=pre_bstride, act=act, precision=precision, ksplit=1,
             stats_into=(stats, ph * n) if fuse else None)
    return (out, stats) if want_stats else out


def conv_rowsplit(plan: ConvPlan, x0: torch.Tensor, *, pre_op: int = L.PRE_NONE, pre=None, pre_bstride: int = 0,
                  act: int = L.ACT_NONE, nchw_out: bool = True) -> torch.Tensor:
    """Small-cout head (pack.pack_conv_rowsplit): kh x 1 implicit GEMM + horizontal gather-sum."""
    rs = plan.rowsplit
    t = conv(plan, x0, pre_op=pre_op, pre=pre, pre_bstride=pre_bstride)
    b, _, h, w = t.shape
    out = torch.empty((b, rs["cout"], h, w), device=x0.device, dtype=torch.float32) if nchw_out \
        else nhwc_empty(b, rs["cout"], h, w, x0.device)
    L.check(L.lib().fusg_hshift_sum(C.byref(desc(t)), plan.dev["rs_bias"].data_ptr(), rs["kw"], rs["pad"], plan.pad_mode,
                                    int(act), C.byref(desc(out)), stream_ptr()), "hshift_sum")
    return out


# ---------------------------------------------------------------------------------------------
# normalisation
# ---------------------------------------------------------------------------------------------
def _nchunk(b: int, c: int, hw: int) -> int:
    zg = (c // 4 + 63) // 64
    want = max(1, 1024 // max(1, b * zg))
    return int(max(1, min(want, max(1, hw // 64), 4096)))


def _chan_stats(x: torch.Tensor):
    b, c, h, w = x.shape
    n = _nchunk(b, c, h * w)
    partial = torch.empty((b, n, c, 2), device=x.device, dtype=torch.float32)
    L.check(L.lib().fusg_chan_stats(C.byref(desc(x)), partial.data_ptr(), n, stream_ptr()), "chan_stats")
    return partial, n


def instnorm_stats(x: torch.Tensor, eps: float = 1e-5) -> Tuple[torch.Tensor, torch.Tensor]:
    """(scale, shift) [B, C] such that InstanceNorm2d(x) = x*scale + shift."""
    b, c = x.shape[:2]
    partial, n = _chan_stats(x)
    ss = torch.empty((2, b, c), device=x.device, dtype=torch.float32)
    L.check(L.lib().fusg_in_finalize(C.byref(desc(x)), partial.data_ptr(), n, float(eps), ss[0].data_ptr(),
                                     ss[1].data_ptr(), stream_ptr()), "in_finalize")
    return ss[0], ss[1]


def layernorm_stats(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5):
    """(scale, shift) [B, C] of the ICN's custom LayerNorm (warp_learn/models.py:26-35)."""
    b, c = x.shape[:2]
    partial, n = _chan_stats(x)
    ss = torch.empty((2, b, c), device=x.device, dtype=torch.float32)
    L.check(L.lib().fusg_ln_finalize(C.byref(desc(x)), partial.data_ptr(), n, float(eps), gamma.data_ptr(),
                                     beta.data_ptr(), ss[0].data_ptr(), ss[1].data_ptr(), stream_ptr()), "ln_finalize")
    return ss[0], ss[1]


def conv_in(plan: ConvPlan, x0: torch.Tensor, eps: float = 1e-5, **kw):
    """conv followed by InstanceNorm2d statistics: returns (raw conv output, (scale, shift)).  The statistics
    come out of the conv epilogue when the launch qualifies, else from the streaming pass."""
    if isinstance(plan, (list, tuple)):                               # transposed conv as four phase plans
        out, stats = conv_transpose_phases(plan, x0, want_stats=True, **kw)
    else:
        out, stats = conv(plan, x0, want_stats=True, **kw)
    if stats is None:
        return out, instnorm_stats(out, eps)
    b, nslots, c, _ = stats.shape
    ss = torch.empty((2, b, c), device=out.device, dtype=torch.float32)
    L.check(L.lib().fusg_in_finalize_slots(stats.data_ptr(), b, nslots, c, float(eps), ss[0].data_ptr(), ss[1].data_ptr(),
                                           stream_ptr()), "in_finalize_slots")
    return out, (ss[0], ss[1])


def conv_ln(plan: ConvPlan, x0: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, **kw):
    """conv followed by the ICN's custom LayerNorm statistics (see layernorm_stats)."""
    out, stats = conv(plan, x0, want_stats=True, **kw)
    if stats is None:
        return out, layernorm_stats(out, gamma, beta, eps)
    b, nslots, c, _ = stats.shape
    ss = torch.empty((2, b, c), device=out.device, dtype=torch.float32)
    L.check(L.lib().fusg_ln_finalize_slots(stats.data_ptr(), b, nslots, c, float(eps), gamma.data_ptr(), beta.data_ptr(),
                                           ss[0].data_ptr(), ss[1].data_ptr(), stream_ptr()), "ln_finalize_slots")
    return out, (ss[0], ss[1])


def affine_act(x: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor], act: int = L.ACT_NONE,
               res: Optional[torch.Tensor] = None, bstride: Optional[int] = None) -> torch.Tensor:
    b, c, h, w = x.shape
    out = nhwc_empty(b, c, h, w, x.device)
    if bstride is None:
        bstride = c
    L.check(L.lib().fusg_affine_act(C.byref(desc(x)), scale.data_ptr() if scale is not None else None,
                                    shift.data_ptr() if shift is not None else None, int(bstride), int(act),
                                    C.byref(desc(res)) if res is not None else None, C.byref(desc(out)), stream_ptr()),
            "affine_act")
    return out


# ---------------------------------------------------------------------------------------------
# data movement / small ops
# ---------------------------------------------------------------------------------------------
def maxpool2(x: torch.Tensor) -> torch.Tensor:
    b, c, h, w = x.shape
    out = nhwc_empty(b, c, h // 2, w // 2, x.device)
    L.check(L.lib().fusg_maxpool2(C.byref(desc(x)), C.byref(desc(out)), stream_ptr()), "maxpool2")
    return out


def upsample2_add(low: torch.Tensor, up1: torch.Tensor) -> torch.Tensor:
    out = nhwc_empty(*up1.shape, up1.device)
    L.check(L.lib().fusg_upsample2_add(C.byref(desc(low)), C.byref(desc(up1)), C.byref(desc(out)), stream_ptr()),
            "upsample2_add")
    return out


def add4d(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if out is None:
        out = nhwc_empty(*a.shape, a.device)
    L.check(L.lib().fusg_add4d(C.byref(desc(a)), C.byref(desc(b)), C.byref(desc(out)), stream_ptr()), "add4d")
    return out


def space_to_depth2(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    b, c, h, w = x.shape
    if out is None:
        out = nhwc_empty(b, 4 * c, h // 2, w // 2, x.device)
    L.check(L.lib().fusg_space_to_depth2(C.byref(desc(x)), C.byref(desc(out)), stream_ptr()), "space_to_depth2")
    return out


def depth_to_space2(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    b, c, h, w = x.shape
    if out is None:
        out = nhwc_empty(b, c // 4, 2 * h, 2 * w, x.device)
    L.check(L.lib().fusg_depth_to_space2(C.byref(desc(x)), C.byref(desc(out)), stream_ptr()), "depth_to_space2")
    return out


def ec_inputs(images: torch.Tensor, edges: torch.Tensor, masks: torch.Tensor, mode: int) -> torch.Tensor:
    b, _, h, w = images.shape
    out = nhwc_empty(b, 4, h, w, images.device)
    L.check(L.lib().fusg_ec_inputs(C.byref(desc(images)), C.byref(desc(edges)), C.byref(desc(masks)), C.byref(desc(out)),
                                   int(mode), stream_ptr()), "ec_inputs")
    return out if mode == 1 else out[:, :3]


def argmax_hw(x: torch.Tensor) -> torch.Tensor:
    """int32 [B, C] row-major first-occurrence argmax over H*W."""
    _require_gpu(x)
    b, c = x.shape[:2]
    idx = torch.empty((b, c), device=x.device, dtype=torch.int32)
    L.check(L.lib().fusg_argmax_hw(C.byref(desc(x)), idx.data_ptr(), stream_ptr()), "argmax_hw")
    return idx


def to_image_u8(x: torch.Tensor) -> torch.Tensor:
    """uint8 [B, H, W, C] = trunc(clip((x+1)/2*255)) (to_image without the LAB branch)."""
    _require_gpu(x)
    b, c, h, w = x.shape
    out = torch.empty((b, h, w, c), device=x.device, dtype=torch.uint8)
    L.check(L.lib().fusg_to_image_u8(C.byref(desc(x)), C.byref(desc(out.permute(0, 3, 1, 2))), stream_ptr()), "to_image_u8")
    return out


def merge_u8(out_: torch.Tensor, img: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """uint8 [B, H, W, C] = trunc((out*m + img*(1-m))*255) (trajectory_inference.py:126-129)."""
    b, c, h, w = out_.shape
    dst = torch.empty((b, h, w, c), device=out_.device, dtype=torch.uint8)
    L.check(L.lib().fusg_merge_u8(C.byref(desc(out_)), C.byref(desc(img)), C.byref(desc(mask)),
                                  C.byref(desc(dst.permute(0, 3, 1, 2))), stream_ptr()), "merge_u8")
    return dst


# ---------------------------------------------------------------------------------------------
# profiler hooks (bench.py roofline leg)
# ---------------------------------------------------------------------------------------------
def prof_enable(on: bool) -> None:
    L.lib().fusg_prof_enable(1 if on else 0)


def prof_reset() -> None:
    L.lib().fusg_prof_reset()


def prof_read(kind: int = 0):
    ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
    L.check(L.lib().fusg_prof_read(kind, C.byref(ms), C.byref(n), C.byref(fl)), "prof_read")
    return ms.value, n.value, fl.value
