"""Shared scaffolding of the drop-in network modules.

The drop-in classes keep the reference's ``nn.Module`` protocol (constructor signature,
``state_dict`` key schema, ``.to/.eval/.load_state_dict/__call__``) but hold *parameters only*:
their sub-modules are parameter holders with no ``forward``; all arithmetic is issued by the
owning network's ``forward*`` through libfusg (ops.py).  Derived quantities (folded / packed
filters, k-tables, BatchNorm scale/shift vectors) are cached per device and rebuilt whenever the
parameters may have changed (``load_state_dict``, ``.to()``, ``.float()`` ..., or ``refresh()``).
"""
from __future__ import annotations

import functools
import math
from typing import Optional

import torch
import torch.nn as nn


class ConvP(nn.Module):
    """Parameter holder with nn.Conv2d's (or nn.ConvTranspose2d's) state_dict schema: weight, bias."""

    def __init__(self, cin: int, cout: int, k: int, bias: bool = True, transposed: bool = False,
                 init: str = "default"):
        super().__init__()
        self.cin, self.cout, self.k, self.transposed = cin, cout, k, transposed
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        self.weight = nn.Parameter(torch.empty(shape))
        if bias:
            self.bias = nn.Parameter(torch.empty(cout))
        else:
            self.register_parameter("bias", None)
        fan_in = shape[1] * k * k
        with torch.no_grad():
            if init == "normal02":                      # edgeconnect BaseNetwork.init_weights('normal', 0.02)
                self.weight.normal_(0.0, 0.02)
                if bias:
                    self.bias.zero_()
            else:                                       # torch's Conv2d default
                nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
                if bias:
                    bound = 1 / math.sqrt(fan_in)
                    self.bias.uniform_(-bound, bound)


class BNP(nn.Module):
    """Parameter holder with nn.BatchNorm2d's schema (eval-mode use only)."""

    def __init__(self, c: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class WNConvP(nn.Module):
    """Holder with the schema of weight_norm(nn.Conv2d, dim=0): bias, weight_g, weight_v."""

    def __init__(self, cin: int, cout: int, k: int):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        self.bias = nn.Parameter(torch.empty(cout))
        v = torch.empty(cout, cin, k, k)
        nn.init.kaiming_uniform_(v, a=math.sqrt(5))
        self.weight_g = nn.Parameter(v.reshape(cout, -1).norm(dim=1).reshape(cout, 1, 1, 1).clone())
        self.weight_v = nn.Parameter(v)
        with torch.no_grad():
            bound = 1 / math.sqrt(cin * k * k)
            self.bias.uniform_(-bound, bound)


class SNConvP(nn.Module):
    """Holder with the schema of nn.utils.spectral_norm(conv): [bias], weight_orig, weight_u, weight_v."""

    def __init__(self, cin: int, cout: int, k: int, bias: bool = True, transposed: bool = False):
        super().__init__()
        self.cin, self.cout, self.k, self.transposed = cin, cout, k, transposed
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        if bias:
            self.bias = nn.Parameter(torch.zeros(cout))
        else:
            self.register_parameter("bias", None)
        w = torch.empty(shape).normal_(0.0, 0.02)
        self.weight_orig = nn.Parameter(w)
        wm = w.permute(1, 0, 2, 3).reshape(cout, -1) if transposed else w.reshape(cout, -1)
        u = nn.functional.normalize(torch.randn(wm.shape[0]), dim=0)
        v = nn.functional.normalize(wm.t().mv(u), dim=0)
        u = nn.functional.normalize(wm.mv(v), dim=0)
        self.register_buffer("weight_u", u)
        self.register_buffer("weight_v", v)


def entry_point(fn):
    """Decorator of the reference-facing entry points (forward, forward_enc_up, ...).

    Device guard: the reference's torch modules run on whatever device their tensors live on; libfusg launches on
    the CURRENT HIP device and ops.stream_ptr() returns the current device's stream, so the call is executed with
    the module's device made current (a model on 'cuda:1' works without torch.cuda.set_device(1)).

    Range guard: with the split-fp16 contraction (ops.PRECISION == "f16x3") a launch that meets an operand outside
    the split's range raises a device-side status word instead of saturating; this wrapper reads the word after the
    call (one 4-byte read = one stream synchronisation, which the reference's callers do anyway when they `.cpu()`
    the result) and, when it is set, repeats the call in exact fp32 with the CPU generators rewound and list
    arguments (the `skips` a VUnet decoder consumes) restored - the caller only ever sees a result every operand of
    which was represented.  Nested entry points (Vunet_fix_res.forward -> forward_*) are checked once, by the
    outermost one; `ops.defer_range_check()` (VehiclePipeline) postpones the check to the caller."""

    def guarded(self, dev, args, kwargs):
        from . import ops
        if not ops.range_guarded() or ops._GUARD["depth"] > 0 or ops._GUARD["deferred"] > 0:
            return fn(self, *args, **kwargs)
        snap = self._rng_snapshot()
        lists = [(a, list(a)) for a in args if isinstance(a, list)]
        ops._GUARD["depth"] += 1
        try:
            out = fn(self, *args, **kwargs)
        finally:
            ops._GUARD["depth"] -= 1
        if not ops.range_exceeded(dev):
            return out
        self._rng_restore(snap)
        for a, saved in lists:
            a[:] = saved
        ops._GUARD["depth"] += 1
        try:
            with ops.precision("f32"):
                return fn(self, *args, **kwargs)
        finally:
            ops._GUARD["depth"] -= 1

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        dev = self._device()
        if dev.type != "cuda":
            return fn(self, *args, **kwargs)                      # raises the "HIP device only" error inside
        if dev.index is not None and dev.index != torch.cuda.current_device():
            with torch.cuda.device(dev):
                return guarded(self, dev, args, kwargs)
        return guarded(self, dev, args, kwargs)

    return wrapper


class FusedNet(nn.Module):
    """Base of the four drop-in networks: plan cache + invalidation + device checks."""

    def __init__(self):
        super().__init__()
        self._plans = None
        self._plans_dev = None
        self._generation = 0

    # -- cache invalidation -----------------------------------------------------------------
    def refresh(self) -> None:
        """Drop the packed-weight cache (call after modifying parameters in place).  Bumps `generation`: a recorded
        pass (pipeline.CompiledPass) holds device pointers into the old cache and refuses to replay after this."""
        self._plans = None
        self._plans_dev = None
        self._generation = getattr(self, "_generation", 0) + 1

    @property
    def generation(self) -> int:
        return getattr(self, "_generation", 0)

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        self.refresh()
        return r

    def _apply(self, fn, *args, **kwargs):
        r = super()._apply(fn, *args, **kwargs)
        self.refresh()
        return r

    # -- host RNG state an entry point consumes (only the VUnet's samplers draw noise) --------
    def _rng_snapshot(self):
        return None

    def _rng_restore(self, snap) -> None:
        pass

    # -- helpers ----------------------------------------------------------------------------
    def _device(self) -> torch.device:
        return next(self.parameters()).device

    def _ensure(self, x: torch.Tensor) -> dict:
        from . import _lib
        from .ops import _require_gpu
        _require_gpu(x)
        _lib.lib()                                   # raises FusgUnavailable when the HIP library is missing
        if self.training:
            raise RuntimeError(f"{type(self).__name__}: inference only - call .eval() first (the reference "
                               "does, run_test.py:35-87)")
        dev = self._device()
        if dev != x.device:
            raise RuntimeError(f"{type(self).__name__} parameters are on {dev}, input on {x.device}")
        if self._plans is None or self._plans_dev != dev:
            with torch.no_grad():
                self._plans = self._build_plans(dev)
            self._plans_dev = dev
        return self._plans

    def _build_plans(self, device) -> dict:        # pragma: no cover - abstract
        raise NotImplementedError


def dev_vec(t: torch.Tensor, device) -> torch.Tensor:
    """Fresh, 16-byte aligned f32 device vector."""
    return t.detach().to(torch.float32).contiguous().to(device).clone()
