"""Deterministic synthetic weights and inputs (build-owned; SURVEY.md §8d "Synthetic inputs").

There are no pretrained checkpoints in this environment, so every parity fixture, test and
benchmark uses weights produced here.  A tensor is a pure function of
``(net, state_dict key, shape, seed)``; the same call in the golden-vector generator
(tools/gen_golden.py, which loads them into the *reference* modules), in the oracle and in the
HIP-backed modules therefore yields bit-identical parameters.

Everything is generated with torch CPU generators (host-side, load-time work; not hot path).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Mapping, Sequence, Tuple

import torch

# Per-net gains chosen so that activations stay O(1) through the whole depth with random weights
# (otherwise tanh saturates / images turn constant and SSIM against the reference is meaningless).
_CONV_GAIN = {"hg": 0.7, "icn": 1.4, "vunet": 0.7, "edge": 1.4, "inpaint": 1.4, "vgg": 1.4142135}


def _gen(net: str, key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(f"{net}/{key}".encode()) + 1000003 * int(seed)) & 0x7FFFFFFF)
    return g


def _normal(shape, std, g):
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32) * std


def _uniform(shape, lo, hi, g):
    return torch.rand(tuple(shape), generator=g, dtype=torch.float32) * (hi - lo) + lo


def synth_state_dict(net: str, schema: Mapping[str, Tuple[Sequence[int], str]], seed: int = 0
                     ) -> "OrderedDict[str, torch.Tensor]":
    """Build a full state_dict for ``net`` in {'hg','icn','vunet','edge','inpaint','vgg'}.

    ``schema`` maps state_dict key -> (shape, dtype-name), in state_dict order (the reference's
    own schema is shipped in the package, schemas/schema_*.json; the HIP-backed modules expose the
    identical schema, which tests assert).
    """
    gain = _CONV_GAIN[net]
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    keys = list(schema.keys())
    for key in keys:
        shape, dtype = schema[key]
        shape = tuple(int(s) for s in shape)
        g = _gen(net, key, seed)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t = torch.tensor(100, dtype=torch.int64)
        elif leaf == "running_mean":
            t = _normal(shape, 0.1, g)
        elif leaf == "running_var":
            t = _uniform(shape, 0.5, 1.5, g)
        elif leaf == "gamma":                                  # ICN LayerNorm
            t = _uniform(shape, 0.5, 1.5, g)
        elif leaf == "beta":
            t = _normal(shape, 0.1, g)
        elif leaf == "bias":
            t = _normal(shape, 0.05, g)
        elif leaf in ("weight", "weight_orig") and len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            t = _normal(shape, gain / fan_in ** 0.5, g)
        elif leaf == "weight" and len(shape) == 2:             # Linear (the CAD classifier's head)
            t = _normal(shape, gain / shape[1] ** 0.5, g)
        elif leaf == "weight" and len(shape) == 1:             # BatchNorm scale
            t = _uniform(shape, 0.5, 1.5, g)
        elif leaf == "weight_v" and len(shape) == 4:           # weight_norm direction
            t = _normal(shape, 0.05, g)
        elif leaf == "weight_g":                               # weight_norm magnitude, needs v
            t = None
        elif leaf in ("weight_u", "weight_v") and len(shape) == 1:   # spectral norm, needs W
            t = None
        else:
            raise KeyError(f"synth: no rule for {net}:{key} {shape}")
        out[key] = t
    # second pass: tensors that are functions of their siblings
    for key in keys:
        if out[key] is not None:
            continue
        shape, _ = schema[key]
        shape = tuple(int(s) for s in shape)
        prefix, leaf = key.rsplit(".", 1)
        g = _gen(net, key, seed)
        if leaf == "weight_g":
            v = out[prefix + ".weight_v"]
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            vnorm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(shape)
            out[key] = vnorm * (gain / (0.05 * fan_in ** 0.5)) * _uniform(shape, 0.8, 1.2, g)
        elif leaf == "weight_u":
            w = out[prefix + ".weight_orig"]
            n_u = shape[0]
            wm = w.reshape(w.shape[0], -1) if n_u == w.shape[0] else \
                w.permute(1, 0, 2, 3).reshape(w.shape[1], -1)
            u = torch.nn.functional.normalize(_normal((wm.shape[0],), 1.0, g), dim=0)
            v = None
            for _ in range(8):                                 # deterministic power iteration
                v = torch.nn.functional.normalize(wm.t().mv(u), dim=0)
                u = torch.nn.functional.normalize(wm.mv(v), dim=0)
            out[key] = u
            out[prefix + ".weight_v"] = v
    for key in keys:
        assert out[key] is not None and tuple(out[key].shape) == tuple(schema[key][0]), key
    return out


def schema_of(state_dict: Mapping[str, torch.Tensor]) -> Dict[str, Tuple[Tuple[int, ...], str]]:
    return OrderedDict((k, (tuple(v.shape), str(v.dtype).replace("torch.", "")))
                       for k, v in state_dict.items())


# ----------------------------------------------------------------------------------------------
# Synthetic inputs.  All draws use a private CPU generator so that they never
# disturb the global CPU RNG stream that VUnet's Sampler consumes (vunet/layers.py:163-167).
# ----------------------------------------------------------------------------------------------
_IMAGENET_MEAN = (0.485, 0.456, 0.406)
_IMAGENET_STD = (0.229, 0.224, 0.225)


def _smooth(x: torch.Tensor, k: int = 9) -> torch.Tensor:
    """Cheap low-pass so that synthetic 'images' have image-like spatial correlation."""
    c = x.shape[1]
    w = torch.ones(c, 1, k, k) / (k * k)
    return torch.nn.functional.conv2d(torch.nn.functional.pad(x, (k // 2,) * 4, mode="replicate"),
                                      w, groups=c)


def synth_inputs(net: str, batch: int, res: int = 256, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device="cpu")
    g.manual_seed(0x5EED0000 + zlib.crc32(net.encode()) % 65521 + 7919 * seed)
    B, R = batch, res
    if net == "hg":
        img = (0.5 * torch.rand(B, 3, R, R, generator=g) + 0.5 * _smooth(torch.rand(B, 3, R, R, generator=g)))
        mean = torch.tensor(_IMAGENET_MEAN).view(1, 3, 1, 1)
        std = torch.tensor(_IMAGENET_STD).view(1, 3, 1, 1)
        return {"x": ((img - mean) / std).contiguous()}
    if net == "icn":
        x = torch.rand(B, 21, R, R, generator=g) * 2 - 1
        # planes (channels 6..20) are exactly -1 where masked (LAB of black, normalised): ~60 %
        m = _smooth(torch.rand(B, 5, R, R, generator=g), 31) > 0.5 - 0.012
        m = m.repeat_interleave(3, dim=1)
        x[:, 6:] = torch.where(m, torch.full_like(x[:, 6:], -1.0), x[:, 6:])
        return {"x": x.contiguous()}
    if net == "vunet":
        x = torch.rand(B, 6, R, R, generator=g) * 2 - 1
        y = torch.rand(B, 3, R, R, generator=g) * 2 - 1
        return {"x": x.contiguous(), "y_tilde": y.contiguous()}
    if net in ("edge", "inpaint"):
        img = torch.rand(B, 3, R, R, generator=g)
        gray = (0.299 * img[:, 0:1] + 0.587 * img[:, 1:2] + 0.114 * img[:, 2:3]).contiguous()
        edge = (torch.rand(B, 1, R, R, generator=g) < 0.05).float()
        yy, xx = torch.meshgrid(torch.arange(R, dtype=torch.float32), torch.arange(R, dtype=torch.float32),
                                indexing="ij")
        cy = R * (0.4 + 0.2 * torch.rand(B, generator=g)).view(B, 1, 1)
        cx = R * (0.4 + 0.2 * torch.rand(B, generator=g)).view(B, 1, 1)
        ry = R * (0.15 + 0.15 * torch.rand(B, generator=g)).view(B, 1, 1)
        rx = R * (0.15 + 0.15 * torch.rand(B, generator=g)).view(B, 1, 1)
        mask = ((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1.0).float().unsqueeze(1)
        return {"img": img.contiguous(), "gray": gray, "edge": edge, "mask": mask.contiguous()}
    raise KeyError(net)
