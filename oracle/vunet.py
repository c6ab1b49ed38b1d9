"""Oracle: VUnet appearance-transfer network (reference vunet/models.py, vunet/layers.py).

Only the configuration the reference runs is restated: Namespace(up_mode='subpixel', w_norm=True,
drop_prob=0.2, vunet_256=True) (run_test.py:82), eval mode (Dropout2d is the identity).
"""
from __future__ import annotations

from typing import List, Mapping, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Mapping[str, torch.Tensor]


def _conv(sd: SD, p: str, x: torch.Tensor, stride: int = 1, padding: int = 0) -> torch.Tensor:
    """MyConv2d with weight_norm(dim=0), vunet/layers.py:26-36: w = g * v / ||v|| per out-channel."""
    w = torch._weight_norm(sd[p + ".conv.weight_v"], sd[p + ".conv.weight_g"], 0)
    return F.conv2d(x, w, sd[p + ".conv.bias"], stride=stride, padding=padding)


def _nin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """NiN = ELU -> 1x1 conv, vunet/layers.py:42-55 (conv lives at layers.1)."""
    return _conv(sd, p + ".layers.1", F.elu(x))


def _residual(sd: SD, p: str, x: torch.Tensor, skip: torch.Tensor = None) -> torch.Tensor:
    """Residual, vunet/layers.py:98-102: x + conv3x3(dropout(ELU(cat[x, skip]))) (conv at layers.2)."""
    residual = x
    if skip is not None:
        x = torch.cat([residual, skip], dim=1)
    return _conv(sd, p + ".layers.2", F.elu(x), padding=1) + residual


def depth_to_space(x: torch.Tensor, bs: int = 2) -> torch.Tensor:
    """DepthToSpace (DCR order), vunet/layers.py:182-193:
    out[b, c, bs*h + i, bs*w + j] = in[b, (i*bs + j) * C + c, h, w]   (NOT F.pixel_shuffle).
    Like the reference, the result is an NHWC-contiguous buffer viewed as NCHW (channels_last
    strides); the CPU conv kernels that consume it pick their blocking from the strides, so the
    memory format is kept to stay bit-identical with the reference on the same machine."""
    b, d, h, w = x.shape
    c = d // (bs * bs)
    y = x.permute(0, 2, 3, 1).reshape(b, h, w, bs, bs, c)            # (b, h, w, i, j, c)
    y = y.permute(0, 1, 3, 2, 4, 5).reshape(b, h * bs, w * bs, c)    # (b, h, i, w, j, c)
    return y.permute(0, 3, 1, 2)


def space_to_depth(x: torch.Tensor, bs: int = 2) -> torch.Tensor:
    """SpaceToDepth, vunet/layers.py:208-218 (inverse of depth_to_space).  Memory format note: the
    reference stacks per-column slabs (torch.stack(stack, 1)), so its result is physically
    [B, W', H', D] viewed as NCHW; reproduced here for bit-identity of the consuming CPU convs."""
    b, c, h, w = x.shape
    y = x.permute(0, 2, 3, 1).reshape(b, h // bs, bs, w // bs, bs, c)   # (b, h, i, w, j, c)
    y = y.permute(0, 3, 1, 2, 4, 5).reshape(b, w // bs, h // bs, bs * bs * c)   # (b, w, h, [i, j, c])
    return y.permute(0, 3, 2, 1)


def _sampler(sd: SD, p: str, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Sampler, vunet/layers.py:163-167: noise is drawn on the CPU *default* generator."""
    mu = _conv(sd, p + ".conv", x, padding=1)          # Sampler.conv is a MyConv2d -> key ...conv.conv.*
    sample = mu + torch.randn(*mu.size()).to(mu.device) * 1.0
    return mu, sample


def _upsample(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """UpSample(mode='subpixel'), vunet/layers.py:144-146."""
    return depth_to_space(_conv(sd, p + ".depth4x", x, padding=1), 2)


def _init_block(sd: SD, p: str, x: torch.Tensor):
    """InitBlock, vunet/models.py:155-162."""
    x = _nin(sd, p + ".nin", x)
    s0 = x = _residual(sd, p + ".residual_0", x)
    s1 = x = _residual(sd, p + ".residual_1", x)
    return x, [s0, s1]


def _down_block(sd: SD, p: str, x: torch.Tensor):
    """DownBlock, vunet/models.py:105-112."""
    x = _conv(sd, p + ".down.down", x, stride=2, padding=1)
    s0 = x = _residual(sd, p + ".residual_0", x)
    s1 = x = _residual(sd, p + ".residual_1", x)
    return x, [s0, s1]


def _up_block(sd: SD, p: str, x, skip_a, skip_b):
    """UpBlock, vunet/models.py:132-136."""
    x = _residual(sd, p + ".residual_0", x, skip_a)
    x = _residual(sd, p + ".residual_1", x, skip_b)
    return _upsample(sd, p + ".up", x)


def _ar_block(sd: SD, p: str, x, skip_a, enc_down_mu=None):
    """AutoRegressiveBlock.forward, vunet/models.py:56-86."""
    x = _residual(sd, p + ".residual_init", x, skip_a)
    x_ = space_to_depth(_residual(sd, p + ".residual_s2d", x), 2)
    if enc_down_mu is not None:
        g = torch.split(space_to_depth(enc_down_mu, 2), 128, 1)
        g = [_nin(sd, f"{p}.nin_{k}", g[k]) for k in range(3)]
    mus, zs = [], []
    for k in range(4):
        mu_k, z_k = _sampler(sd, f"{p}.sampler_{k}", x_)
        mus.append(mu_k)
        zs.append(z_k)
        if k < 3:
            cond = g[k] if enc_down_mu is not None else _nin(sd, f"{p}.nin_{k}", z_k)
            x_ = _residual(sd, f"{p}.residual_{k}", x_, cond)
    mu_0 = depth_to_space(torch.cat(mus, 1), 2).contiguous()
    z_0 = depth_to_space(torch.cat(zs, 1), 2)
    return x, mu_0, z_0


def vunet_enc_up(sd: SD, x: torch.Tensor):
    """forward_enc_up, vunet/models.py:333-353."""
    x, _ = _init_block(sd, "app_encoder_1", x)
    for name in ("app_encoder_1_a", "app_encoder_1_b", "app_encoder_1_c", "app_encoder_2", "app_encoder_3"):
        x, _ = _down_block(sd, name, x)
    skips = [_nin(sd, "app_skip_3_c", x)]
    x, sl = _down_block(sd, "app_encoder_4", x)
    outputs = [sl[-2], x]
    skips.append(_nin(sd, "app_skip_4_c", x))
    return outputs, skips


def vunet_enc_down(sd: SD, enc_up_outputs: Sequence[torch.Tensor], skips: Sequence[torch.Tensor]):
    """forward_enc_down, vunet/models.py:390-408."""
    x = _conv(sd, "app_bottleneck", enc_up_outputs[-1])
    x = _residual(sd, "app_decoder_1_a", x, skips[-1])
    mu_0, z_0 = _sampler(sd, "app_decoder_1_b", x)
    x_ = _conv(sd, "app_decoder_1_c", torch.cat([enc_up_outputs[-2], z_0], 1))
    x = _residual(sd, "app_decoder_1_d", x, x_)
    x = _upsample(sd, "app_decoder_1_e", x)
    x = _residual(sd, "app_decoder_2_a", x, None)
    mu_1, z_1 = _sampler(sd, "app_decoder_2_b", x)
    return [mu_0, mu_1], [z_0, z_1]


def vunet_dec_up(sd: SD, x: torch.Tensor):
    """forward_dec_up, vunet/models.py:355-388."""
    skips: List[torch.Tensor] = []
    x, sl = _init_block(sd, "shape_encoder_1", x)
    skips += [_nin(sd, "shape_skip_1_b", sl[-2]), _nin(sd, "shape_skip_1_c", sl[-1])]
    x, sl = _down_block(sd, "shape_encoder_1_a", x)
    skips += [_nin(sd, "shape_skip_1_a_b", sl[-2]), _nin(sd, "shape_skip_1_a_c", sl[-1])]
    for i in range(2, 7):
        x, sl = _down_block(sd, f"shape_encoder_{i}", x)
        skips += [_nin(sd, f"shape_skip_{i}_b", sl[-2]), _nin(sd, f"shape_skip_{i}_c", sl[-1])]
    return [x], skips


def vunet_dec_down(sd: SD, dec_up_outputs, skips: List[torch.Tensor], enc_down_mu=()):
    """forward_dec_down, vunet/models.py:410-459.  Pops ``skips`` empty, like the reference."""
    mu, z = [], []
    x = _conv(sd, "shape_bottleneck", dec_up_outputs[-1])
    for blk in (1, 2):
        skip_a = skips.pop()
        skip_b = skips.pop()
        m = None if len(enc_down_mu) == 0 else enc_down_mu[blk - 1]
        x, mu_k, z_k = _ar_block(sd, f"shape_decoder_{blk}", x, skip_a, m)
        mu.append(mu_k)
        z.append(z_k)
        x = _nin(sd, f"shape_decoder_{blk}_n", torch.cat([x, z_k], 1))
        x = _residual(sd, f"shape_decoder_{blk}_o", x, skip_b)
        x = _upsample(sd, f"shape_decoder_{blk}_p", x)
    for name in ("shape_decoder_3", "shape_decoder_4", "shape_decoder_5", "shape_decoder_5_a"):
        skip_a = skips.pop()
        skip_b = skips.pop()
        x = _up_block(sd, name, x, skip_a, skip_b)
    skip_a = skips.pop()
    skip_b = skips.pop()
    # EndBlock, vunet/models.py:181-185
    x = _residual(sd, "shape_decoder_6.residual_0", x, skip_a)
    x = _residual(sd, "shape_decoder_6.residual_1", x, skip_b)
    x = _conv(sd, "shape_decoder_6.conv", x, padding=1)   # EndBlock.conv is a MyConv2d -> keys shape_decoder_6.conv.conv.*
    assert not skips
    return x, mu, z


def vunet_forward(sd: SD, y_tilde: torch.Tensor, x: torch.Tensor = None, mean_mode: str = "mean_appearance",
                  first_frame_like_traj_test: bool = False):
    """Vunet_fix_res.forward, vunet/models.py:461-481 (passes z_app to forward_dec_down).

    ``first_frame_like_traj_test=True`` reproduces the call sequence of
    trajectory_inference.py:230-233 instead, which passes **mu_app**."""
    assert mean_mode in ("mean_appearance", "mean_shape")
    if mean_mode == "mean_appearance":
        outs, skips = vunet_enc_up(sd, x)
        mu_app, z_app = vunet_enc_down(sd, outs, skips)
        d_outs, d_skips = vunet_dec_up(sd, y_tilde)
        x_tilde, mu_shape, _ = vunet_dec_down(sd, d_outs, d_skips,
                                              mu_app if first_frame_like_traj_test else z_app)
        return x_tilde, mu_app, mu_shape
    d_outs, d_skips = vunet_dec_up(sd, y_tilde)
    x_tilde, _, _ = vunet_dec_down(sd, d_outs, d_skips)
    return x_tilde
