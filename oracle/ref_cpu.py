"""CPU oracle for the EEG-GAN progressive 1-D WGAN-GP training step.

TEST INFRASTRUCTURE ONLY.  This module is a CPU restatement, in stock PyTorch
ops, of the arithmetic the reference performs on its hot path.  It exists to
check the HIP path and to be timed as the CPU baseline.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; nothing in the shipped package may.

Parity status: PINNED.  ``tools/make_golden.py`` imports the reference's own
modules in the build container, runs them on seeded inputs and checks this
restatement against them (fp64: to round-off) before writing the fixtures
under ``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks the
restatement against those fixtures wherever the tests run.

Everything is functional: a network is a flat ``dict`` of tensors keyed by the
reference's ``state_dict`` names plus a ``dict`` of equalized-lr scales keyed
by the conv's module path (the reference keeps that scale as a Python float
outside the state_dict, layers.py:165-169).

Reference lines each function follows are cited in its docstring
(paths relative to the reference repository root).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from fractions import Fraction
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-8  # utils.py:14


# --------------------------------------------------------------------------- config
@dataclass
class NetConfig:
    """Default-config constants of the path (defaults.yml:16-58, dataset.py:28-51)."""
    initial_kernel_size: int = 32
    num_rgb_channels: int = 5
    fmap_base: int = 1024
    fmap_max: int = 256
    fmap_min: int = 64
    kernel_size: int = 3
    scale_up: Sequence[int] = (4, 2, 2, 3, 4, 5, 3)
    scale_down: Sequence[int] = (1, 1, 1, 2, 3, 4, 2)
    act_alpha: float = 0.2
    latent_size: int = 256
    g_dropout: float = 0.2
    d_dropout: float = 0.2
    group_size: int = 4
    temporal_groups_per_window: int = 1    # MinibatchStddev windows of initial_kernel_size / this many steps (layers.py:72)
    tanh_scale: float = 0.5
    # options beside the defaults (network.py:152-153, 225-230, 400-401, 453-457)
    residual: bool = False                 # identity / 1x1-conv skips around every block but the 32-sample ones
    rgb_generation_mode: str = "pggan"     # 'residual' (sum of all stages' signals) | 'mean'
    no_tanh: bool = False                  # to_rgb without the ScaledTanh
    normalize_latents: bool = True
    input_to_all_layers: bool = False      # the critic sees the down-sampled input at every stage
    deep: bool = False                     # blocks of four layers through a bottleneck, with slice / concatenation skips
    self_attention_layers: Sequence[int] = ()  # stages with a SelfAttention layer (network.py:304-310, 503-509)
    g_act_norm: Optional[str] = "pixel"    # the generator layers' norm: 'pixel' | 'batch' | None (network.py:229, 257)
    z_to_bn: bool = False                  # batch norms modulated by the latent (network.py:253, 299, 317-322)
    split_z: bool = False                  # one share of the latent per block (network.py:294-295, 317-322, 368-371)
    to_rgb_mode: str = "pggan"             # 'sngan' | 'sagan' | 'biggan' (network.py:173-187)
    d_act_norm: Optional[str] = None       # the critic layers' norm: None | 'batch' (network.py:455, 496-497)
    g_conv_only: bool = False              # latents [B, latent, L], k-tap first layer (network.py:33-36, 303, 364-365)
    d_conv_only: bool = False              # k-tap last layer, score = mean over time (network.py:416-418, 554)

    def nf(self, stage: int) -> int:
        """network.py:279-280."""
        return min(max(int(self.fmap_base / (2.0 ** stage)), self.fmap_min), self.fmap_max)

    @property
    def num_stages(self) -> int:
        return len(self.scale_up)

    def seq_len(self, depth: int) -> int:
        t = self.initial_kernel_size
        for i in range(depth):
            t = t * self.scale_up[i] // self.scale_down[i]
        return t


# --------------------------------------------------------------------------- parameter layout
def conv_shapes_generator(cfg: NetConfig) -> List[Tuple[str, int, int, int]]:
    """(module path, Cin, Cout, k) of every conv in G, in construction order
    (network.py:302-312, 158-161, 175-176)."""
    out = []

    def block(prefix, ch_in, ch_out, first):
        k0 = cfg.initial_kernel_size if first else cfg.kernel_size
        out.append((f"{prefix}.convs.0", ch_in, ch_out, k0))
        out.append((f"{prefix}.convs.1", ch_out, ch_out, cfg.kernel_size))
        out.append((f"{prefix}.to_rgb.0", ch_out, cfg.num_rgb_channels, 1))

    block("block0", cfg.latent_size, cfg.nf(1), True)
    for i in range(cfg.num_stages):
        block(f"blocks.{i}", cfg.nf(i + 1), cfg.nf(i + 2), False)
    return out


def conv_shapes_discriminator(cfg: NetConfig) -> List[Tuple[str, int, int, int]]:
    """(module path, Cin, Cout, k) of every conv in D, in construction order
    (network.py:501-525, 405-429): the last block is built first, then blocks
    R-1..0 which land at list positions 0..R-1, then ``linear``."""
    out = []
    R = cfg.num_stages
    # last block: stored at index R; net.0 is the parameter-less MinibatchStddev
    out.append((f"blocks.{R}.net.1", cfg.nf(1) + 1, cfg.nf(1), cfg.kernel_size))
    out.append((f"blocks.{R}.net.2", cfg.nf(1), cfg.nf(0), cfg.initial_kernel_size))
    out.append((f"blocks.{R}.from_rgb", cfg.num_rgb_channels, cfg.nf(1), 1))
    for pos, i in enumerate(range(R - 1, -1, -1)):
        ch_in, ch_out = cfg.nf(i + 2), cfg.nf(i + 1)
        out.append((f"blocks.{pos}.net.0", ch_in, ch_in, cfg.kernel_size))
        out.append((f"blocks.{pos}.net.1", ch_in, ch_out, cfg.kernel_size))
        out.append((f"blocks.{pos}.from_rgb", cfg.num_rgb_channels, ch_in, 1))
    out.append(("linear", cfg.nf(0), 1, 1))
    return out


def init_params(shapes, act_alpha_of, generator: torch.Generator, dtype=torch.float32):
    """Unit-RMS equalized init (layers.py:156-169): kaiming-normal with the
    layer's leaky slope, bias zero, then weight /= RMS(weight) and the RMS is
    kept as the runtime input scale.  This draws from ``generator`` directly
    and is NOT stream-compatible with the reference's nn.Conv1d construction
    (which also consumes draws for the default reset); fixtures therefore carry
    weights produced by the shipped package's modules, not by this helper.
    """
    params: Dict[str, torch.Tensor] = {}
    scales: Dict[str, float] = {}
    for path, cin, cout, k in shapes:
        a = act_alpha_of(path)
        a = a if a > 0 else 1.0
        gain = math.sqrt(2.0 / (1.0 + a * a))
        std = gain / math.sqrt(cin * k)
        w = torch.randn(cout, cin, k, generator=generator, dtype=torch.float64) * std
        s = float(torch.sqrt(torch.mean(w * w)))
        params[f"{path}.conv.conv.weight"] = (w / s).to(dtype)
        params[f"{path}.conv.conv.bias"] = torch.zeros(cout, dtype=dtype)
        scales[path] = s
    return params, scales


# --------------------------------------------------------------------------- primitive ops
def equalized_conv1d(x, w, b, scale: float, pad: int):
    """layers.py:173-174: the INPUT is multiplied by the scale, then a plain
    cross-correlation with zero padding and an unscaled bias."""
    return F.conv1d(x * scale, w, b, padding=pad)


def pixel_norm(h):
    """utils.py:133-136: h * rsqrt(mean over channels of h^2 + 1e-8)."""
    return h * torch.rsqrt(torch.mean(h * h, dim=1, keepdim=True) + EPS)


def leaky_relu(x, slope: float):
    """layers.py:196-200 (nn.LeakyReLU, never in place)."""
    return F.leaky_relu(x, slope)


class DropSource:
    """Source of the GDrop multipliers (layers.py:32-38).

    The reference draws ``(1+strength) ** np.random.normal(size=[B, C, 1])`` on
    the host at every call, in execution order, in fp64, then casts to the
    activation dtype.  ``mode='numpy'`` reproduces that stream from the global
    numpy RNG; ``mode='list'`` replays recorded multipliers; ``mode='off'``
    returns None (deterministic).
    """

    def __init__(self, mode: str = "numpy", recorded: Optional[List[np.ndarray]] = None):
        self.mode = mode
        self.recorded = list(recorded) if recorded is not None else []
        self.cursor = 0
        self.log: List[np.ndarray] = []

    def draw(self, strength: float, b: int, c: int) -> Optional[np.ndarray]:
        if self.mode == "off" or not strength:
            return None
        if self.mode == "list":
            r = self.recorded[self.cursor]
            self.cursor += 1
            assert r.shape == (b, c, 1), (r.shape, (b, c, 1))
        else:
            r = (1.0 + strength) ** np.random.normal(size=[b, c, 1])
        self.log.append(r)
        return r


def gdrop(x, strength: float, src: DropSource):
    """layers.py:32-38."""
    r = src.draw(strength, x.shape[0], x.shape[1])
    if r is None:
        return x
    return x * torch.from_numpy(r).to(x.dtype)


def layer_weight(params, prefix: str, training: bool = True, eps: float = 1e-12):
    """The weight of the conv at ``prefix`` (an nn.Conv1d's path).  With spectral norm (layers.py:170-171:
    torch.nn.utils.spectral_norm, training mode) the parameter is ``weight_orig`` and every call does one power
    iteration on the stored vectors, in place and without gradient -- v = normalize(W^T u), u = normalize(W v) on the
    weight as a matrix [output channels, rest] -- then divides by sigma = u^T W v (the vectors as constants)."""
    if f"{prefix}.weight_orig" not in params:
        return params[f"{prefix}.weight"]
    w, u, v = params[f"{prefix}.weight_orig"], params[f"{prefix}.weight_u"], params[f"{prefix}.weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
    u, v = u.detach().clone(), v.detach().clone()
    return w / torch.dot(u, torch.mv(wm, v))


def has_layer(params, prefix: str) -> bool:
    return f"{prefix}.weight" in params or f"{prefix}.weight_orig" in params


def conditional_batch_norm(x, params, scales, prefix: str, z=None, y=None):
    """ConditionalBatchNorm.forward in training mode, layers.py:117-134: statistics of the batch over (batch, time);
    'BN' mode (affine normaliser) when the layer has no modulation network, else a normaliser without affine
    parameters, then out + gamma * out + beta with (gamma, beta) = the two halves of the modulation network's output,
    repeated over time: 'CBN' a linear 1x1 layer on the class embedding y [B, E, 1]; 'SM' a two-layer 1x1 network
    (ReLU in between, layers.py:107-111) on the latent [B, latent, 1]; 'CSM' the same on [latent; y] stacked along the
    channels."""
    out = F.batch_norm(x, None, None, params.get(f"{prefix}.normalizer.weight"), params.get(f"{prefix}.normalizer.bias"),
                       training=True, eps=1e-5)
    if has_layer(params, f"{prefix}.embed.conv.conv"):  # 'CBN'
        e = general_conv(y, params, scales, f"{prefix}.embed", act_alpha=-1)
    elif has_layer(params, f"{prefix}.embed.0.conv.conv"):
        z3 = z.unsqueeze(2) if z.dim() == 2 else z
        cond = z3 if y is None else torch.cat([resample(z3, z3.shape[2], y.shape[2]), y], dim=1)  # 'SM' | 'CSM'
        e = general_conv(cond, params, scales, f"{prefix}.embed.0", act_alpha=0.0)
        e = general_conv(e, params, scales, f"{prefix}.embed.1", act_alpha=-1)
    else:
        return out
    e = resample(e, e.shape[2], out.shape[2])
    gamma, beta = e.chunk(2, dim=1)
    return out + gamma * out + beta


def general_conv(x, params, scales, path, *, pad=None, act_alpha=0.2, norm=None, drop=0.0,
                 src: Optional[DropSource] = None, z=None, y=None, conv_noise=None):
    """layers.py:207-213 for the default options: conv -> (* conv_noise) -> LeakyReLU ->
    (PixelNorm) -> (GDrop).  ``act_alpha < 0`` means linear."""
    if has_layer(params, f"{path}.conv.net.0.conv"):
        # separable layer (layers.py:137-147): a depthwise conv with its own scale and bias and the layer's padding,
        # then a 1x1 conv; activation, norm and multiplier follow the pair
        w0, b0 = layer_weight(params, f"{path}.conv.net.0.conv"), params[f"{path}.conv.net.0.conv.bias"]
        k = w0.shape[2]
        pad = (k - 1) // 2 if pad is None else pad
        h = F.conv1d(x * scales[f"{path}.conv.net.0"], w0, b0, padding=pad, groups=w0.shape[0])
        h = equalized_conv1d(h, layer_weight(params, f"{path}.conv.net.1.conv"), params[f"{path}.conv.net.1.conv.bias"],
                             scales[f"{path}.conv.net.1"], 0)
    else:
        w = layer_weight(params, f"{path}.conv.conv")
        b = params.get(f"{path}.conv.conv.bias")  # (absent: a layer built with bias=False)
        k = w.shape[2]
        pad = (k - 1) // 2 if pad is None else pad
        h = equalized_conv1d(x, w, b, scales[path], pad)
    if conv_noise is not None:  # layers.py:209-210
        h = h * conv_noise
    if norm == "batch":  # layers.py:211-212: between conv and activation
        h = conditional_batch_norm(h, params, scales, f"{path}.norm", z, y)
    if act_alpha >= 0:
        h = leaky_relu(h, act_alpha) if act_alpha > 0 else F.relu(h)
    if norm == "pixel":
        h = pixel_norm(h)
    if drop:
        h = gdrop(h, drop, src)
    return h


def self_attention(params, scales, prefix: str, x):
    """SelfAttention.forward, layers.py:56-64: linear 1x1 convs to query, key and value, key and value max-pooled by
    4 along time, the softmax taken over the (pooled) key positions, the result of the last 1x1 conv scaled by the
    scalar gamma and added to the input.  Returns (output, attention map [B, T/4, T])."""
    lin = dict(act_alpha=-1)
    query = general_conv(x, params, scales, f"{prefix}.query_conv", **lin)
    key = F.max_pool1d(general_conv(x, params, scales, f"{prefix}.key_conv", **lin), 4)
    value = F.max_pool1d(general_conv(x, params, scales, f"{prefix}.value_conv", **lin), 4)
    attention = torch.softmax(torch.bmm(key.permute(0, 2, 1), query), dim=1)
    out = general_conv(torch.bmm(value, attention), params, scales, f"{prefix}.final_conv", **lin)
    return params[f"{prefix}.gamma"] * out + x, attention


def linear_interp_upsample(x, p: int):
    """F.interpolate(scale_factor=p, mode='linear', align_corners=False)
    (utils.py:204-205) written out: output j samples the input at
    src = (j + 0.5) / p - 0.5 clamped below at 0, between floor(src) and
    min(floor(src) + 1, T - 1)."""
    T = x.shape[-1]
    j = torch.arange(T * p, dtype=x.dtype)
    src = torch.clamp((j + 0.5) / p - 0.5, min=0.0)
    i0 = torch.floor(src).long()
    i1 = torch.clamp(i0 + 1, max=T - 1)
    lam = (src - i0.to(x.dtype))
    return x[..., i0] * (1.0 - lam) + x[..., i1] * lam


def avg_pool_down(x, q: int):
    """F.avg_pool1d(k=q, stride=q, pad=0, ceil_mode=False,
    count_include_pad=True) (utils.py:208-209)."""
    T = x.shape[-1]
    n = T // q
    return x[..., : n * q].reshape(*x.shape[:-1], n, q).mean(dim=-1)


def resample(x, signal_freq: int, desired_freq: int):
    """utils.py:221-251 for 3-D torch input: ratio = desired/signal reduced to
    lowest terms; a length-1 signal is repeated; otherwise linear upsample by
    the numerator then average-pool by the denominator."""
    if x.shape[2] == 1:
        ratio = desired_freq / signal_freq
        assert ratio == int(ratio)
        return x.repeat(1, 1, int(ratio))
    r = Fraction(int(desired_freq), int(signal_freq))
    if r.numerator != 1:
        x = linear_interp_upsample(x, r.numerator)
    if r.denominator != 1:
        x = avg_pool_down(x, r.denominator)
    return x


def datapoint_from_depth(x, target_depth: int, cfg: "NetConfig" = None):
    """dataset.py:169-173: a full-rate window [B, C, T_max] taken down to stage ``target_depth``
    one stage transition at a time (ratio scale_down/scale_up of that transition)."""
    cfg = cfg or NetConfig()
    for index in range(len(cfg.scale_up) - 1, target_depth - 1, -1):
        x = resample(x, cfg.scale_up[index], cfg.scale_down[index])
    return x


def alpha_fade_reals(x, depth: int, alpha: float, cfg: "NetConfig" = None):
    """dataset.py:175-180: x + (t - x) * (1 - alpha), t = x taken down one stage and back up."""
    cfg = cfg or NetConfig()
    if alpha == 1:
        return x
    up, down = cfg.scale_up[depth - 1], cfg.scale_down[depth - 1]
    t = resample(resample(x, up, down), down, up)
    return x + (t - x) * (1 - alpha)


def minibatch_stddev(x, group_size: int = 4, window: int = 32):
    """layers.py:74-90: per window of ``window`` = kernel_size // temporal_groups_per_window time steps, samples are split into ``G=min(B, group_size)``
    interleaved groups (sample i sits with i + B/G, ...); the per-(member,
    channel, time) standard deviation over the group (no epsilon) is averaged
    over channels and time and appended as one extra channel."""
    B, C, T = x.shape
    G = min(B, group_size)
    cols = []
    for start in range(0, T, window):
        y = x[..., start:start + window]
        Tw = y.shape[-1]
        y = y.reshape(G, B // G, C, Tw)
        y = y - y.mean(dim=0, keepdim=True)
        sd = torch.sqrt((y * y).mean(dim=0))          # [B/G, C, Tw]
        s = sd.mean(dim=(1, 2), keepdim=True)          # [B/G, 1, 1]
        cols.append(s.repeat(G, 1, Tw))
    return torch.cat([x, torch.cat(cols, dim=2)], dim=1)


# --------------------------------------------------------------------------- networks
def g_block(params, scales, prefix: str, h, cfg: NetConfig, src: "DropSource", first: bool = False,
            last: bool = False, z=None, y=None):
    """GBlock.forward, network.py:206-218 (no residual, not deep): two GeneralConvs with pixel norm and
    GDrop, the first one with the full-length kernel and pad = k-1 in the first block (network.py:157-159);
    ``last`` appends to_rgb = linear 1x1 conv + ScaledTanh (network.py:173-176, 188-191; layers.py:17-23)."""
    ls = dict(act_alpha=cfg.act_alpha, norm=cfg.g_act_norm, drop=cfg.g_dropout, src=src, z=z, y=y)
    x = h

    def noise(i):
        """network.py:203-208: N(0, 1)[B, C, 1] from torch's global (host) generator times the learned weight
        ``noises.i`` [1, C, 1], drawn right before layer i runs; None without ``per_channel_noise``."""
        weight = params.get(f"{prefix}.noises.{i}")
        if weight is None:
            return None
        return torch.randn(x.shape[0], *weight.shape[1:]).to(weight.dtype) * weight

    first = first and not cfg.g_conv_only  # (network.py:35, 303: a fully convolutional generator has no full-length layer)
    h = general_conv(h, params, scales, f"{prefix}.convs.0", pad=cfg.initial_kernel_size - 1 if first else None,
                     conv_noise=noise(0), **ls)
    h = general_conv(h, params, scales, f"{prefix}.convs.1", conv_noise=noise(1), **ls)
    if cfg.deep:  # network.py:167-169, 213-216; PassChannelResidual (layers.py:216-223): the input's first channels
        h = general_conv(h, params, scales, f"{prefix}.convs.2", conv_noise=noise(2), **ls)
        h = general_conv(h, params, scales, f"{prefix}.convs.3", conv_noise=noise(3), **ls)
        h = x[:, :h.shape[1]] + h
    elif cfg.residual and not first:  # network.py:195-199, 217-218
        h = h + _skip(params, scales, f"{prefix}.residual", x)
    if last:
        return g_to_rgb(params, scales, prefix, h, cfg)
    return h


def _skip(params, scales, path, x):
    """The skip of a residual block: the identity when the block keeps the channel count, a linear 1x1 conv otherwise."""
    if has_layer(params, f"{path}.conv.conv"):
        return general_conv(x, params, scales, path, act_alpha=-1)
    return x


def g_to_rgb(params, scales, prefix, h, cfg: NetConfig):
    """to_rgb, network.py:173-191, then ScaledTanh unless ``no_tanh``.  'pggan': a linear 1x1 conv; 'sngan': a k = 3
    conv to the signal channels with LeakyReLU(0.2); 'sagan': a k = 3 conv with LeakyReLU(0.2) that keeps the channel
    count, then the linear 1x1 conv; 'biggan': BatchNorm1d (training mode), ReLU, a linear k = 3 conv."""
    base = f"{prefix}.to_rgb" if cfg.no_tanh else f"{prefix}.to_rgb.0"
    mode = cfg.to_rgb_mode
    if mode == "pggan":
        out = general_conv(h, params, scales, base, act_alpha=-1)
    elif mode == "sngan":
        out = general_conv(h, params, scales, base, act_alpha=0.2)
    elif mode == "sagan":
        out = general_conv(h, params, scales, f"{base}.0", act_alpha=0.2)
        out = general_conv(out, params, scales, f"{base}.1", act_alpha=-1)
    elif mode == "biggan":
        out = F.relu(F.batch_norm(h, None, None, params[f"{base}.0.weight"], params[f"{base}.0.bias"], training=True,
                                  eps=1e-5))
        out = general_conv(out, params, scales, f"{base}.2", act_alpha=-1)
    else:
        raise ValueError(mode)
    return out if cfg.no_tanh else torch.tanh(out * cfg.tanh_scale)


def d_block(params, scales, pos: int, h, cfg: NetConfig, src: "DropSource", first: bool = False):
    """DBlock.forward, network.py:441-449 (no residual, not deep): optional from_rgb (1x1 conv + LeakyReLU,
    network.py:428-429), then two GeneralConvs with GDrop; the last block of the list prepends
    MinibatchStddev and ends in the full-length kernel without padding (network.py:403-418)."""
    ls = dict(act_alpha=cfg.act_alpha, norm=cfg.d_act_norm, drop=cfg.d_dropout, src=src)
    if first:
        h = general_conv(h, params, scales, f"blocks.{pos}.from_rgb", act_alpha=cfg.act_alpha)
    if pos == cfg.num_stages:
        x = h
        base = 0  # (without MinibatchStddev, network.py:405-406, the convs are net.0, net.1, ...)
        if cfg.group_size >= 0:
            h = minibatch_stddev(h, cfg.group_size, cfg.initial_kernel_size // cfg.temporal_groups_per_window)
            base = 1
        n = base + (3 if cfg.deep else 1)  # (deep: two more layers in the middle, network.py:411-415)
        for i in range(base, n):
            h = general_conv(h, params, scales, f"blocks.{pos}.net.{i}", **ls)
        if not cfg.d_conv_only:
            return general_conv(h, params, scales, f"blocks.{pos}.net.{n}", pad=0, **ls)
        # fully convolutional critic (network.py:416-418, 431-438): a k-tap layer, and the block keeps its skip
        h = general_conv(h, params, scales, f"blocks.{pos}.net.{n}", **ls)
        if cfg.deep:
            skip = f"blocks.{pos}.residual.net"
            if has_layer(params, f"{skip}.conv.conv"):
                return h + torch.cat([x, general_conv(x, params, scales, skip, act_alpha=-1)], dim=1)
            return h + x
        if cfg.residual:
            h = h + _skip(params, scales, f"blocks.{pos}.residual", x)
        return h
    x = h
    h = general_conv(h, params, scales, f"blocks.{pos}.net.0", **ls)
    h = general_conv(h, params, scales, f"blocks.{pos}.net.1", **ls)
    if cfg.deep:  # ConcatResidual (layers.py:226-239)
        h = general_conv(h, params, scales, f"blocks.{pos}.net.2", **ls)
        h = general_conv(h, params, scales, f"blocks.{pos}.net.3", **ls)
        skip = f"blocks.{pos}.residual.net"
        if has_layer(params, f"{skip}.conv.conv"):
            return h + torch.cat([x, general_conv(x, params, scales, skip, act_alpha=-1)], dim=1)
        return h + x
    if cfg.residual:  # network.py:431-438, 447-448
        h = h + _skip(params, scales, f"blocks.{pos}.residual", x)
    return h


def generator_forward(params, scales, z, depth: int, alpha: float, cfg: NetConfig,
                      src: Optional[DropSource] = None, y=None):
    """network.py:354-396 (default options: pggan rgb mode, pixel norm,
    normalize_latents, no conditioning).  Returns x[B, 5, T(depth)]."""
    src = src or DropSource("off")

    def gblock(prefix, h, first=False):
        layer = -1 if first else int(prefix.split(".")[1])
        return g_block(params, scales, prefix, h, cfg, src, first=first, z=z_of(layer), y=y)

    def to_rgb(prefix, h):
        return g_to_rgb(params, scales, prefix, h, cfg)

    def attend(layer, h):  # network.py:324-330: before stage ``layer`` grows (never before the last stage, 383-390)
        if layer in cfg.self_attention_layers:
            return self_attention(params, scales, f"dummy.{list(cfg.self_attention_layers).index(layer)}", h)[0]
        return h

    if y is not None:  # network.py:355-361, 284-292: class scores [B, classes] -> the shared embedding [B, E, 1]
        y = y.unsqueeze(2) if y.dim() == 2 else y
        if has_layer(params, "y_encoder.conv.conv"):
            y = general_conv(y, params, scales, "y_encoder", act_alpha=-1)
    h = pixel_norm(z) if cfg.normalize_latents else z
    if h.dim() == 2:  # (network.py:364-367: a fully convolutional generator is given [B, latent, L])
        h = h.unsqueeze(2)
    whole = h
    if cfg.split_z:  # network.py:294-295, 368-371: the first share is the first layer's input
        share = cfg.latent_size // (cfg.num_stages + 2)
        h = whole[:, :share]

    def z_of(layer):  # network.py:317-322: what the batch norms of block ``layer`` see (block0: -1)
        if not cfg.z_to_bn:
            return None
        return whole[:, (2 + layer) * share:(3 + layer) * share] if cfg.split_z else whole

    h = gblock("block0", h, first=True)
    if cfg.rgb_generation_mode != "pggan":
        # network.py:332-349, 372-392: every stage's signal, each sum resampled to the next stage's length
        saved = [to_rgb("block0", h)]
        for i in range(depth):
            if i < depth - 1:
                h = attend(i, h)
            h = gblock(f"blocks.{i}", resample(h, cfg.scale_down[i], cfg.scale_up[i]))
            saved.append(to_rgb(f"blocks.{i}", h))
        total = saved[0]
        for rgb in saved[1:]:
            total = resample(total, total.shape[2], rgb.shape[2]) + rgb
        if cfg.rgb_generation_mode == "residual":
            return total if alpha == 1.0 else total - (1.0 - alpha) * saved[-1]
        total = total / len(saved)
        if alpha == 1.0:
            return total
        return (total * len(saved) - saved[-1]) / (len(saved) - 1) * (1.0 - alpha) + total * alpha
    if depth == 0:
        return to_rgb("block0", h)
    for i in range(depth - 1):
        h = resample(attend(i, h), cfg.scale_down[i], cfg.scale_up[i])
        h = gblock(f"blocks.{i}", h)
    h = resample(h, cfg.scale_down[depth - 1], cfg.scale_up[depth - 1])
    ult = to_rgb(f"blocks.{depth - 1}", gblock(f"blocks.{depth - 1}", h))
    if alpha == 1.0:
        return ult
    prev = f"blocks.{depth - 2}" if depth > 1 else "block0"
    return to_rgb(prev, h) * (1.0 - alpha) + ult * alpha


def discriminator_forward(params, scales, x, depth: int, alpha: float, cfg: NetConfig,
                          src: Optional[DropSource] = None, y=None):
    """network.py:528-560 (default options).  Returns (score[B], h[B, 256, 1])."""
    src = src or DropSource("off")
    R = cfg.num_stages

    def from_rgb(pos, v):
        return general_conv(v, params, scales, f"blocks.{pos}.from_rgb", act_alpha=cfg.act_alpha)

    def dblock(pos, h):
        return d_block(params, scales, pos, h, cfg, src)

    first = R - depth                      # blocks[-(depth+1)]
    h = dblock(first, from_rgb(first, x))
    x_low = None
    if depth > 0:
        up, dn = cfg.scale_up[depth - 1], cfg.scale_down[depth - 1]
        h = resample(h, up, dn)
        if alpha < 1.0 or cfg.input_to_all_layers:
            x_low = resample(x, up, dn)
            low = from_rgb(first + 1, x_low)
            if cfg.input_to_all_layers:  # network.py:537-538
                h = (h * alpha + low) / (1.0 + alpha)
            else:
                h = h * alpha + (1.0 - alpha) * low
    for i in range(depth, 0, -1):
        h = dblock(R + 1 - i, h)
        if i > 1:
            h = resample(h, cfg.scale_up[i - 2], cfg.scale_down[i - 2])
            if cfg.input_to_all_layers:  # network.py:546-549
                x_low = resample(x_low, cfg.scale_up[i - 2], cfg.scale_down[i - 2])
                h = (h + from_rgb(R + 2 - i, x_low)) / 2.0
        if (i - 2) in cfg.self_attention_layers:  # network.py:550-553
            h = self_attention(params, scales, f"dummy.{list(cfg.self_attention_layers).index(i - 2)}", h)[0]
    o = general_conv(h, params, scales, "linear", act_alpha=-1).mean(dim=2).squeeze()
    if y is not None:  # network.py:555-557: projection on a linear embedding of the classes
        w = layer_weight(params, "class_emb")
        o = o + ((y @ w.t()).unsqueeze(2) * h).mean(dim=2).sum(dim=1)
    return o, h


# --------------------------------------------------------------------------- losses
def d_loss_wgan_gp(dp, ds, gp_, gs, real, z, depth, alpha, cfg, mix, src_d, src_g,
                   drift_eps=1e-3, grad_lambda=10.0, target=1.0):
    """losses.py:73-120, ``wgan_gp`` branch.  ``mix`` is the per-sample mixing
    factor [B, 1, 1] (losses.py:13-18).  Call order of the stochastic layers:
    D(real), G(z) without graph, D(fake), D(x_hat)."""
    d_real, _ = discriminator_forward(dp, ds, real, depth, alpha, cfg, src_d)
    with torch.no_grad():
        fake = generator_forward(gp_, gs, z, depth, alpha, cfg, src_g)
    d_fake, _ = discriminator_forward(dp, ds, fake, depth, alpha, cfg, src_d)
    loss = d_fake.mean() - d_real.mean() + (d_real ** 2).mean() * drift_eps
    x_hat = (mix * real.detach() + (1.0 - mix) * fake.detach()).requires_grad_(True)
    pred, _ = discriminator_forward(dp, ds, x_hat, depth, alpha, cfg, src_d)
    g = torch.autograd.grad(pred, x_hat, torch.ones_like(pred), create_graph=True,
                            retain_graph=True)[0].reshape(real.shape[0], -1)
    pen = g.norm(p=2, dim=1) - target
    loss = loss + (pen ** 2).mean() * grad_lambda / (target ** 2)
    return loss, dict(d_real=d_real, d_fake=d_fake, fake=fake, x_hat=x_hat, pred=pred, gp_grad=g)


def g_loss_wgan(dp, ds, gp_, gs, z, depth, alpha, cfg, src_d, src_g):
    """losses.py:40-48,70 for ``wgan*``: -mean(D(G(z)))."""
    fake = generator_forward(gp_, gs, z, depth, alpha, cfg, src_g)
    d_fake, _ = discriminator_forward(dp, ds, fake, depth, alpha, cfg, src_d)
    return -d_fake.mean(), dict(fake=fake, d_fake=d_fake)


def critic_score_loss(loss_type, d_real, d_fake, drift_eps):
    """The score terms of losses.py:82-101 -> (loss, gain on the penalty or None for a gain of one)."""
    F = torch.nn.functional
    ones, zeros = torch.ones_like(d_real), torch.zeros_like(d_real)
    if loss_type == "hinge":
        return F.relu(1.0 - d_real).mean() + F.relu(1.0 + d_fake).mean(), None
    if loss_type == "rsgan":
        return F.binary_cross_entropy_with_logits(d_real - d_fake, ones), None
    if loss_type == "rasgan":
        return (F.binary_cross_entropy_with_logits(d_real - d_fake.mean(), ones)
                + F.binary_cross_entropy_with_logits(d_fake - d_real.mean(), zeros)) / 2.0, None
    if loss_type == "rahinge":
        return (torch.mean(F.relu(1.0 - (d_real - d_fake.mean())))
                + torch.mean(F.relu(1.0 + (d_fake - d_real.mean())))) / 2, None
    if loss_type == "wgan_theirs":
        f, r = d_fake.mean(), d_real.mean()
        return f - r + (f + r) ** 2 * drift_eps, F.relu(r - f)
    if loss_type == "wgan_gp":
        return d_fake.mean() - d_real.mean() + (d_real ** 2).mean() * drift_eps, None
    raise ValueError("Invalid loss type")


def d_loss_any(loss_type, dp, ds, gp_, gs, real, z, depth, alpha, cfg, mix, src_d, src_g,
               drift_eps=1e-3, grad_lambda=10.0, target=1.0):
    """losses.py:73-120 for every loss type.  Call order of the stochastic layers: D(real), G(z) without
    graph, D(fake), and D(x_hat) unless the penalty is skipped (grad_lambda == 0, or wgan_theirs' gain is zero)."""
    d_real, _ = discriminator_forward(dp, ds, real, depth, alpha, cfg, src_d)
    with torch.no_grad():
        fake = generator_forward(gp_, gs, z, depth, alpha, cfg, src_g)
    d_fake, _ = discriminator_forward(dp, ds, fake, depth, alpha, cfg, src_d)
    loss, gain = critic_score_loss(loss_type, d_real.reshape(-1), d_fake.reshape(-1), drift_eps)
    if grad_lambda == 0 or (gain is not None and float(gain) == 0.0):
        return loss, dict(d_real=d_real, d_fake=d_fake, fake=fake, gain=gain)
    x_hat = (mix * real.detach() + (1.0 - mix) * fake.detach()).requires_grad_(True)
    pred, _ = discriminator_forward(dp, ds, x_hat, depth, alpha, cfg, src_d)
    g = torch.autograd.grad(pred, x_hat, torch.ones_like(pred), create_graph=True,
                            retain_graph=True)[0].reshape(real.shape[0], -1)
    pen = g.norm(p=2, dim=1) - target
    if loss_type == "wgan_theirs":
        pen = torch.relu(pen)
    pen_loss = (pen ** 2).mean() * grad_lambda / (target ** 2)
    loss = loss + (pen_loss if gain is None else gain * pen_loss)
    return loss, dict(d_real=d_real, d_fake=d_fake, fake=fake, gain=gain, gp_grad=g)


def g_loss_any(loss_type, dp, ds, gp_, gs, real, z, depth, alpha, cfg, src_d, src_g, random_multiply=False,
               feature_matching_lambda=0.0):
    """losses.py:40-70.  Call order: G(z), D(fake), and D(real) without graph for the relativistic losses (or,
    after the loss, for the feature-matching term).  ``random_multiply`` draws its factor from Python's ``random``
    right after the critic's pass over the fakes, like the reference (losses.py:46)."""
    import random
    F = torch.nn.functional
    fake = generator_forward(gp_, gs, z, depth, alpha, cfg, src_g)
    d_fake, fake_features = discriminator_forward(dp, ds, fake, depth, alpha, cfg, src_d)
    d_fake = d_fake.reshape(-1)
    scale = random.random() if random_multiply else 1.0

    def finish(loss, info, real_features=None):
        if feature_matching_lambda != 0.0:  # losses.py:62-69 (the term's weight is one whatever the lambda)
            if real_features is None:
                with torch.no_grad():
                    _, real_features = discriminator_forward(dp, ds, real, depth, alpha, cfg, src_d)
            diff = real_features.mean(dim=0) - fake_features.mean(dim=0)
            loss = loss + (diff * diff).mean()
        return loss * scale, info

    if loss_type == "hinge" or loss_type.startswith("wgan"):
        return finish(-d_fake.mean(), dict(fake=fake, d_fake=d_fake))
    with torch.no_grad():
        d_real, real_features = discriminator_forward(dp, ds, real, depth, alpha, cfg, src_d)
    d_real = d_real.reshape(-1)
    ones, zeros = torch.ones_like(d_fake), torch.zeros_like(d_fake)
    if loss_type == "rsgan":
        loss = F.binary_cross_entropy_with_logits(d_fake - d_real, ones)
    elif loss_type == "rasgan":
        loss = (F.binary_cross_entropy_with_logits(d_fake - d_real.mean(), ones)
                + F.binary_cross_entropy_with_logits(d_real - d_fake.mean(), zeros)) / 2.0
    elif loss_type == "rahinge":
        loss = (torch.mean(F.relu(1.0 + (d_real - torch.mean(d_fake))))
                + torch.mean(F.relu(1.0 - (d_fake - torch.mean(d_real))))) / 2
    else:
        raise ValueError("Invalid loss type")
    return finish(loss, dict(fake=fake, d_fake=d_fake, d_real=d_real), real_features)


# --------------------------------------------------------------------------- schedule
MINIBATCH_OVERRIDE = {0: 256, 1: 256, 2: 128, 3: 128, 4: 48, 5: 32, 6: 32, 7: 32}  # plugins.py:29-30


def alpha_map(max_depth, training_kimg=400, transition_kimg=400):
    """plugins.py:66-76 with the class overrides (plugins.py:33-34)."""
    over = {1: 200, 2: 200, 3: 200, 4: 200}
    pts, ptr = [], 0
    for i in range(max_depth):
        ptr += int(over.get(i + 1, training_kimg) * 1000)
        pts.append(ptr)
        ptr += int(over.get(i + 1, transition_kimg) * 1000)
        pts.append(ptr)
    return pts


def calc_progress(cur_nimg, amap, max_depth):
    """plugins.py:78-97."""
    depth, alpha = 0, 1.0
    for i, point in enumerate(amap):
        if cur_nimg == point:
            break
        if cur_nimg > point and i % 2 == 0:
            depth += 1
        if cur_nimg < point and i % 2 == 1:
            alpha = (cur_nimg - amap[i - 1]) / (point - amap[i - 1])
            break
        if cur_nimg < point:
            break
    return min(max_depth, depth), alpha


# --------------------------------------------------------------------------- full step
@dataclass
class OracleTrainer:
    """One ``Trainer.train()`` (trainer.py:79-96): 5 D iterations, each with
    fresh reals and latents, then 1 G iteration; Adam(betas=(0, .99), eps=1e-8)
    (defaults.yml:61-64, train.py:131-136)."""
    dp: Dict[str, torch.Tensor]
    ds: Dict[str, float]
    gp: Dict[str, torch.Tensor]
    gs: Dict[str, float]
    cfg: NetConfig
    depth: int
    alpha: float = 1.0
    lr: float = 1e-3
    d_repeats: int = 5
    opt_d: Optional[torch.optim.Optimizer] = field(default=None)
    opt_g: Optional[torch.optim.Optimizer] = field(default=None)

    def __post_init__(self):
        for p in list(self.dp.values()) + list(self.gp.values()):
            p.requires_grad_(True)
        self.opt_d = torch.optim.Adam(list(self.dp.values()), self.lr, betas=(0.0, 0.99), eps=1e-8)
        self.opt_g = torch.optim.Adam(list(self.gp.values()), self.lr, betas=(0.0, 0.99), eps=1e-8)

    def d_iter(self, real, z, mix, src_d, src_g):
        self.opt_d.zero_grad(set_to_none=True)
        loss, _ = d_loss_wgan_gp(self.dp, self.ds, self.gp, self.gs, real, z, self.depth, self.alpha,
                                 self.cfg, mix, src_d, src_g)
        grads = torch.autograd.grad(loss, [p for p in self.dp.values()], allow_unused=True)
        for p, g in zip(self.dp.values(), grads):
            p.grad = g
        self.opt_d.step()
        return loss.detach()

    def g_iter(self, z, src_d, src_g):
        self.opt_g.zero_grad(set_to_none=True)
        loss, _ = g_loss_wgan(self.dp, self.ds, self.gp, self.gs, z, self.depth, self.alpha, self.cfg,
                              src_d, src_g)
        grads = torch.autograd.grad(loss, [p for p in self.gp.values()], allow_unused=True)
        for p, g in zip(self.gp.values(), grads):
            p.grad = g
        self.opt_g.step()
        return loss.detach()

    def step(self, reals, zs, mixes, src_d, src_g):
        """reals/mixes: d_repeats entries; zs: d_repeats + 1 entries."""
        dl = None
        for i in range(self.d_repeats):
            dl = self.d_iter(reals[i], zs[i], mixes[i], src_d, src_g)
        gl = self.g_iter(zs[self.d_repeats], src_d, src_g)
        return gl, dl


def synthetic_eeg(b: int, channels: int, t: int, rng: np.random.Generator, fs_full=60.0, t_full=1920):
    """Synthetic EEG-like reals: per channel a sum of 6 sinusoids with
    f ~ U(1, 30) Hz (sampled at fs = 60 * T / 1920 Hz), random phase, 1/f
    amplitude, plus N(0, 0.1^2) noise, min-max scaled to [-1, 1] per sample
    (same normalisation as dataset.py:139-144)."""
    fs = fs_full * t / t_full
    n = np.arange(t)[None, None, None, :] / fs
    f = rng.uniform(1.0, 30.0, size=(b, channels, 6, 1))
    ph = rng.uniform(0, 2 * np.pi, size=(b, channels, 6, 1))
    x = (np.sin(2 * np.pi * f * n + ph) / f).sum(axis=2) + rng.normal(0, 0.1, size=(b, channels, t))
    lo = x.min(axis=(1, 2), keepdims=True)
    hi = x.max(axis=(1, 2), keepdims=True)
    return ((x - lo) / (hi - lo) * 2.0 - 1.0).astype(np.float32)
