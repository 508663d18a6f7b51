"""``Generator`` / ``Discriminator`` (and their blocks) with the constructor signatures,
attributes and state-dict keys of the reference's ``network.py`` (GBlock 151-221,
Generator 224-396, DBlock 399-449, Discriminator 452-560), running on the HIP kernels.

Beside the default configuration these options run on the kernels: ``residual`` blocks (identity or
1x1-conv skips, network.py:195-199, 217-218, 435-438, 447-448), the generator's ``rgb_generation_mode``
'residual' / 'mean' (network.py:332-349), ``no_tanh``, ``normalize_latents=False``, ReLU layers
(``act_alpha=0``), non-equalized layers and the other initialisers, ``separable`` layers (layers.py:137-147), ``deep`` blocks (four layers through a bottleneck with channel-slice /
concatenation skips, network.py:157-169, 192-193, 407-418, 432-433) and
the critic's ``input_to_all_layers``
(network.py:533-549) and ``self_attention_layers`` (layers.py:41-64; network.py:304-310, 324-330, 503-509, 550-553), and in the
generator ``act_norm='batch'`` with or without ``z_to_bn`` (layers.py:93-134, 190-213; network.py:299, 317-322)
and ``split_z`` (network.py:270-272, 294-295, 317-322, 368-371) and the 'sngan' / 'sagan' / 'biggan' forms of
``to_rgb_mode`` (network.py:173-187), ``spectral`` in either network (spectral norm of every conv weight,
layers.py:170-171) and ``conv_only`` (fully convolutional networks: k-tap layers in place of the full-length ones, the
critic's score the mean over time; network.py:33-36, 364-365, 416-418, 554), ``act_norm='batch'`` / ``'pixel'`` in the critic
(differentiated twice under the gradient penalty) and ``num_classes`` (class scores y [B, classes] into the generator's
batch norms through a shared embedding, network.py:284-292, 355-361, and into the critic's projection term, 518-523,
555-557).  Any other option raises ``NotImplementedError`` in the constructor.  ``forward`` accepts
either a tensor or the dict the reference's losses pass (``{'z': ...}`` / ``{'x': ...}``).
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from . import layers, ops
from ._lib import ACT_NONE
from .layers import ConcatResidual, ConditionalBatchNorm, GeneralConv, MinibatchStddev, PassChannelResidual, ScaledTanh, SelfAttention
from .utils import pixel_norm, resample_signal


def _require(cond, what):
    if not cond:
        raise NotImplementedError(f"{what} is outside the accelerated default configuration")


class ToRgb(nn.Sequential):
    """[GeneralConv(k=1, linear), ScaledTanh]; runs as one conv kernel with a tanh epilogue."""

    def forward(self, x):
        return self[0](x, _tanh_scale=self[1].scale)


class NormReluConv(nn.Sequential):
    """[BatchNorm1d, ReLU, GeneralConv] (the 'biggan' to_rgb, network.py:183-185): the norm and the ReLU run as one
    kernel pair of the library; the ``nn.BatchNorm1d`` holds the parameters and running statistics."""

    def forward(self, x):
        bn = self[0]
        train = bn.training or bn.running_mean is None
        if train and bn.track_running_stats:
            bn.num_batches_tracked += 1
        h = ops.bn_act(x, None, bn.weight, bn.bias, None, None, bn.running_mean, bn.running_var, 0.0, bn.eps,
                       0.1 if bn.momentum is None else bn.momentum, train)
        return self[2](h)


class GBlock(nn.Module):
    def __init__(self, ch_in, ch_out, ch_rgb, k_size=3, initial_kernel_size=None, is_residual=False, no_tanh=False,
                 deep=False, per_channel_noise=False, to_rgb_mode="pggan", **layer_settings):
        super().__init__()
        first = initial_kernel_size is not None
        hidden = max((ch_in // 4) if deep else ch_out, 4)  # (network.py:157-169: four layers through a bottleneck)
        self.convs = nn.ModuleList([
            GeneralConv(ch_in, hidden, kernel_size=initial_kernel_size if first else k_size,
                        pad=initial_kernel_size - 1 if first else None, **layer_settings),
            GeneralConv(hidden, hidden if deep else ch_out, kernel_size=k_size, **layer_settings)])
        # (network.py:162-172) one learned weight per layer and channel, zero at construction, for the noise that
        # multiplies the layer's conv output; registered in the reference's order (state-dict keys ``noises.i``)
        widths = [hidden, hidden if deep else ch_out] + ([hidden, ch_out] if deep else [])
        self.noises = nn.ParameterList([nn.Parameter(torch.zeros(1, widths[0], 1)),
                                        nn.Parameter(torch.zeros(1, widths[1], 1))]) if per_channel_noise else None
        if deep:
            self.convs.append(GeneralConv(hidden, hidden, kernel_size=k_size, **layer_settings))
            self.convs.append(GeneralConv(hidden, ch_out, kernel_size=k_size, **layer_settings))
            if per_channel_noise:
                self.noises.append(nn.Parameter(torch.zeros(1, widths[2], 1)))
                self.noises.append(nn.Parameter(torch.zeros(1, widths[3], 1)))
        rgb_settings = {k: layer_settings[k] for k in ("equalized", "spectral", "init")}
        if to_rgb_mode == "pggan":  # (network.py:175-187)
            to_rgb = GeneralConv(ch_out, ch_rgb, kernel_size=1, act_alpha=-1, **rgb_settings)
        elif to_rgb_mode in ("sngan", "sagan"):
            to_rgb = GeneralConv(ch_out, ch_rgb if to_rgb_mode == "sngan" else ch_out, kernel_size=3, act_alpha=0.2,
                                 **rgb_settings)
            if to_rgb_mode == "sagan":
                to_rgb = nn.Sequential(to_rgb, GeneralConv(ch_out, ch_rgb, kernel_size=1, act_alpha=-1, **rgb_settings))
        elif to_rgb_mode == "biggan":
            to_rgb = NormReluConv(nn.BatchNorm1d(ch_out), nn.ReLU(),
                                  GeneralConv(ch_out, ch_rgb, kernel_size=3, act_alpha=-1, **rgb_settings))
        else:
            raise ValueError()
        if no_tanh:  # (network.py:188-191)
            self.to_rgb = to_rgb
        elif isinstance(to_rgb, GeneralConv) and to_rgb.act_alpha < 0:
            self.to_rgb = ToRgb(to_rgb, ScaledTanh())
        else:
            self.to_rgb = nn.Sequential(to_rgb, ScaledTanh())
        self.residual = None
        if deep:
            self.residual = PassChannelResidual()
        elif not first and is_residual:  # (network.py:195-197, constructed after to_rgb: same random stream)
            self.residual = nn.Identity() if ch_in == ch_out else GeneralConv(ch_in, ch_out, 1, act_alpha=-1,
                                                                              **rgb_settings)
        self.deep = deep

    @staticmethod
    def get_per_channel_noise(noise_weight, batch_size):
        """N(0, 1)[B, C, 1] * weight[1, C, 1] (network.py:203-205), or None without a weight.  The draw follows the
        layers' noise mode: from torch's host generator like the reference's (so equal seeds give equal draws), or
        on the device (no host round trip; graph-capturable).  The product is two tiny tensors: torch's own kernel."""
        if noise_weight is None:
            return None
        shape = (batch_size,) + tuple(noise_weight.shape[1:])
        if layers.get_drop_mode() == "numpy":
            draw = torch.randn(shape).to(noise_weight.device)
        else:
            draw = torch.randn(shape, device=noise_weight.device)
        return draw * noise_weight

    def forward(self, x, y=None, z=None, last=False):
        h = x
        for i, conv in enumerate(self.convs):
            weight = self.noises[i] if self.noises is not None else None
            h = conv(h, y=y, z=z, conv_noise=self.get_per_channel_noise(weight, x.size(0)))
        if self.deep:
            h = self.residual(h, x)
        elif self.residual is not None:
            h = ops.axpby(h, 1.0, self.residual(x), 1.0)
        return self.to_rgb(h) if last else h


class Generator(nn.Module):
    def __init__(self, initial_kernel_size, num_rgb_channels, fmap_base, fmap_max, fmap_min, kernel_size,
                 self_attention_layers, progression_scale_up, progression_scale_down, residual, separable, equalized,
                 init, act_alpha, num_classes, deep, z_distribution, spectral=False, latent_size=256, no_tanh=False,
                 per_channel_noise=False, to_rgb_mode="pggan", z_to_bn=False, split_z=False, dropout=0.2,
                 act_norm="pixel", conv_only=False, shared_embedding_size=32, normalize_latents=True,
                 rgb_generation_mode="pggan"):
        super().__init__()
        _require(num_classes == 0 or act_norm == "batch", "class conditioning without batch norm layers")
        _require(not conv_only or not split_z, "conv_only with split_z")
        _require(not z_to_bn or (act_norm == "batch" and not deep), "z_to_bn without batch norm layers, or in deep blocks")
        if rgb_generation_mode not in ("pggan", "residual", "mean"):
            raise ValueError("invalid rgb_generation_mode: {}".format(rgb_generation_mode))
        _require(act_norm in (None, "pixel", "batch"), f"act_norm={act_norm!r}")
        assert len(progression_scale_up) == len(progression_scale_down)
        stages = len(progression_scale_up)
        self.progression_scale_up = progression_scale_up
        self.progression_scale_down = progression_scale_down
        self.depth = 0
        self.alpha = 1.0
        # optional one-element device tensor mirroring alpha (kept current by graphs.GraphedTrainer)
        self.alpha_dev = None
        self.split_z = bool(split_z) and not deep  # (network.py:270-272)
        self.z_distribution = z_distribution
        # (network.py:33-36, 274, 303, 364-365: a fully convolutional generator takes latents [B, latent, L] and starts
        # with an ordinary k-tap layer instead of the full-length one)
        self.conv_only = bool(conv_only)
        self.initial_kernel_size = initial_kernel_size
        self.normalize_latents = normalize_latents
        self.z_to_bn = bool(z_to_bn)

        def nf(stage):
            return min(max(int(fmap_base / (2.0 ** stage)), fmap_min), fmap_max)

        if latent_size is None:
            latent_size = nf(0)
        self.input_latent_size = latent_size
        # (network.py:284-292) the classes enter every batch norm through one shared linear embedding
        if num_classes != 0:
            if shared_embedding_size > 0:
                self.y_encoder = GeneralConv(num_classes, shared_embedding_size, kernel_size=1, equalized=False,
                                             act_alpha=-1, spectral=False, bias=False)
                num_classes = shared_embedding_size
            else:
                self.y_encoder = nn.Identity()
        else:
            self.y_encoder = None
        if self.split_z:  # network.py:294-295: one share of the latent per block plus one for the first layer's input
            latent_size //= stages + 2
        self.latent_size = latent_size
        block_settings = dict(ch_rgb=num_rgb_channels, k_size=kernel_size, is_residual=residual, deep=deep,
                              no_tanh=no_tanh, per_channel_noise=per_channel_noise, to_rgb_mode=to_rgb_mode)
        layer_settings = dict(z_to_bn_size=latent_size if z_to_bn else 0, equalized=equalized, spectral=spectral,
                              init=init, act_alpha=act_alpha, do=dropout, num_classes=num_classes, act_norm=act_norm,
                              bias=True, separable=separable)
        self.block0 = GBlock(latent_size, nf(1), initial_kernel_size=None if conv_only else initial_kernel_size,
                             **block_settings, **layer_settings)
        self.self_attention = dict()  # (network.py:304-310: stage -> layer; registered through ``dummy``)
        attention = [SelfAttention(nf(layer + 1), spectral, init) for layer in self_attention_layers or ()]
        self.self_attention.update(zip(self_attention_layers or (), attention))
        if attention:
            self.dummy = nn.ModuleList(attention)
        self.blocks = nn.ModuleList(
            [GBlock(nf(i + 1), nf(i + 2), **block_settings, **layer_settings) for i in range(stages)])
        self.max_depth = len(self.blocks)
        self.deep = deep
        self.rgb_generation_mode = rgb_generation_mode

    def _grow(self, h, stage, maps=None):
        if maps is not None and stage in self.self_attention:  # (network.py:324-330: before the stage grows)
            h, maps[stage] = self.self_attention[stage](h)
        return resample_signal(h, self.progression_scale_down[stage], self.progression_scale_up[stage])

    def _combine_rgbs(self, saved):
        """The signals of all stages so far, each resampled to the next one's length and added (network.py:332-349):
        'residual' = their sum, 'mean' = their mean; during a fade-in the newest stage's signal enters with weight
        alpha."""
        total = saved[0]
        for rgb in saved[1:]:
            total = ops.axpby(resample_signal(total, total.shape[2], rgb.shape[2]), 1.0, rgb, 1.0)
        n, alpha = len(saved), float(self.alpha)
        if self.rgb_generation_mode == "residual":
            return total if alpha == 1.0 else ops.axpby(total, 1.0, saved[-1], -(1.0 - alpha))
        if alpha == 1.0:
            return ops.axpby(total, 1.0 / n, None, 0.0) if n > 1 else total
        # (mean * n - newest) / (n - 1) * (1 - alpha) + mean * alpha, with mean = total / n
        ca = (1.0 - alpha) / (n - 1) + alpha / n
        return ops.axpby(total, ca, saved[-1], -(1.0 - alpha) / (n - 1))

    def forward(self, z, y=None):
        if isinstance(z, dict):
            z, y = z["z"], z.get("y", y)
        if y is not None:  # (network.py:355-361) class scores [B, classes] -> the shared embedding [B, E, 1]
            if self.y_encoder is None:
                y = None
            else:
                y = self.y_encoder(y.unsqueeze(2) if y.dim() == 2 else y)
        if self.normalize_latents:
            z = pixel_norm(z)
        if self.conv_only and z.dim() != 3:
            raise ValueError("a conv_only generator takes latents [B, latent, L]")
        h = z.unsqueeze(2) if z.dim() == 2 else z
        if self.split_z:
            # network.py:317-322, 368-371: the first share feeds the first layer, share 2 + l the batch norms of
            # block l (block0: l = -1); without z_to_bn the other shares are unused
            share, whole = self.latent_size, h
            h = whole[:, :share].contiguous()
            zs = (lambda l: whole[:, (2 + l) * share:(3 + l) * share].contiguous()) if self.z_to_bn else (lambda l: None)
        else:
            zc = h if self.z_to_bn else None  # the (normalised) latent, for the layers' batch norms
            zs = lambda l: zc  # noqa: E731
        d = self.depth
        maps = {}  # attention maps by stage; the newest stage never gets the layer (network.py:383-390)
        if self.rgb_generation_mode != "pggan":
            saved = []
            h = self.block0(h, y, zs(-1))
            saved.append(self.block0.to_rgb(h))
            for i in range(d):
                h = self.blocks[i](self._grow(h, i, maps if i < d - 1 else None), y, zs(i))
                saved.append(self.blocks[i].to_rgb(h))
            return {"x": self._combine_rgbs(saved), "y": y}, maps
        if d == 0:
            return {"x": self.block0(h, y, zs(-1), last=True), "y": y}, maps
        h = self.block0(h, y, zs(-1))
        for i in range(d - 1):
            h = self.blocks[i](self._grow(h, i, maps), y, zs(i))
        h = self._grow(h, d - 1)
        ult = self.blocks[d - 1](h, y, zs(d - 1), last=True)
        if self.alpha == 1.0:
            return {"x": ult, "y": y}, maps
        prev = self.blocks[d - 2] if d > 1 else self.block0
        alpha = self.alpha_dev if self.alpha_dev is not None else self.alpha
        return {"x": ops.blend(ult, prev.to_rgb(h), alpha), "y": y}, maps


class DBlock(nn.Module):
    def __init__(self, ch_in, ch_out, ch_rgb, sample_rate, k_size=3, initial_kernel_size=None, is_residual=False,
                 deep=False, group_size=4, temporal_groups_per_window=1, conv_disc=False, sinc=False,
                 **layer_settings):
        super().__init__()
        _require(not sinc, "DBlock(sinc)")
        last = initial_kernel_size is not None
        stats = last and group_size >= 0
        full_length = last and not conv_disc  # (network.py:416-418: a fully convolutional critic keeps k-tap layers)
        hidden = max((ch_out // 4) if deep else ch_in, 4)  # (network.py:407-418)
        net = []
        if stats:
            net.append(MinibatchStddev(group_size, temporal_groups_per_window, initial_kernel_size))
        net.append(GeneralConv(ch_in + (1 if stats else 0), hidden, kernel_size=k_size, **layer_settings))
        if deep:
            net.append(GeneralConv(hidden, hidden, kernel_size=k_size, **layer_settings))
            net.append(GeneralConv(hidden, hidden, kernel_size=k_size, **layer_settings))
        net.append(GeneralConv(hidden, ch_out, kernel_size=initial_kernel_size if full_length else k_size,
                               pad=0 if full_length else None, **layer_settings))
        self.net = nn.Sequential(*net)
        rgb_settings = {k: layer_settings[k] for k in ("equalized", "spectral", "init")}
        self.from_rgb = GeneralConv(ch_rgb, ch_in, kernel_size=1, act_alpha=layer_settings["act_alpha"],
                                    **rgb_settings)
        self.residual = None
        if not full_length and deep:  # (network.py:431-438: the block with the full-length layer has no skip)
            self.residual = ConcatResidual(ch_in, ch_out, **rgb_settings)
        elif not full_length and is_residual:
            self.residual = nn.Identity() if ch_in == ch_out else GeneralConv(ch_in, ch_out, kernel_size=1,
                                                                              act_alpha=-1, **rgb_settings)
        self.deep = deep

    def forward(self, x, first=False):
        if first:
            x = self.from_rgb(x)
        h = self.net(x)
        if self.deep and self.residual is not None:
            return self.residual(h, x)
        if self.residual is not None:
            h = ops.axpby(h, 1.0, self.residual(x), 1.0)
        return h


class Discriminator(nn.Module):
    def __init__(self, initial_kernel_size, num_rgb_channels, fmap_base, fmap_max, fmap_min, kernel_size,
                 self_attention_layers, progression_scale_up, progression_scale_down, residual, separable, equalized,
                 init, act_alpha, num_classes, deep, spectral=False, dropout=0.2, act_norm=None, group_size=4,
                 temporal_groups_per_window=1, conv_only=False, input_to_all_layers=False, initial_sampling_rate=1.0,
                 sinc=False):
        super().__init__()
        _require(not sinc, "sinc")
        _require(act_norm in (None, "batch", "pixel"), f"act_norm={act_norm!r} in the discriminator")
        assert len(progression_scale_up) == len(progression_scale_down)
        stages = len(progression_scale_up)
        self.progression_scale_up = progression_scale_up
        self.progression_scale_down = progression_scale_down
        self.depth = 0
        self.alpha = 1.0
        # optional one-element device tensor mirroring alpha (kept current by graphs.GraphedTrainer)
        self.alpha_dev = None
        self.input_to_all_layers = input_to_all_layers

        def nf(stage):
            return min(max(int(fmap_base / (2.0 ** stage)), fmap_min), fmap_max)

        layer_settings = dict(equalized=equalized, spectral=spectral, init=init, act_alpha=act_alpha, do=dropout,
                              num_classes=0, act_norm=act_norm, bias=True, separable=separable)
        block_settings = dict(ch_rgb=num_rgb_channels, k_size=kernel_size, is_residual=residual, conv_disc=conv_only,
                              group_size=group_size, temporal_groups_per_window=temporal_groups_per_window, deep=deep,
                              sinc=sinc)
        # the coarsest block is constructed first (it draws its initial weights first), then the
        # others from fine to coarse, which end up at list positions 0 .. stages-1
        last_block = DBlock(nf(1), nf(0), initial_kernel_size=initial_kernel_size,
                            sample_rate=initial_sampling_rate, **block_settings, **layer_settings)
        up, down = np.array(progression_scale_up), np.array(progression_scale_down)
        lens = [initial_kernel_size * np.sum(up[:i] / down[:i]) for i in range(stages + 1)]
        self.self_attention = dict()  # (network.py:503-509)
        attention = [SelfAttention(nf(layer + 1), spectral, init) for layer in self_attention_layers or ()]
        self.self_attention.update(zip(self_attention_layers or (), attention))
        if attention:
            self.dummy = nn.ModuleList(attention)
        self.blocks = nn.ModuleList(
            [DBlock(nf(i + 2), nf(i + 1), sample_rate=lens[i + 1] * initial_sampling_rate / initial_kernel_size,
                    **block_settings, **layer_settings) for i in range(stages - 1, -1, -1)] + [last_block])
        # (network.py:518-523) projection critic: a linear embedding of the classes, multiplied into the features
        self.class_emb = None
        if num_classes != 0:
            self.class_emb = nn.Linear(num_classes, nf(0), False)
            if spectral:
                self.class_emb = nn.utils.spectral_norm(self.class_emb)
        self.linear = GeneralConv(nf(0), 1, kernel_size=1, equalized=equalized, act_alpha=-1, spectral=spectral,
                                  init=init)
        self.max_depth = len(self.blocks) - 1
        for m in self.modules():  # (the layers' norms are differentiated twice under the gradient penalty)
            if isinstance(m, (ConditionalBatchNorm, GeneralConv)):
                m.second_order = True

    def _shrink(self, h, stage):
        return resample_signal(h, self.progression_scale_up[stage], self.progression_scale_down[stage])

    def forward(self, x, y=None):
        """``x``: a batch (tensor or {'x': tensor}) -> (scores, features, {})."""
        if isinstance(x, dict):
            x, y = x["x"], x.get("y", y)
        d = self.depth
        maps = {}
        h = self.blocks[-(d + 1)](x, True)
        x_low = None
        if d > 0:
            h = self._shrink(h, d - 1)
            if self.input_to_all_layers:
                # every stage also sees the (down-sampled) input through its own from_rgb (network.py:533-549)
                x_low = self._shrink(x, d - 1)
                a = float(self.alpha)
                h = ops.axpby(h, a / (1.0 + a), self.blocks[-d].from_rgb(x_low), 1.0 / (1.0 + a))
            elif self.alpha < 1.0:
                low = self.blocks[-d].from_rgb(self._shrink(x, d - 1))
                alpha = self.alpha_dev if self.alpha_dev is not None else self.alpha
                h = ops.blend(h, low, alpha)
        for i in range(d, 0, -1):
            h = self.blocks[-i](h)
            if i > 1:
                h = self._shrink(h, i - 2)
                if self.input_to_all_layers:
                    x_low = self._shrink(x_low, i - 2)
                    h = ops.axpby(h, 0.5, self.blocks[-i + 1].from_rgb(x_low), 0.5)
            if (i - 2) in self.self_attention:  # (network.py:550-553)
                h, maps[i] = self.self_attention[i - 2](h)
        o = self.linear(h)
        if o.shape[2] != 1:  # (network.py:554) the mean over time, as a product with a constant vector
            ones = o.new_ones((o.shape[0], o.shape[2], 1))
            o = ops.bmm(o, ones, alpha=1.0 / o.shape[2])
        o = o.reshape(o.shape[0])
        if y is not None:
            # (network.py:555-557) + sum_c emb[b, c] * mean_t h[b, c, t]: the embedding as a 1x1 layer on the class
            # scores, the sum as a batched product (both twice differentiable)
            if self.class_emb is None:
                raise ValueError("this critic was built without classes")
            emb_w = self.class_emb
            if hasattr(emb_w, "weight_orig"):
                w, _ = ops.spectral_norm(emb_w.weight_orig, emb_w.weight_u, emb_w.weight_v, 1e-12, self.training)
            else:
                w = emb_w.weight
            emb = ops.conv1d(y.reshape(y.shape[0], -1, 1).contiguous(), w.unsqueeze(2), None, None, 1.0, ACT_NONE)
            proj = ops.bmm(emb, h, trans_a=True, alpha=1.0 / h.shape[2])  # [B, 1, T]
            if proj.shape[2] != 1:
                proj = ops.bmm(proj, proj.new_ones((proj.shape[0], proj.shape[2], 1)))
            o = ops.axpby(o, 1.0, proj.reshape(proj.shape[0]), 1.0)
        return o, h, maps
