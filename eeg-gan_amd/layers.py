"""Layer modules with the names, constructor signatures and state-dict layout of the reference's
``layers.py``; every forward runs fused HIP kernels (see ``ops.py``).

Supported: the options the default configuration uses (reference defaults.yml:16-58) plus ReLU / linear layers,
non-equalized layers, the other initialisers, separable layers (a depthwise conv followed by a 1x1 conv,
layers.py:137-147), self attention (layers.py:41-64), batch norm plain, modulated by the latent or conditioned on a
class embedding (layers.py:93-134), spectral norm (layers.py:170-171) and MinibatchStddev over several temporal
windows (layers.py:81-89).  What is left out (strided convs, GDrop axes other than (0, 1)) raises
``NotImplementedError`` at construction: there is no second code path.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn
from torch.nn.init import calculate_gain

from . import ops
from ._lib import ACT_LRELU, ACT_NONE
from .utils import pixel_norm

# How GDropLayer draws its multipliers:
#   'numpy'  - from the global numpy RNG on the host, fp64, exactly like the reference
#              (layers.py:32-38): identical seeds give identical multipliers;
#   'device' - N(0,1) drawn on the device (no host round trip; graph-capturable); same
#              distribution, different stream.
_drop_mode = "numpy"


def set_drop_mode(mode: str) -> None:
    global _drop_mode
    if mode not in ("numpy", "device"):
        raise ValueError(mode)
    _drop_mode = mode


def get_drop_mode() -> str:
    return _drop_mode


_segments = 1


class batch_segments:
    """Context manager: inside it the batch given to the networks is a concatenation of ``n`` independent
    minibatches (e.g. reals and fakes stacked into one critic pass).  Only MinibatchStddev couples
    samples of a batch, so only it needs to know (reference layers.py:67-90 groups within ONE minibatch)."""

    def __init__(self, n: int):
        self.n = int(n)

    def __enter__(self):
        global _segments
        self.prev, _segments = _segments, self.n
        return self

    def __exit__(self, *exc):
        global _segments
        _segments = self.prev
        return False


class _DropPool:
    """Device-mode multipliers are drawn CHUNK values at a time (one launch of the library's counter-based
    generator) and handed out as slices, instead of two tiny launches per layer and pass.  A chunk is never
    written again after it is filled, so slices saved for backward stay valid; ``reset`` drops the
    current chunk so that an iteration (and a captured graph of it) starts with a fresh draw.

    The generator's state (epochs the kernel itself advances) lives in device memory, one row of a zeroed table
    per user.  Eager draws share a row per (device, stream): launches on one stream run one after the other.  Every
    draw recorded during a graph capture gets a row of its own: the graphs of one iteration are captured on one
    stream but replayed side by side on several, and a replay draws a new stream of values each time because its
    row's epoch has moved on.  Tables are allocated outside captures only (memory allocated during a capture belongs
    to the graph, and zeroing it there would be replayed too): an eager draw makes sure enough rows are left for
    the captures that follow it, so the first draw on a device must be an eager one -- the trainers' warm-up runs."""

    CHUNK = 1 << 18  # one critic iteration at the final stage takes ~190 000 multipliers
    ROWS = 1024      # per table; 1 KB a row
    SPARE = 128      # rows an eager draw leaves for captures

    def __init__(self):
        self._chunks = {}
        self._tables = {}  # device -> [tables]: earlier ones stay alive, captured graphs hold their addresses
        self._used = {}    # device -> rows handed out of the newest table
        self._eager = {}   # (device, stream) -> (row tensor, salt)
        self._salt = 0

    def reset(self):
        self._chunks.clear()

    def _new_row(self, device):
        dkey = str(device)
        capturing = torch.cuda.is_current_stream_capturing()
        tables = self._tables.setdefault(dkey, [])
        if not tables or self._used[dkey] + (1 if capturing else self.SPARE) > self.ROWS:
            if capturing:
                raise RuntimeError("device-mode GDrop: no generator state left for a draw inside a graph capture; "
                                   "draw once outside the capture first (a warm-up run)")
            tables.append(torch.zeros((self.ROWS, ops.LOGNORMAL_STATE_WORDS), device=device, dtype=torch.int32))
            self._used[dkey] = 0
        row = tables[-1][self._used[dkey]]
        self._used[dkey] += 1
        self._salt += 1
        return row, self._salt

    def _state(self, device):
        if torch.cuda.is_current_stream_capturing():
            return self._new_row(device)
        key = (str(device), torch.cuda.current_stream().cuda_stream)
        ent = self._eager.get(key)
        if ent is None:
            ent = self._eager[key] = self._new_row(device)
        elif self._used[key[0]] + self.SPARE > self.ROWS:
            self._new_row(device)  # (tops the spare rows up with a new table; the row itself goes unused)
        return ent

    def take(self, n, base, device):
        # a chunk belongs to the stream that drew it: passes that run side by side on two streams draw their own
        key = (str(device), float(base), torch.cuda.current_stream().cuda_stream)
        ent = self._chunks.get(key)
        n4 = (n + 3) & ~3
        if ent is None or ent[1] + n4 > ent[0].numel():
            state, salt = self._state(device)
            buf = torch.empty(max(self.CHUNK, n4), device=device, dtype=torch.float32)
            ent = self._chunks[key] = [ops.lognormal_fill_raw(buf, base, torch.initial_seed(), salt, state), 0]
        out = ent[0][ent[1]:ent[1] + n]
        ent[1] += n4
        return out


_drop_pool = _DropPool()


_forced_masks = None


class forced_masks:
    """Context manager for parity tests: inside it the LeakyReLU layers of the critic's default form (conv ->
    LeakyReLU -> GDrop, reference layers.py:196-213) take the branch of every output from the next tensor of
    ``masks`` (its sign; one [B, Cout, T] tensor per layer call, in call order) instead of from their own
    pre-activations.  With the branches of a reference run, the layers compute the reference's piecewise-linear
    function exactly, so gradients can be compared without the noise of branches that flip under rounding."""

    def __init__(self, masks):
        self.masks = iter(masks)

    def __enter__(self):
        global _forced_masks
        self.prev, _forced_masks = _forced_masks, self
        return self

    def __exit__(self, *exc):
        global _forced_masks
        _forced_masks = self.prev
        return False

    def next_for(self, shape):
        m = next(self.masks)
        if tuple(m.shape) != tuple(shape):
            raise RuntimeError(f"forced_masks: the next mask has shape {tuple(m.shape)}, the layer's output {tuple(shape)}")
        return m.to(ops._act_dtype(shape[1], shape[2]))  # (the storage type of this layer's output)


def drop_pool_reset() -> None:
    _drop_pool.reset()


class PixelNorm(nn.Module):
    def forward(self, x):
        return pixel_norm(x)


class ScaledTanh(nn.Tanh):
    """tanh(scale * x).  On the path it is always the epilogue of a to_rgb conv and is fused there."""

    def __init__(self, scale=0.5):
        super().__init__()
        self.scale = scale

    def forward(self, x):
        return ops.scaled_tanh(x, self.scale)


class GDropLayer(nn.Module):
    """Multiplicative log-normal noise shared over time: x * (1+strength)**N(0,1)[B, C, 1]."""

    def __init__(self, strength=0.2, axes=(0, 1)):
        super().__init__()
        self.strength = strength
        self.axes = [axes] if isinstance(axes, int) else list(axes)
        if sorted(self.axes) != [0, 1]:
            raise NotImplementedError("GDropLayer: only per-(sample, channel) noise (axes=(0, 1)) is supported")

    def sample(self, batch, channels, device):
        """Multipliers [B, C] (float32, on `device`) or None when the layer is off."""
        if not self.strength:
            return None
        if _drop_mode == "numpy":
            rnd = (1 + self.strength) ** np.random.normal(size=[batch, channels, 1])
            return torch.from_numpy(rnd.astype(np.float32).reshape(batch, channels)).to(device)
        return _drop_pool.take(batch * channels, 1.0 + self.strength, device).view(batch, channels)

    def forward(self, x, deterministic=False):
        if deterministic or not self.strength:
            return x
        r = self.sample(x.shape[0], x.shape[1], x.device)
        return ops._MaskMul.apply(x, x, r, 1.0)


class MinibatchStddev(nn.Module):
    def __init__(self, group_size=4, temporal_groups_per_window=1, kernel_size=32):
        super().__init__()
        self.group_size = group_size if group_size != 0 else 1e6
        self.kernel_size = kernel_size
        self.stride_size = self.kernel_size // temporal_groups_per_window

    def forward(self, x):
        if self.group_size < 0:
            return x
        per_segment = x.shape[0] // _segments
        # statistics per temporal window of ``stride_size`` steps (layers.py:81-89); the default is one window
        return ops.minibatch_stddev(x, int(min(per_segment, self.group_size)), _segments, self.stride_size)


class ConditionalBatchNorm(nn.Module):
    """Batch normalisation over (batch, time) per channel, optionally modulated per sample (reference
    layers.py:93-134).  Modes: 'BN' (no conditioning: an affine ``nn.BatchNorm1d``'s arithmetic); 'CBN' (a normaliser
    without affine parameters, then ``out + gamma * out + beta`` with gamma / beta from a linear 1x1 layer on the
    class embedding); 'SM' / 'CSM' (gamma / beta from a two-layer 1x1 network on the latent, or on the latent and the
    class embedding stacked along the channels).  ``normalizer`` is the parameter and running-statistics container
    (same state-dict keys); training / evaluation follows ``self.training``.  The conditioning inputs have one time
    step (per-sample modulation).  First-order differentiable unless ``second_order`` is set."""

    def __init__(self, num_features, num_classes, latent_size, spectral):
        super().__init__()
        no_cond = latent_size == 0 and num_classes == 0
        self.num_features = num_features
        self.normalizer = nn.BatchNorm1d(num_features, affine=no_cond)
        if no_cond:
            self.mode = "BN"
        elif latent_size == 0:
            self.embed = GeneralConv(num_classes, num_features * 2, kernel_size=1, equalized=False, act_alpha=-1,
                                     spectral=spectral, bias=False)
            self.mode = "CBN"
        else:
            self.embed = nn.Sequential(
                GeneralConv(latent_size + num_classes, num_features * 2, kernel_size=1, equalized=False, act_alpha=0.0,
                            spectral=spectral, bias=True),
                GeneralConv(num_features * 2, num_features * 2, kernel_size=1, equalized=False, act_alpha=-1,
                            spectral=spectral, bias=False))
            self.mode = "SM" if num_classes == 0 else "CSM"

    def _conditioning(self, z, y):
        """The modulation network's input (layers.py:121-130): the class embedding ('CBN'), the latent ('SM') or the
        latent's channels followed by the embedding's, the latent resampled to the embedding's length ('CSM');
        [B, channels, steps] with one step for per-sample conditioning."""
        as3 = lambda t: t.unsqueeze(2) if t.dim() == 2 else t  # noqa: E731
        if self.mode in ("CBN", "CSM"):
            if y is None:
                raise ValueError(f"ConditionalBatchNorm in '{self.mode}' mode needs the class embedding")
            y = as3(y)
        if self.mode in ("SM", "CSM"):
            if z is None:
                raise ValueError(f"ConditionalBatchNorm in '{self.mode}' mode needs the latent")
            z = as3(z)
        if self.mode == "CBN":
            return y
        if self.mode == "SM":
            return z
        from .utils import resample_signal
        z = resample_signal(z.contiguous(), z.shape[2], y.shape[2])
        nz, ny, steps = z.shape[1], y.shape[1], y.shape[2]
        return ops.channel_window(y, ops.channel_window(z, None, 0, 0, nz, channels=nz + ny, length=steps), 0, nz, ny)

    def modulation(self, z, y=None, length=None):
        """(gamma, beta) from the latent [B, latent(, steps)] and / or the class embedding [B, classes(, steps)];
        (None, None) in 'BN' mode.  Per-sample conditioning (one step) gives [B, num_features] each; conditioning
        with a time axis is resampled to ``length`` steps, the layer's (layers.py:131-133), and gives
        [B, num_features, length] each."""
        if self.mode == "BN":
            return None, None
        e = self.embed(self._conditioning(z, y))  # [B, 2 F, steps]
        n, steps = self.num_features, e.shape[2]
        if steps == 1:
            gamma = ops.channel_window(e, None, 0, 0, n, channels=n, length=1).reshape(e.shape[0], n)
            beta = ops.channel_window(e, None, n, 0, n, channels=n, length=1).reshape(e.shape[0], n)
            return gamma, beta
        from .utils import resample_signal
        if length is None:
            raise ValueError("ConditionalBatchNorm: conditioning with a time axis needs the layer's length")
        e = resample_signal(e, steps, length)
        return (ops.channel_window(e, None, 0, 0, n, channels=n, length=length),
                ops.channel_window(e, None, n, 0, n, channels=n, length=length))

    # the critic's norms are differentiated twice (gradient penalty): the Discriminator sets this, and the layer then
    # runs as a chain of multilinear pieces instead of the fused first-order kernel pair
    second_order = False

    def apply_with(self, c, z, drop, slope, y=None):
        """drop * lrelu_slope(this norm of c): the epilogue GeneralConv asks for (slope 1 and no drop: the norm alone)."""
        bn = self.normalizer
        train = self.training or bn.running_mean is None
        momentum = 0.1 if bn.momentum is None else bn.momentum
        gmod, bmod = self.modulation(z, y, length=c.shape[2])
        if train and bn.track_running_stats:
            bn.num_batches_tracked += 1
        if gmod is not None and gmod.dim() == 3:
            # conditioning with a time axis (layers.py:131-134): the norm alone, then out + gamma * out + beta with
            # full-size gamma / beta, then the layer's activation and multipliers
            if self.second_order:
                raise NotImplementedError("a modulated batch norm that is differentiated twice")
            o = ops.bn_act(c, None, bn.weight, bn.bias, None, None, bn.running_mean, bn.running_var, 1.0, bn.eps,
                           momentum, train)
            o = ops.axpby(ops.axpby(o, 1.0, ops._Product.apply(o, gmod), 1.0), 1.0, bmod, 1.0)
            return o if (drop is None and slope == 1.0) else ops.lrelu_drop(o, drop, slope)
        if self.second_order:
            if self.mode != "BN":
                raise NotImplementedError("a latent-modulated batch norm that is differentiated twice")
            o = ops.batch_norm_twice(c, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum, train)
            return o if (drop is None and slope == 1.0) else ops.lrelu_drop(o, drop, slope)
        return ops.bn_act(c, drop, bn.weight, bn.bias, gmod, bmod, bn.running_mean, bn.running_var, slope, bn.eps,
                          momentum, train)

    def forward(self, x, y=None, z=None):
        return self.apply_with(x, z, None, 1.0, y=y)




class SelfAttention(nn.Module):
    """out = gamma * final(value P) + x with P = softmax over the key positions of key^T query, key and value
    max-pooled by 4 along time; query / key / value / final are linear 1x1 convs without bias and gamma is a scalar
    parameter that starts at zero (reference layers.py:41-64).  Returns (out, P).  Twice differentiable (the critic's
    gradient penalty passes through it).  float32 tensors only."""

    def __init__(self, channels_in, spectral, init="xavier_uniform"):
        super().__init__()
        d_key = max(channels_in // 8, 2)
        conf = dict(kernel_size=1, equalized=False, spectral=spectral, init=init, bias=False, act_alpha=-1)
        self.gamma = nn.Parameter(torch.tensor(0.0), requires_grad=True)
        self.pooling = nn.MaxPool1d(4)  # (parameter-less; kept for module-tree parity)
        self.key_conv = GeneralConv(channels_in, d_key, **conf)
        self.query_conv = GeneralConv(channels_in, d_key, **conf)
        self.value_conv = GeneralConv(channels_in, channels_in // 2, **conf)
        self.final_conv = GeneralConv(channels_in // 2, channels_in, **conf)
        self.softmax = nn.Softmax(dim=-1)
        self.scale = 1.0

    def forward(self, x):
        pool = self.pooling.kernel_size
        query = self.query_conv(x)
        key = ops.maxpool(self.key_conv(x), pool)
        value = ops.maxpool(self.value_conv(x), pool)
        attention = ops.softmax_dim1(ops.bmm(key, query, trans_a=True, alpha=1.0 / self.scale))
        out = self.final_conv(ops.bmm(value, attention))
        return ops.axpby(ops.mul_param(out, self.gamma), 1.0, x, 1.0), attention


class PassChannelResidual(nn.Module):
    """x + y[:, :C] where C is x's channel count (reference layers.py:216-223, the branch that works for [B, C, T]
    tensors: y has at least as many channels as x; a y of one time step is repeated over time, as broadcasting
    does in the reference)."""

    def forward(self, x, y):
        if x.size(1) > y.size(1):
            raise NotImplementedError("PassChannelResidual: a skip with fewer channels than the block's output "
                                      "(the reference indexes a fourth dimension there and fails)")
        return ops.channel_window(y, x, 0, 0, x.size(1))


class EqualizedConv1d(nn.Module):
    """Conv1d whose INPUT is multiplied by the RMS of the freshly initialised weight; the stored
    weight is that init divided by its RMS (reference layers.py:150-174).  ``self.conv`` is an
    ``nn.Conv1d`` used as the parameter container, so state-dict keys and the RNG stream consumed
    at construction match the reference; its own forward is never called."""

    def __init__(self, in_channels, out_channels, kernel_size, padding, spectral, equalized, init, act_alpha, bias,
                 groups, stride):
        super().__init__()
        if stride != 1 or (groups != 1 and not (groups == in_channels == out_channels)):
            raise NotImplementedError("EqualizedConv1d: strides and groups other than 1 or the channel count (depthwise) "
                                      "are not supported")
        self.depthwise = groups != 1
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias,
                              groups=groups)
        w = self.conv.weight
        with torch.no_grad():
            if bias:
                self.conv.bias.zero_()
            slope = act_alpha if act_alpha > 0 else 1
            if init == "kaiming_normal":
                nn.init.kaiming_normal_(w, a=slope)
            elif init == "xavier_uniform":
                nn.init.xavier_uniform_(w, gain=calculate_gain("leaky_relu", param=slope))
            elif init == "orthogonal":
                nn.init.orthogonal_(w, gain=calculate_gain("leaky_relu", param=slope))
            self.scale = (torch.mean(w ** 2) ** 0.5).item() if equalized else 1.0
            w.copy_(w / self.scale)
        self.kernel_size = kernel_size
        self.padding = padding
        self.spectral = bool(spectral)
        if spectral:
            # (layers.py:170-171) torch's own registration: the parameter becomes ``weight_orig`` and the vectors of
            # the power iteration the buffers ``weight_u`` / ``weight_v``, drawn from the same random stream as the
            # reference's.  Its forward hook never runs (the container's forward is never called): ``weight()`` does
            # the iteration and the normalisation with the library's kernels.
            self.conv = nn.utils.spectral_norm(self.conv)
            self._flip = None  # the normalised weight is a new tensor at every forward: its flipped view is not cached
        else:
            self._flip = ops.FlipCache(self.conv.weight)

    def weight(self):
        """The weight the conv runs with: the parameter, or with spectral norm the parameter over its largest singular
        value as one power iteration per forward in training mode estimates it."""
        if not self.spectral:
            return self.conv.weight
        c = self.conv
        w, self.last_sigma = ops.spectral_norm(c.weight_orig, c.weight_u, c.weight_v, 1e-12, self.training)
        return w

    def kind(self, x):
        k, pad = self.kernel_size, self.padding
        if k in (1, 3) and pad == (k - 1) // 2:
            return "same"
        if pad == k - 1 and x.shape[2] == 1:
            return "head"
        if pad == 0 and x.shape[2] == k:
            return "tail"
        raise NotImplementedError(f"conv with kernel_size={k}, padding={pad} on length {x.shape[2]} is not supported")

    def forward(self, x):
        if self.depthwise:
            return ops.dwconv1d(x, self.weight(), self.conv.bias, self.scale, self.padding)
        if self.kind(x) != "same":
            raise NotImplementedError("a bare full-length conv is only available fused inside GeneralConv")
        return ops.conv1d(x, self.weight(), self.conv.bias, None, self.scale, ACT_NONE, flip=self._flip)


class EqualizedSeparableConv1d(nn.Module):
    """A depthwise conv (its own bias, no activation) followed by a 1x1 conv (reference layers.py:137-147).  Inside
    GeneralConv the 1x1 half carries the layer's fused epilogue."""

    def __init__(self, in_channels, out_channels, kernel_size, padding, spectral, equalized, init, act_alpha, bias,
                 groups, stride):
        super().__init__()
        self.net = nn.Sequential(
            EqualizedConv1d(in_channels, in_channels, kernel_size, padding, spectral, equalized, init, -1, True,
                            groups=in_channels, stride=stride),
            EqualizedConv1d(in_channels, out_channels, 1, 0, spectral, equalized, init, act_alpha, bias, groups, 1))

    def forward(self, x):
        return self.net(x)


class ConcatResidual(nn.Module):
    """h + cat([x, net(x)], dim=1) with ``net`` a linear 1x1 conv to the missing channels (reference
    layers.py:226-239); h + x when the channel counts agree."""

    def __init__(self, ch_in, ch_out, equalized, spectral, init):
        super().__init__()
        assert ch_out >= ch_in
        self.net = GeneralConv(ch_in, ch_out - ch_in, kernel_size=1, equalized=equalized, act_alpha=-1,
                               spectral=spectral, init=init) if ch_out > ch_in else None

    def forward(self, h, x):
        if self.net is None:
            return ops.axpby(h, 1.0, x, 1.0)
        out = ops.channel_window(x, h, 0, 0, x.size(1))
        extra = self.net(x)
        return ops.channel_window(extra, out, 0, x.size(1), extra.size(1))


class GeneralConv(nn.Module):
    """conv -> (LeakyReLU) -> (PixelNorm) -> (GDrop), executed as one or two fused kernels.

    The discriminator form (no norm) is twice differentiable; the generator form (pixel norm or
    the to_rgb tanh) supports a first-order backward, which is all the training step asks of it."""

    def __init__(self, in_channels, out_channels, z_to_bn_size=0, kernel_size=3, equalized=True, pad=None,
                 act_alpha=0.2, do=0, num_classes=0, act_norm=None, spectral=False, init="kaiming_normal", bias=True,
                 separable=False, stride=1):
        super().__init__()
        if act_norm not in (None, "pixel", "batch"):
            raise NotImplementedError("GeneralConv: act_norm other than None / 'pixel' / 'batch' is not supported")
        pad = (kernel_size - 1) // 2 if pad is None else pad
        conv_class = EqualizedSeparableConv1d if separable else EqualizedConv1d
        self.conv = conv_class(in_channels, out_channels, kernel_size, padding=pad, spectral=spectral,
                               equalized=equalized, init=init, act_alpha=act_alpha,
                               bias=bias if act_norm != "batch" else False, groups=1, stride=stride)
        self.separable = bool(separable)
        # (layers.py:190-192: constructed right after the conv, i.e. its modulation network draws its weights next)
        self.norm = ConditionalBatchNorm(out_channels, num_classes, z_to_bn_size, spectral) if act_norm == "batch" else None
        self.act_alpha = act_alpha
        self.act_norm = act_norm
        net = []
        if act_alpha >= 0:
            net.append(nn.ReLU() if act_alpha == 0 else nn.LeakyReLU(act_alpha))
        if act_norm == "pixel":
            net.append(PixelNorm())
        self.drop = GDropLayer(strength=do) if do != 0 else None
        if self.drop is not None:
            net.append(self.drop)
        self.net = nn.Sequential(*net)  # parameter-less; kept for module-tree parity

    # the critic's layers are differentiated twice (gradient penalty): the Discriminator sets this, and a pixel norm
    # then runs as a chain of multilinear pieces instead of inside the first-order fused layer
    second_order = False

    def _multipliers(self, x):
        if self.drop is None:
            return None
        pointwise = self.conv.net[1] if self.separable else self.conv
        return self.drop.sample(x.shape[0], pointwise.conv.out_channels, x.device)

    def _with_conv_noise(self, x, c, y, z, conv_noise):
        """conv -> * noise[b, ch] -> (norm) -> activation -> (PixelNorm) -> (GDrop) piece by piece (reference
        layers.py:207-213 with ``conv_noise``): the multiplier sits between the conv and the activation, and its
        gradient needs the conv's own output, so this form does not run as the fused layer.  First-order
        differentiable (the generator's layers)."""
        w, b, s = c.weight(), c.conv.bias, c.scale
        kind = c.kind(x)
        if kind == "tail" or self.second_order:
            raise NotImplementedError("per-channel conv noise in the critic's layers")
        xin = x.squeeze(2) if kind == "head" else x
        lin = ops.g_layer(xin, w, b, None, s, 0.0, mode="linear", kind=kind, flip=c._flip)
        h = ops.row_scale(lin, conv_noise)
        slope = float(max(self.act_alpha, 0.0))
        mult = self._multipliers(x)
        if self.act_norm == "batch":
            return self.norm.apply_with(h, z, mult, slope if self.act_alpha >= 0 else 1.0, y=y)
        if self.act_alpha >= 0:
            h = ops.lrelu_drop(h, None, slope)
        if self.act_norm == "pixel":
            h = ops.pixel_norm(h)
        return h if mult is None else ops.lrelu_drop(h, mult, 1.0)

    def forward(self, x, y=None, z=None, conv_noise=None, _tanh_scale=None):
        c = self.conv
        if conv_noise is not None:
            if self.separable or _tanh_scale is not None:
                raise NotImplementedError("per-channel conv noise on a separable or to_rgb layer")
            return self._with_conv_noise(x, c, y, z, conv_noise)
        if self.separable:
            # the depthwise half (which also does the full-length layers' change of length), then the 1x1 half as an
            # ordinary same-padded layer with this layer's epilogue
            x = c.net[0](x)
            c = c.net[1]
        w, b, s = c.weight(), c.conv.bias, c.scale
        kind = c.kind(x)
        slope = float(max(self.act_alpha, 0.0))
        if _tanh_scale is not None:  # to_rgb: linear conv + tanh(scale * .)
            if self.act_alpha >= 0 or self.act_norm is not None or self.drop is not None or kind != "same":
                raise NotImplementedError("fused tanh epilogue needs a plain linear conv")
            return ops.g_layer(x, w, b, None, s, 0.0, mode="tanh", kind="same", tanh_scale=_tanh_scale, flip=c._flip)
        if self.act_norm == "batch":
            # conv without bias -> batch norm (-> modulation by the latent) -> (leaky) ReLU -> GDrop (layers.py:207-213)
            if self.norm.second_order:  # the critic: every piece twice differentiable
                if kind == "same":
                    c_lin = ops.conv1d(x, w, b, None, s, ACT_NONE, flip=c._flip)
                elif kind == "tail":  # (the full-length layer with slope 1: linear)
                    c_lin = ops.tail_conv(x, w, b, None, s, 1.0)
                else:
                    raise NotImplementedError("batch norm after a length-1 (head) conv in the critic")
            else:
                if kind == "tail":
                    raise NotImplementedError("batch norm after the full-length layer outside the critic")
                xin = x.squeeze(2) if kind == "head" else x
                c_lin = ops.g_layer(xin, w, b, None, s, 0.0, mode="linear", kind=kind, flip=c._flip)
            return self.norm.apply_with(c_lin, z, self._multipliers(x), slope if self.act_alpha >= 0 else 1.0, y=y)
        if self.act_norm == "pixel" and self.second_order:
            # the critic (everything twice differentiable): conv + (leaky) ReLU, the norm from multilinear pieces,
            # then the multipliers
            if kind == "same":
                a = ops.conv1d(x, w, b, None, s, ACT_LRELU if self.act_alpha >= 0 else ACT_NONE, slope, flip=c._flip)
            elif kind == "tail":
                a = ops.tail_conv(x, w, b, None, s, slope if self.act_alpha >= 0 else 1.0)
            else:
                raise NotImplementedError("pixel norm after a length-1 (head) conv in the critic")
            y_ = ops.pixel_norm_twice(a)
            drop = self._multipliers(x)
            return y_ if drop is None else ops.lrelu_drop(y_, drop, 1.0)
        if self.act_norm == "pixel":
            if self.act_alpha < 0 or kind == "tail":
                raise NotImplementedError("pixel norm is fused with a (leaky-)ReLU same/head conv only")
            xin = x.squeeze(2) if kind == "head" else x
            return ops.g_layer(xin, w, b, self._multipliers(x), s, slope, mode="pn", kind=kind, flip=c._flip)
        if self.act_alpha < 0:
            if self.drop is not None or kind != "same":
                raise NotImplementedError("linear conv with drop / full-length kernel is not supported")
            return ops.conv1d(x, w, b, None, s, ACT_NONE, flip=c._flip)
        if kind == "same":
            forced = None if _forced_masks is None else _forced_masks.next_for((x.shape[0], w.shape[0], x.shape[2]))
            return ops.conv1d(x, w, b, self._multipliers(x), s, ACT_LRELU, slope, flip=c._flip, forced_mask=forced)
        if kind == "tail":
            forced = None if _forced_masks is None else _forced_masks.next_for((x.shape[0], w.shape[0], 1))
            return ops.tail_conv(x, w, b, self._multipliers(x), s, slope, forced_mask=forced)
        raise NotImplementedError("the length-1 (head) conv is only supported with pixel norm")
