"""The CPC feature network's inference pass on the device (reference cpc/cpc_network.py): the encoder, the
p-norm poolers and the GRU contextualizer that ``Network.forward(x, no_loss=True)`` evaluates to produce the
latents the Frechet distance is computed on (reference plugins.py:574-648).

Same class names, constructor defaults, module tree and ``state_dict`` keys as the reference for these parts,
built in the reference's order so that equal seeds give equal initial weights; a reference checkpoint loads with
``strict=False`` (the loss heads, cpc_network.py:297-306, are training-only and are not built).  Only the
default architecture is accelerated (conv encoder 32/64/128 with strides 5/4/3, one-layer unidirectional GRU);
other options raise ``NotImplementedError`` at construction.  Inference only: the modules must be in eval mode
(BatchNorm uses its running statistics, dropout is off)."""
from __future__ import annotations

from collections import namedtuple

import torch
from torch import nn

from ._lib import call

NetworkLosses = namedtuple("NetworkLosses", ["global_", "local_", "prediction_"])
NetworkAccuracies = namedtuple("NetworkAccuracies", ["global_", "local_", "prediction_"])
NetworkLatents = namedtuple("NetworkLatents", ["z", "c", "z_p", "c_p"])
NetworkOutput = namedtuple("NetworkOutput", ["losses", "accuracies", "latents"])


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a HIP device tensor, got {t.device}; this package has no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


def dense(x, w, bias=None, scale=None, shift=None, *, relu=False, w_is_kn=False, alpha=1.0, eye=0.0):
    """shift + scale * act(bias + alpha * x w^T) + eye * I for a 2-D ``x``; ``w`` is [M, K] (``nn.Linear``) or
    [K, M] with ``w_is_kn`` (a plain matrix product)."""
    x, w = _chk(x, "dense x"), _chk(w, "dense w")
    if x.dim() != 2 or w.dim() != 2:
        raise RuntimeError("dense: x and w must be matrices")
    N, K = x.shape
    M = w.shape[1] if w_is_kn else w.shape[0]
    if (w.shape[0] if w_is_kn else w.shape[1]) != K:
        raise RuntimeError(f"dense: x {tuple(x.shape)} and w {tuple(w.shape)} do not match")
    for name, t in (("bias", bias), ("scale", scale), ("shift", shift)):
        if t is not None and (t.numel() != M or not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous()):
            raise RuntimeError(f"dense: {name} must be a contiguous float32 device vector of {M} elements")
    y = torch.empty((N, M), device=x.device, dtype=torch.float32)
    call("eeggan_dense", x.data_ptr(), w.data_ptr(), _p(bias), _p(scale), _p(shift), y.data_ptr(), N, K, M, K, M,
         int(w_is_kn), float(alpha), float(eye), int(relu), _stream())
    return y


def _bn_affine(bn: nn.BatchNorm1d):
    """Eval-mode BatchNorm1d as y = scale * x + shift."""
    with torch.no_grad():
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale
    return scale.contiguous(), shift.contiguous()


def _inference_only(module: nn.Module):
    if module.training:
        raise NotImplementedError(f"{type(module).__name__}: inference only -- call .eval() first")


class ConvEncoder(nn.Sequential):
    def __init__(self, input_channels=5, dropout=0.1, down_ratios=None, channel_sizes=None, bn=True, sinc_encoder=False):
        if sinc_encoder:
            raise NotImplementedError("ConvEncoder: the sinc front end is not supported")
        if down_ratios is None:
            down_ratios = [5, 4, 3]
        if channel_sizes is None:
            channel_sizes = [32, 64, 128]
        if list(down_ratios) != [5, 4, 3] or list(channel_sizes) != [32, 64, 128] or not bn:
            raise NotImplementedError("ConvEncoder: only the default 32/64/128 encoder with strides 5/4/3 and batch "
                                      "norm is accelerated")
        kernel_sizes = [2 * dr - 1 for dr in down_ratios]
        act = nn.ReLU()
        net = []
        last_layer = len(down_ratios) - 1
        self.z_size = channel_sizes[-1]
        for i, (dr, cs, ks) in enumerate(zip(down_ratios, channel_sizes, kernel_sizes)):
            net.append(nn.Conv1d(input_channels, cs, ks, stride=dr, padding=(ks - 1) // 2))
            net.append(act)
            if dropout != 0.0 and i != last_layer:
                net.append(nn.Dropout(dropout))
            if i != last_layer:
                net.append(nn.BatchNorm1d(cs))
            input_channels = cs
        super().__init__(*net)

    def codes(self, x):
        """(z [B, 128, T/60], the same codes time-major [B, T/60, 128])."""
        _inference_only(self)
        x = _chk(x, "ConvEncoder x")
        convs = [m for m in self if isinstance(m, nn.Conv1d)]
        norms = [m for m in self if isinstance(m, nn.BatchNorm1d)]
        B, cin, T = x.shape
        if cin != convs[0].in_channels or T % 60:
            raise RuntimeError(f"ConvEncoder: input {tuple(x.shape)}: {convs[0].in_channels} channels and a length "
                               "that is a multiple of 60 are expected")
        wt = [c.weight.detach().permute(1, 2, 0).contiguous() for c in convs]  # [ci, j, co]
        (s1, t1), (s2, t2) = _bn_affine(norms[0]), _bn_affine(norms[1])
        z = torch.empty((B, 128, T // 60), device=x.device, dtype=torch.float32)
        zt = torch.empty((B, T // 60, 128), device=x.device, dtype=torch.float32)
        call("eeggan_cpc_encoder", x.data_ptr(), wt[0].data_ptr(), convs[0].bias.data_ptr(), s1.data_ptr(), t1.data_ptr(),
             wt[1].data_ptr(), convs[1].bias.data_ptr(), s2.data_ptr(), t2.data_ptr(), wt[2].data_ptr(),
             convs[2].bias.data_ptr(), z.data_ptr(), zt.data_ptr(), B, cin, T, _stream())
        return z, zt

    def forward(self, x):
        return self.codes(x)[0]


class PNormPooling(nn.Module):
    def __init__(self, input_size, batch_norm=True, mlp_sizes=None, p=None, be_mean_pool=False):
        super().__init__()
        if be_mean_pool or (p is not None and list(p) != [1.0, float("+inf")]):
            raise NotImplementedError("PNormPooling: only the default [1, inf]-norm pooling is accelerated")
        self.is_pool = False
        if mlp_sizes is None:
            mlp_sizes = [input_size * 2, input_size]
        if len(mlp_sizes) != 2:
            raise NotImplementedError("PNormPooling: a two-layer MLP is expected")
        self.p = [1.0, float("+inf")]
        sizes = [input_size * 2] + list(mlp_sizes)
        self.pool_size = sizes[-1]
        layers = [nn.Linear(sizes[0], sizes[1]), nn.ReLU()]
        if batch_norm:
            layers.append(nn.BatchNorm1d(sizes[1]))
        layers.append(nn.Linear(sizes[1], sizes[2]))
        self.network = nn.Sequential(*layers)

    def forward(self, x):
        _inference_only(self)
        x = _chk(x, "PNormPooling x")
        B, C, T = x.shape
        norms = torch.empty((B, 2 * C), device=x.device, dtype=torch.float32)
        call("eeggan_pnorm_pool", x.data_ptr(), norms.data_ptr(), B, C, T, _stream())
        lin1, lin2 = self.network[0], self.network[-1]
        scale, shift = _bn_affine(self.network[2]) if len(self.network) == 4 else (None, None)
        h = dense(norms, lin1.weight.detach(), lin1.bias.detach(), scale, shift, relu=True)
        return dense(h, lin2.weight.detach(), lin2.bias.detach())


class AutoRNN(nn.Module):
    def __init__(self, input_size, hidden_size, bidirectional=False, cell_type="GRU", num_layers=1, dropout=0):
        super().__init__()
        if bidirectional or cell_type.lower() != "gru" or num_layers != 1 or input_size != 128 or hidden_size != 128:
            raise NotImplementedError("AutoRNN: only a one-layer unidirectional GRU(128, 128) is accelerated")
        self.forward_rnn = nn.GRU(input_size, hidden_size, num_layers, dropout=dropout)
        self.backward_rnn = None
        self.c_size = hidden_size

    def forward(self, x, x_time_major=None):
        """x [B, 128, T] (or its time-major copy [B, T, 128] when the caller has one) -> c [B, 128, T]."""
        _inference_only(self)
        x = _chk(x, "AutoRNN x")
        B, C, T = x.shape
        xt = x_time_major if x_time_major is not None else x.permute(0, 2, 1).contiguous()
        rnn = self.forward_rnn
        gi = dense(xt.reshape(B * T, C), rnn.weight_ih_l0.detach(), rnn.bias_ih_l0.detach())
        c = torch.empty((B, self.c_size, T), device=x.device, dtype=torch.float32)
        call("eeggan_gru_forward", gi.data_ptr(), _chk(rnn.weight_hh_l0.detach(), "W_hh").data_ptr(),
             rnn.bias_hh_l0.detach().data_ptr(), c.data_ptr(), B, self.c_size, T, _stream())
        return c


class Network(nn.Module):
    def __init__(self, input_channels, bidirectional=False, contextualizer_num_layers=1, contextualizer_dropout=0,
                 use_transformer=False, prediction_k=4, have_global=True, have_local=True, residual_encoder=False,
                 global_mode="bilinear", local_mode="bilinear", sinc_encoder=False):
        super().__init__()
        if residual_encoder or use_transformer or bidirectional or not have_global:
            raise NotImplementedError("Network: the residual encoder, the transformer contextualizer, the bidirectional "
                                      "GRU and mean pooling are outside the accelerated default configuration")
        # same construction order as the reference (cpc_network.py:283-296): equal seeds, equal weights
        self.encoder = ConvEncoder(input_channels, sinc_encoder=sinc_encoder)
        self.z_pooler = PNormPooling(self.encoder.z_size)
        self.contextualizer = AutoRNN(input_size=self.encoder.z_size, hidden_size=self.encoder.z_size,
                                      bidirectional=bidirectional, num_layers=contextualizer_num_layers,
                                      dropout=contextualizer_dropout, cell_type="GRU")
        # (the reference passes `not have_global and not have_local` in the position of `batch_norm`,
        # cpc_network.py:296: with the defaults this pooler has no normalisation layer)
        self.c_pooler = PNormPooling(self.contextualizer.c_size, not have_global and not have_local)

    def forward(self, x, no_loss=False):
        if not no_loss:
            raise NotImplementedError("Network: only the inference pass (no_loss=True) is available; the contrastive "
                                      "loss heads are training-only")
        z, zt = self.encoder.codes(x)
        z_pooled = self.z_pooler(z)
        c = self.contextualizer(z, zt)
        c_pooled = self.c_pooler(c)
        return NetworkOutput(losses=None, accuracies=None, latents=NetworkLatents(z, c, z_pooled, c_pooled))
