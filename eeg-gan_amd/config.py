"""The default configuration of the training step (reference defaults.yml:3-72, dataset.py:28-51,
plugins.py:29-34) and helpers that build the default networks, optimizers and loss closures the
way the reference's train.py:100-119,131-136,167-170 wires them."""
from __future__ import annotations

from functools import partial

import torch

PROGRESSION_SCALE_UP = [4, 2, 2, 3, 4, 5, 3]
PROGRESSION_SCALE_DOWN = [1, 1, 1, 2, 3, 4, 2]
INITIAL_KERNEL_SIZE = 32
NUM_CHANNELS = 5
LATENT_SIZE = 256
RANDOM_SEED = 1373
BASE_LR = 0.001
ADAM = dict(betas=(0.0, 0.99), eps=1e-8, weight_decay=0)
LOSS = dict(loss_type="wgan_gp", iwass_target=1.0, iwass_drift_epsilon=0.001, grad_lambda=10.0)
MINIBATCH = {0: 256, 1: 256, 2: 128, 3: 128, 4: 48, 5: 32, 6: 32, 7: 32}


def seq_len(depth: int) -> int:
    t = INITIAL_KERNEL_SIZE
    for i in range(depth):
        t = t * PROGRESSION_SCALE_UP[i] // PROGRESSION_SCALE_DOWN[i]
    return t


def stage_channels(stage: int, fmap_base: int = 1024, fmap_max: int = 256, fmap_min: int = 64) -> int:
    """Channels of progression stage ``stage`` (reference network.py:279-280, ``nf(stage + 1)``)."""
    return min(max(int(fmap_base / (2.0 ** (stage + 1))), fmap_min), fmap_max)


def forward_flops(depth: int):
    """(F_G, F_D): conv flops (2 x multiply-adds) of one generator and one critic forward pass per sample at
    ``depth`` with alpha = 1, for the default networks: the k = 32 first layer counted with its 32 effective taps
    on the length-1 latent, the critic's 256 + 1 channel layer, the 1x1 signal layers and the scoring layer
    included.  One training step is 8 F_G + 62 F_D (5 critic iterations of F_G + 12 F_D and a generator
    iteration of 3 F_G + 2 F_D; weight gradients the reference forms and never uses are not counted)."""
    k, k0 = 3, INITIAL_KERNEL_SIZE
    c = [stage_channels(s) for s in range(depth + 1)]
    t = [seq_len(s) for s in range(depth + 1)]
    conv = lambda cin, cout, taps, length: 2.0 * cin * cout * taps * length
    f_g = conv(LATENT_SIZE, c[0], k0, 1) + conv(c[0], c[0], k, t[0])
    f_d = conv(c[0] + 1, c[0], k, t[0]) + conv(c[0], c[0], k0, 1) + conv(c[0], 1, 1, 1)
    for s in range(1, depth + 1):
        f_g += conv(c[s - 1], c[s], k, t[s]) + conv(c[s], c[s], k, t[s])
        f_d += conv(c[s], c[s], k, t[s]) + conv(c[s], c[s - 1], k, t[s])
    f_g += conv(c[depth], NUM_CHANNELS, 1, t[depth])
    f_d += conv(NUM_CHANNELS, c[depth], 1, t[depth])
    return f_g, f_d


def step_flops(depth: int, batch: int) -> float:
    f_g, f_d = forward_flops(depth)
    return (8.0 * f_g + 62.0 * f_d) * batch


def shared_model_kwargs():
    return dict(initial_kernel_size=INITIAL_KERNEL_SIZE, num_rgb_channels=NUM_CHANNELS, fmap_base=1024, fmap_max=256,
                fmap_min=64, kernel_size=3, self_attention_layers=[], progression_scale_up=PROGRESSION_SCALE_UP,
                progression_scale_down=PROGRESSION_SCALE_DOWN, residual=False, separable=False, equalized=True,
                init="kaiming_normal", act_alpha=0.2, num_classes=0, deep=False)


GENERATOR_KWARGS = dict(spectral=False, latent_size=LATENT_SIZE, no_tanh=False, per_channel_noise=False,
                        to_rgb_mode="pggan", split_z=False, dropout=0.2, act_norm="pixel", normalize_latents=True,
                        z_to_bn=False, conv_only=False, shared_embedding_size=32, rgb_generation_mode="pggan")
DISCRIMINATOR_KWARGS = dict(spectral=False, dropout=0.2, act_norm=None, group_size=4, temporal_groups_per_window=1,
                            conv_only=False, input_to_all_layers=False)


def build_models(seed: int = RANDOM_SEED, generator_cls=None, discriminator_cls=None, shared=None, generator=None,
                 discriminator=None):
    """Generator then Discriminator, constructed right after seeding torch, on the CPU.  ``shared`` / ``generator`` /
    ``discriminator``: overrides of the default constructor arguments (e.g. ``shared=dict(residual=True)``)."""
    if generator_cls is None:
        from .network import Discriminator, Generator
        generator_cls, discriminator_cls = Generator, Discriminator
    torch.manual_seed(seed)
    common = {**shared_model_kwargs(), **(shared or {})}
    g = generator_cls(**common, z_distribution="normal", **{**GENERATOR_KWARGS, **(generator or {})})
    d = discriminator_cls(**common, **{**DISCRIMINATOR_KWARGS, **(discriminator or {})})
    return g, d


def loss_closures(**overrides):
    """(critic loss, generator loss) bound to the reference's default loss settings (defaults.yml), or to
    ``overrides`` of them (``loss_type``, ``grad_lambda``, ...)."""
    from .losses import discriminator_loss, generator_loss
    settings = {**LOSS, **overrides}
    d_loss = partial(discriminator_loss, **settings)
    g_loss = partial(generator_loss, random_multiply=False, loss_type=settings["loss_type"], feature_matching_lambda=0.0)
    return d_loss, g_loss


def stage_lr(depth: int, base_lr: float = BASE_LR, minibatch_default: int = 256) -> float:
    return minibatch_default * base_lr / MINIBATCH.get(depth, minibatch_default)
