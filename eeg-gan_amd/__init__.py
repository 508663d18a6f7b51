"""MI355X-native EEG-GAN training step: the reference's ``network`` / ``layers`` / ``losses`` /
``trainer`` / ``plugins`` surface on hand-written gfx950 HIP kernels (``libeeggan_hip.so``)."""
from . import _lib  # noqa: F401  (binding only; the shared library is loaded at first use)

__all__ = ["layers", "network", "losses", "trainer", "plugins", "utils", "ops", "optim", "parallel"]
