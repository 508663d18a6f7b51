"""Shared helpers of the test-suite (the oracle is imported only from tests)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu().flatten()
    b = torch.as_tensor(b).detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-300))


def maxabs(a, b):
    a = torch.as_tensor(a).detach().double().cpu().flatten()
    b = torch.as_tensor(b).detach().double().cpu().flatten()
    return float((a - b).abs().max())


def scales_of(module):
    # (key: the layer's path, i.e. the module's name without the ".conv" a plain layer ends in; the two halves of a
    # separable layer, "... .conv.net.0" / "... .conv.net.1", keep their full names)
    return {(name[:-len(".conv")] if name.endswith(".conv") else name): m.scale for name, m in module.named_modules()
            if type(m).__name__ == "EqualizedConv1d"}


def oracle_params(module, dtype=torch.float64):
    params = {k: v.detach().cpu().clone().to(dtype) for k, v in module.state_dict().items()}
    return params, scales_of(module)


def subsample(t, n=512):
    f = torch.as_tensor(t).detach().flatten()
    stride = max(1, f.numel() // n)
    return f[::stride][:n].double().cpu().numpy()


def check_weights(fx, g, d):
    """The fixture stores (sum, sum of squares) of every initial weight of the reference run.
    The seeded CPU normal generator is only reproducible to ~1 ulp per element across CPU models
    (vectorised log/cos differ), so the checksums are compared to 1e-5, which still separates a
    different random stream (the sums then differ in the first digits)."""
    for net, m in (("G", g), ("D", d)):
        for k, v in m.state_dict().items():
            v = v.double()
            got = np.array([float(v.sum()), float((v * v).sum())])
            np.testing.assert_allclose(got, fx[f"wsum_{net}.{k}"], rtol=1e-5, atol=1e-2,
                                       err_msg=f"{net}.{k} initial weights differ from the reference's")


# Cases of tests/golden/blocks.npz (tools/make_golden.py:block_fixture): name, network, module path,
# keyword arguments of the block's forward, input shape.  Inputs and output cotangents are re-drawn from
# the fixture's generator seed in this order (x of a case, then its gy).
BLOCK_CASES = (
    ("g_head", "g", "block0", dict(last=False), (8, 256, 1)),
    ("g_mid_rgb", "g", "blocks.3", dict(last=True), (4, 64, 768)),
    ("g_wide", "g", "blocks.1", dict(last=False), (4, 256, 256)),
    ("d_first", "d", "blocks.1", dict(first=True), (4, 5, 1280)),
    ("d_mid", "d", "blocks.4", dict(first=False), (4, 64, 512)),
    ("d_last", "d", "blocks.7", dict(first=False), (8, 256, 32)),
)


def block_inputs(fx):
    """Yields (case, x, gy) with the tensors the fixture was made from (checked by their checksums to the
    ~1 ulp per element that seeded CPU normals are reproducible to across CPU models)."""
    import numpy as np
    gen = torch.Generator().manual_seed(int(fx["generator_seed"]))
    for case in BLOCK_CASES:
        name, shape = case[0], case[4]
        x = torch.randn(*shape, generator=gen)
        gy = torch.randn(*[int(v) for v in fx[f"{name}_yshape"]], generator=gen)
        for t, key in ((x, f"{name}_x_check"), (gy, f"{name}_gy_check")):
            got = np.array([float(t.double().sum()), float((t.double() ** 2).sum())])
            assert abs(got[1] - fx[key][1]) < 1e-5 * fx[key][1], f"{key}: the generator does not reproduce the input"
        yield case, x, gy
