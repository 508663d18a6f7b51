"""Evaluation metrics of the reference's plugins on the device (reference plugins.py:361-451: the 1-D sliced
Wasserstein distance).  Same class name, constructor and methods as the reference's plugin; the patches,
their normalisation, the projections, the sort and the distance run as HIP kernels, and the generated and
real batches never leave the GPU (the reference moves them to the host, plugins.py:408).

Random draws follow the reference exactly: patch positions come one by one from numpy's global generator
(plugins.py:442) and the directions from torch's CPU generator (plugins.py:386), in the reference's order, so
equal seeds give equal draws."""
from __future__ import annotations

import gc

import numpy as np
import torch

from . import _lib
from ._lib import call
from .plugins import Plugin
from .utils import cudize

EPSILON = 1e-8


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a HIP device tensor, got {t.device}; this package has no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


class SlicedWDistance(Plugin):
    MAX_SORT_KEYS = 1 << 30  # keys per sort (8 bytes each)

    def __init__(self, progression_scale: int, output_snapshot_ticks: int, patches_per_item: int = 16,
                 patch_size: int = 49, max_items: int = 1024, number_of_projections: int = 512,
                 dir_repeats: int = 4, dirs_per_repeat: int = 128):
        super().__init__([(1, "epoch")])
        # when: every ``output_snapshot_ticks`` epochs, on at most ``max_items`` signals per set
        self.output_snapshot_ticks, self.max_items = output_snapshot_ticks, max_items
        # patches: ``patches_per_item`` windows of ``patch_size`` samples, one set per pyramid level
        self.progression_scale, self.patches_per_item, self.patch_size = progression_scale, patches_per_item, patch_size
        # directions (``number_of_projections`` is carried, and unused, as in the reference plugins.py:371)
        self.number_of_projections, self.dir_repeats, self.dirs_per_repeat = (number_of_projections, dir_repeats,
                                                                              dirs_per_repeat)

    def register(self, trainer):
        trainer.stats["swd"] = dict(log_name="swd", log_epoch_fields=["{val:.2f}", "{epoch:.2f}"],
                                    val=float("nan"), epoch=0)
        self.trainer = trainer

    # ---- device pieces ---------------------------------------------------------------------------------
    def _sorted_projections(self, desc: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
        """64-bit keys (direction << 32 | ordered float), every direction's projections sorted and contiguous."""
        rows, c = desc.shape
        D = dirs.shape[1]
        n = rows * D
        keys = torch.empty(n, device=desc.device, dtype=torch.int64)
        out = torch.empty_like(keys)
        call("eeggan_swd_project_keys", desc.data_ptr(), dirs.data_ptr(), keys.data_ptr(), rows, c, D, _stream())
        nbytes = _lib.load().eeggan_swd_sort_workspace(n, D)
        ws = torch.empty(nbytes, device=desc.device, dtype=torch.uint8)
        call("eeggan_swd_sort", keys.data_ptr(), out.data_ptr(), n, D, ws.data_ptr(), nbytes, _stream())
        return out

    def sliced_wasserstein(self, A, B):
        """Mean over ``dir_repeats`` sets of ``dirs_per_repeat`` random unit directions of the mean absolute
        difference between the sorted projections of A and B ([rows, components], equal shapes)."""
        A, B = _dev_f32(A, "sliced_wasserstein A"), _dev_f32(B, "sliced_wasserstein B")
        if A.shape != B.shape or A.dim() != 2:
            raise RuntimeError(f"sliced_wasserstein: shapes {tuple(A.shape)} and {tuple(B.shape)} must agree")
        lib = _lib.load()
        nbytes = lib.eeggan_swd_distance_workspace()
        ws = torch.empty(nbytes, device=A.device, dtype=torch.uint8)
        # directions are taken in chunks so that one sort holds at most MAX_SORT_KEYS keys (four key buffers and
        # the sort's workspace are alive at once); a chunk's mean enters with the share of the directions it holds
        chunk = max(1, min(self.dirs_per_repeat, self.MAX_SORT_KEYS // max(1, A.shape[0])))
        spans = [(lo, min(lo + chunk, self.dirs_per_repeat)) for lo in range(0, self.dirs_per_repeat, chunk)]
        results = torch.empty((self.dir_repeats, len(spans)), device=A.device, dtype=torch.float32)
        for repeat in range(self.dir_repeats):
            dirs = torch.randn(A.shape[1], self.dirs_per_repeat)  # host generator, like the reference
            dirs /= torch.sqrt((dirs * dirs).sum(dim=0, keepdim=True) + EPSILON)
            for j, (lo, hi) in enumerate(spans):
                part = dirs[:, lo:hi].contiguous().to(A.device)
                ka, kb = self._sorted_projections(A, part), self._sorted_projections(B, part)
                call("eeggan_swd_distance", ka.data_ptr(), kb.data_ptr(), ka.numel(), results[repeat, j:].data_ptr(),
                     ws.data_ptr(), nbytes, _stream())
                del ka, kb
        weights = torch.tensor([(hi - lo) / self.dirs_per_repeat for lo, hi in spans], dtype=torch.float64)
        return float((results.double().cpu() * weights).sum(dim=1).mean().item())

    def _descriptors(self, batch: torch.Tensor, items: int, step: int = 1) -> torch.Tensor:
        """Normalised patch descriptors of ``batch[:, :, ::step]``: [items * patches_per_item * patch_size, c]."""
        _, c, t_full = batch.shape
        t_level = -(-t_full // step)  # length of batch[:, :, ::step]
        max_index = t_level - self.patch_size
        n = items * self.patches_per_item
        starts = np.empty(n, dtype=np.int32)
        for i in range(n):  # one draw per patch, in the reference's order (plugins.py:440-442)
            starts[i] = np.random.randint(0, max_index)
        starts_d = torch.from_numpy(starts).to(batch.device)
        desc = torch.empty((n * self.patch_size, c), device=batch.device, dtype=torch.float32)
        call("eeggan_swd_patches", batch.data_ptr(), starts_d.data_ptr(), desc.data_ptr(), items, c, t_full, step,
             self.patches_per_item, self.patch_size, _stream())
        nbytes = _lib.load().eeggan_swd_normalize_workspace()
        ws = torch.empty(nbytes, device=batch.device, dtype=torch.uint8)
        call("eeggan_swd_normalize", desc.data_ptr(), desc.shape[0], c, EPSILON, ws.data_ptr(), nbytes, _stream())
        return desc

    def get_descriptors(self, batch1, batch2):
        """One distance per "pyramid level" (as many levels as the signal can be divided by
        ``progression_scale`` while staying at least a patch long).  Like the reference as written, every level
        takes fresh random patches from the FULL-RATE batches: it rebuilds its batch list from the originals at
        the top of each level (plugins.py:436) and decimates only after the patches are taken (plugins.py:449),
        so the decimated signals are never read."""
        batch1, batch2 = _dev_f32(batch1, "get_descriptors batch1"), _dev_f32(batch2, "get_descriptors batch2")
        b, c, t_max = batch1.shape
        if batch2.shape[0] < b or batch2.shape[1] != c:
            raise RuntimeError("get_descriptors: the second batch needs at least as many items and equal channels")
        if self.progression_scale < 2:
            raise ValueError("progression_scale must be at least 2")
        num_levels, t = 0, t_max
        while t >= self.patch_size:
            num_levels += 1
            t //= self.progression_scale
        swd = []
        for _ in range(num_levels):
            d1 = self._descriptors(batch1, b)
            d2 = self._descriptors(batch2, b)
            swd.append(self.sliced_wasserstein(d1, d2))
        return swd

    def _gather(self, next_batch) -> torch.Tensor:
        """Batches from ``next_batch()`` until ``max_items`` signals are there; one batch is enough when the
        signals are shorter than a patch (nothing can be measured on them: reference plugins.py:409-410)."""
        batches, have = [], 0
        while have < self.max_items:
            batches.append(next_batch())
            if batches[-1].size(2) < self.patch_size:
                break
            have += batches[-1].size(0)
        return torch.cat(batches, dim=0)

    def epoch(self, epoch_index):
        """Every ``output_snapshot_ticks`` epochs: generated signals against reals, one distance per level and
        their mean last, into ``stats['swd']`` (reference plugins.py:397-424; both sets stay on the device)."""
        if epoch_index % self.output_snapshot_ticks:
            return
        gc.collect()
        tr = self.trainer
        with torch.no_grad():
            fakes = self._gather(lambda: tr.generator(cudize(next(tr.random_latents_generator)))[0]["x"].detach())
            reals = self._gather(lambda: cudize(next(tr.dataiter))["x"])
        per_level = self.get_descriptors(fakes, reals)
        if per_level:
            per_level.append(np.array(per_level).mean())
        tr.stats["swd"].update(val=per_level, epoch=epoch_index)


class FidCalculator(Plugin):
    """Frechet distance between CPC features of generated and of real signals (reference plugins.py:454-648).

    The statistics (``torch_cov``, ``sqrt_newton_schulz``, ``torch_calculate_frechet_distance``, ``calc_fid``)
    keep the reference's names and arithmetic and run on the device.  ``epoch`` follows the reference's flow
    with the calls its feature network actually offers (the reference plugin at HEAD still calls a constructor
    signature and an ``inference_forward`` method its network no longer has, plugins.py:469-477, 589)."""

    def __init__(self, num_channels, create_dataloader_fun, target_seq_len, num_samples=1024 * 16,
                 output_snapshot_ticks=25, calc_for_z=True, calc_for_zp=True, calc_for_c=True, calc_for_cp=True,
                 network=None):
        super().__init__([(1, "epoch")])
        from .cpc import Network
        self.create_dataloader_fun = create_dataloader_fun
        self.output_snapshot_ticks = output_snapshot_ticks
        self.last_depth = -1
        self.last_alpha = -1
        self.target_seq_len = target_seq_len
        self.calc_z, self.calc_zp, self.calc_c, self.calc_cp = calc_for_z, calc_for_zp, calc_for_c, calc_for_cp
        self.num_samples = num_samples
        self.network = cudize(network if network is not None else Network(num_channels)).eval()
        self.real_stats = {}

    def register(self, trainer):
        self.trainer = trainer
        fields, dd = [], {"epoch": 0}
        for flag, key in ((self.calc_c, "c_fake"), (self.calc_z, "z_fake"), (self.calc_cp, "cp_fake"),
                          (self.calc_zp, "zp_fake")):
            if flag:
                fields.append("{" + key + ":.2f}")
                dd[key] = float("nan")
        fields.append("{epoch:.2f}")
        self.trainer.stats["FID"] = {"log_name": "FID", "log_epoch_fields": fields, **dd}

    # ---- statistics ------------------------------------------------------------------------------------
    @staticmethod
    def mean_and_cov(m):
        """(column means, unbiased covariance of the columns) of a feature matrix [N, F], one pass over it."""
        m = _dev_f32(m, "mean_and_cov m")
        N, F = m.shape
        lib = _lib.load()
        nbytes = lib.eeggan_feature_stats_workspace(N, F)
        ws = torch.empty(nbytes, device=m.device, dtype=torch.uint8)
        mean = torch.empty(F, device=m.device, dtype=torch.float32)
        cov = torch.empty((F, F), device=m.device, dtype=torch.float32)
        call("eeggan_feature_stats", m.data_ptr(), N, F, mean.data_ptr(), cov.data_ptr(), ws.data_ptr(), nbytes, _stream())
        return mean, cov

    @staticmethod
    def torch_cov(m, rowvar=False):
        if m.dim() > 2:
            raise ValueError("m has more than 2 dimensions")
        if m.dim() < 2:
            m = m.view(1, -1)
        if rowvar and m.size(0) != 1:
            m = m.t()
        return FidCalculator.mean_and_cov(m)[1]

    @staticmethod
    def sqrt_newton_schulz(A, numIters, dtype=None):
        """Matrix square root of A [batch, n, n] by the coupled Newton-Schulz iteration (plugins.py:540-555)."""
        from .cpc import dense
        A = _dev_f32(A, "sqrt_newton_schulz A")
        if A.dim() != 3 or A.shape[1] != A.shape[2]:
            raise RuntimeError("sqrt_newton_schulz: expected [batch, n, n]")
        out = torch.empty_like(A)
        for b in range(A.shape[0]):
            a = A[b]
            norm = a.mul(a).sum().sqrt()  # (one scalar of glue on an n x n matrix)
            n = a.shape[0]
            Y = (a / norm).contiguous()
            Z = torch.eye(n, device=a.device, dtype=torch.float32)
            for _ in range(numIters):
                T = dense(Z, Y, w_is_kn=True, alpha=-0.5, eye=1.5)  # 0.5 (3 I - Z Y)
                Y = dense(Y, T, w_is_kn=True)
                Z = dense(T, Z, w_is_kn=True)
            out[b] = Y * torch.sqrt(norm)
        return out

    @staticmethod
    def torch_calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
        from .cpc import dense
        assert mu1.shape == mu2.shape, "Training and test mean vectors have different lengths"
        assert sigma1.shape == sigma2.shape, "Training and test covariances have different dimensions"
        diff = mu1 - mu2
        prod = dense(_dev_f32(sigma1, "sigma1"), _dev_f32(sigma2, "sigma2"), w_is_kn=True)
        covmean = FidCalculator.sqrt_newton_schulz(prod.unsqueeze(0), 50).squeeze(0)
        return diff.dot(diff) + torch.trace(sigma1) + torch.trace(sigma2) - 2 * torch.trace(covmean)

    @staticmethod
    def eig_calculate_frechet_distance(mu1, sigma1, mu2, sigma2):
        """The distance with tr sqrt(sigma1 sigma2) taken from eigenvalues, in fp64 on the device: the eigenvalues
        of sigma1 sigma2 are those of the symmetric r sigma2 r with r = sqrt(sigma1), so the trace of the principal
        square root is the sum of their roots.  It stands where the reference leaves the Newton-Schulz iteration
        for a host ``scipy.linalg.sqrtm`` of the product (plugins.py:517-537, 560-562) and gives that result
        without the detour over the host; it does not break down on singular covariances."""
        s1, s2 = sigma1.double(), sigma2.double()
        w, v = torch.linalg.eigh(0.5 * (s1 + s1.t()))
        r = (v * w.clamp_min(0).sqrt()) @ v.t()
        m = r @ s2 @ r
        lam = torch.linalg.eigvalsh(0.5 * (m + m.t())).clamp_min(0)
        diff = (mu1 - mu2).double()
        return diff.dot(diff) + torch.trace(s1) + torch.trace(s2) - 2 * lam.sqrt().sum()

    @staticmethod
    def calc_fid(mu1, std1, mu2, std2):
        """Newton-Schulz first; when it does not converge (NaN: an ill-conditioned product in fp32), the
        eigenvalue form, as the reference then switches to its second method (plugins.py:558-563)."""
        fid = FidCalculator.torch_calculate_frechet_distance(mu1, std1, mu2, std2).item()
        if fid != fid:
            fid = FidCalculator.eig_calculate_frechet_distance(mu1, std1, mu2, std2).item()
        return fid

    # ---- features --------------------------------------------------------------------------------------
    def features(self, x):
        """The four feature matrices of a batch, shaped as the reference shapes them before its statistics
        (plugins.py:590-597): z and c as rows of 128 numbers read straight from [B, 128, T] memory."""
        from .utils import resample_signal
        x = resample_signal(x, x.size(2), self.target_seq_len, True)
        lat = self.network(x, no_loss=True).latents
        return {"z": lat.z.reshape(-1, lat.z.size(1)), "c": lat.c.contiguous().view(-1, lat.c.size(1)),
                "zp": lat.z_p, "cp": lat.c_p}

    def _stats_of(self, batches):
        """Mean and covariance per feature kind over a stream of input batches (at least num_samples samples)."""
        kinds = [k for k, on in (("z", self.calc_z), ("c", self.calc_c), ("zp", self.calc_zp), ("cp", self.calc_cp)) if on]
        parts = {k: [] for k in kinds}
        seen = 0
        with torch.no_grad():
            for x in batches:
                feats = self.features(x)
                for k in kinds:
                    parts[k].append(feats[k])
                seen += x.size(0)
                if seen >= self.num_samples:
                    break
        return {k: self.mean_and_cov(torch.cat(v, dim=0)) for k, v in parts.items()}

    def epoch(self, epoch_index):
        if epoch_index % self.output_snapshot_ticks != 0:
            return
        tr = self.trainer
        if not (self.last_depth == tr.dataset.model_depth and self.last_alpha == tr.dataset.alpha):
            loader = self.create_dataloader_fun(min(tr.stats["minibatch_size"], 1024), False, tr.dataset.model_depth,
                                                tr.dataset.alpha)
            self.real_stats = self._stats_of(cudize(data["x"]) for data in loader)
            self.last_depth, self.last_alpha = tr.dataset.model_depth, tr.dataset.alpha

        def fakes():
            while True:
                yield tr.generator(cudize(next(tr.random_latents_generator)))[0]["x"]

        fake_stats = self._stats_of(fakes())
        for k, (mu, cov) in fake_stats.items():
            tr.stats["FID"][k + "_fake"] = self.calc_fid(mu, cov, *self.real_stats[k])
        tr.stats["FID"]["epoch"] = epoch_index
