"""``VQCodec`` -- the coarse quantiser of ``AnnLite(n_cells > 1)``; mirrors annlite/core/codec/vq.py:9-100.

Reference semantics kept: ``fit`` = one k-means over the raw training vectors (sklearn ``KMeans(n_clusters,
max_iter=iter, n_init)``, vq.py:41-50 -- never normalised, whatever the metric), ``encode`` = nearest centroid in
squared L2 (``scipy.cluster.vq.vq``, vq.py:81-90: first minimum wins), ``codebook`` = f32 [n_clusters, D].
Training is statistical parity only (unseeded sklearn in the reference); seedable here via ``self.seed``.

The assignment step runs on the GPU (``annlite_ivf_select_cells`` with one probe); the centroid update is a
scatter-add (plumbing).  No CPU fallback.
"""
from typing import Optional

import numpy as np
import torch

from ... import ops
from ...enums import Metric
from .base import BaseCodec


def _is_np(x) -> bool:
    return isinstance(x, np.ndarray)


class VQCodec(BaseCodec):
    def __init__(self, n_clusters: int, metric: Metric = Metric.EUCLIDEAN, iter: int = 100, n_init: int = 4, *args, **kwargs):
        super().__init__(require_train=True)
        self.n_clusters = int(n_clusters)
        self.metric = metric
        self.iter = int(iter)
        self.n_init = int(n_init)
        self.seed: Optional[int] = None
        self._codebook: Optional[np.ndarray] = None
        self._cb_dev = {}
        self._sums = None  # partial_fit accumulators (device)
        self._counts = None
        self._pf_cb = None

    def __hash__(self):  # vq.py:31-32
        return hash((self.__class__.__name__, self.n_clusters, self.metric))

    def __getstate__(self):
        st = self.__dict__.copy()
        st['_cb_dev'] = {}
        for key in ('_sums', '_counts', '_pf_cb'):
            if st.get(key) is not None:
                st[key] = st[key].cpu().numpy()
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        if not isinstance(self.__dict__.get('_cb_dev'), dict):  # (files written before the per-device cache)
            self._cb_dev = {}
        for key in ('_sums', '_counts', '_pf_cb'):
            if isinstance(getattr(self, key, None), np.ndarray):
                setattr(self, key, None)  # streaming state does not survive a reload (codebook does)

    # ------------------------------------------------------------------ device views
    @property
    def codebook(self) -> np.ndarray:
        self._check_trained()
        return self._codebook

    @property
    def codebook_dev(self) -> torch.Tensor:
        self._check_trained()
        dev = ops.device()
        cb = self._cb_dev.get(dev)
        if cb is None:
            cb = self._cb_dev[dev] = ops.to_dev(np.ascontiguousarray(self._codebook, dtype=np.float32))
        return cb

    def _set_codebook(self, cb: torch.Tensor):
        self._codebook = cb.detach().cpu().numpy().astype(np.float32)
        self._cb_dev = {}
        self._is_trained = True

    # ------------------------------------------------------------------ training
    @staticmethod
    def _assign(x: torch.Tensor, cb: torch.Tensor) -> torch.Tensor:
        return ops.ivf_select_cells(0, x, cb, 1)[:, 0].to(torch.int64)

    def _lloyd_step(self, x: torch.Tensor, cb: torch.Tensor):
        """one assignment + update; returns (new centres, counts, inertia)"""
        a = self._assign(x, cb)
        C, D = cb.shape
        sums = torch.zeros((C, D), dtype=torch.float32, device=x.device).index_add_(0, a, x)
        counts = torch.zeros((C,), dtype=torch.float32, device=x.device).index_add_(
            0, a, torch.ones_like(a, dtype=torch.float32))
        new = torch.where(counts[:, None] > 0, sums / counts.clamp(min=1.0)[:, None], cb)
        inertia = ((x - cb[a]) ** 2).sum()
        return new, counts, inertia

    @staticmethod
    def _kmeanspp(x: torch.Tensor, C: int, gen: torch.Generator) -> torch.Tensor:
        """k-means++ seeding (sklearn's default ``init``): every next centre is drawn with probability
        proportional to the squared distance to the nearest centre chosen so far."""
        N = x.shape[0]
        cb = torch.empty((C, x.shape[1]), dtype=torch.float32, device=x.device)
        first = torch.randint(0, N, (1,), generator=gen, device=x.device)
        cb[0] = x[first[0]]
        d2 = ((x - cb[0]) ** 2).sum(1)
        x2 = (x * x).sum(1)
        trials = 2 + int(np.log(C))  # sklearn's greedy variant: the best of a few draws per step
        for c in range(1, C):
            total = d2.sum()
            if float(total.item()) <= 0.0:  # fewer distinct rows than centres
                cb[c] = x[torch.randint(0, N, (1,), generator=gen, device=x.device)[0]]
                continue
            cand = torch.multinomial(d2 / total, trials, replacement=True, generator=gen)
            xc = x[cand]
            dist = (x2[None, :] + (xc * xc).sum(1)[:, None] - 2.0 * (xc @ x.T)).clamp_(min=0.0)  # [trials, N]
            dc = torch.minimum(d2[None, :], dist)
            best = int(torch.argmin(dc.sum(1)).item())
            cb[c] = x[cand[best]]
            d2 = dc[best]
        return cb

    def fit(self, x):
        """vq.py:34-50 (k-means++ seeding, Lloyd iterations, best of ``n_init`` runs by inertia)."""
        if _is_np(x):
            assert x.dtype == np.float32
        else:
            assert x.dtype == torch.float32
        assert x.ndim == 2
        x = ops.to_dev(x, torch.float32)
        N, D = x.shape
        C = self.n_clusters
        if N < C:  # sklearn: ValueError
            raise ValueError(f'n_samples={N} should be >= n_clusters={C}.')
        gen = torch.Generator(device=x.device)
        if self.seed is not None:
            gen.manual_seed(int(self.seed))
        else:
            gen.seed()
        tol = 1e-4 * float(x.var(dim=0, unbiased=False).mean().item())
        best, best_inertia = None, float('inf')
        for _ in range(max(1, self.n_init)):
            cb = self._kmeanspp(x, C, gen)
            for _it in range(max(1, self.iter)):
                new, counts, _ = self._lloyd_step(x, cb)
                empty = torch.nonzero(counts == 0).flatten()
                if empty.numel():  # relocate empty clusters onto random rows (sklearn relocates them too)
                    new[empty] = x[torch.randint(0, N, (empty.numel(),), generator=gen, device=x.device)]
                shift = float(((new - cb) ** 2).sum().item())
                cb = new
                if shift <= tol:
                    break
            _, _, inertia = self._lloyd_step(x, cb)
            if float(inertia.item()) < best_inertia:
                best, best_inertia = cb, float(inertia.item())
        self._set_codebook(best)

    def partial_fit(self, x):
        """vq.py:52-69 (MiniBatchKMeans): streaming update -- the first batch seeds the centres, every batch
        moves a centre to the running mean of all rows assigned to it so far."""
        assert x.ndim == 2
        x = ops.to_dev(x, torch.float32)
        C = self.n_clusters
        if self._sums is None:
            assert x.shape[0] >= C, f'n_samples={x.shape[0]} should be >= n_clusters={C}'
            gen = torch.Generator(device=x.device)
            if self.seed is not None:
                gen.manual_seed(int(self.seed))
            else:
                gen.seed()
            self._pf_cb = x[torch.randperm(x.shape[0], generator=gen, device=x.device)[:C]].clone()
            self._sums = torch.zeros_like(self._pf_cb)
            self._counts = torch.zeros((C,), dtype=torch.float32, device=x.device)
        a = self._assign(x, self._pf_cb)
        self._sums.index_add_(0, a, x)
        self._counts.index_add_(0, a, torch.ones_like(a, dtype=torch.float32))
        self._pf_cb = torch.where(self._counts[:, None] > 0, self._sums / self._counts.clamp(min=1.0)[:, None], self._pf_cb)

    def build_codebook(self):
        """vq.py:71-76."""
        assert self._sums is not None, 'partial_fit has not been called'
        self._set_codebook(self._pf_cb)

    # ------------------------------------------------------------------ use
    def encode(self, x):
        """vq.py:78-90: id of the closest centroid (squared L2, first minimum) per row; numpy in -> numpy out."""
        self._check_trained()
        if _is_np(x):
            assert x.dtype == np.float32
        assert x.ndim == 2
        cells = self._assign(ops.to_dev(x, torch.float32), self.codebook_dev)
        return cells.cpu().numpy().astype(np.int32) if _is_np(x) else cells

    def decode(self, x):  # vq.py:92-93
        return None
