"""Product-quantisation codec on MI355X -- drop-in for ``annlite.core.codec.pq.PQCodec``.

Same constructor, method names, argument meaning, return shapes/dtypes and assertion behaviour as
the reference (annlite/core/codec/pq.py:16-368); each method cites the lines it mirrors.  All
arithmetic runs in the HIP kernels of ``libannlite_hip.so``:

  ======================  ===========================================  ==========================
  method                  reference                                    kernel (annlite_amd/csrc)
  ======================  ===========================================  ==========================
  fit / partial_fit       pq.py:89-156 (sklearn KMeans / MiniBatch)    codec.hip  encode_kernel<ACCUM>
  encode                  pq.py:158-177 (scipy vq per sub-space)       codec.hip  encode_kernel
  decode                  pq.py:179-198                                codec.hip  decode_kernel
  precompute_adc          pq.py:200-224 -> pyx:85-145                  lut.hip    lut_bmk_kernel
  get_dist_mat            pq.py:293-325 -> pyx:149-274                 lut.hip    (L2 VALU, IP MFMA)
  DistanceTable.adist     pq.py:350-368 -> pyx:52-80                   scan.hip   adc_dist_kernel
  ======================  ===========================================  ==========================

numpy in -> numpy out (like the reference); torch device tensors in -> torch device tensors out.
The object is picklable (codebooks travel as numpy; ``BaseCodec.dump/load``, codec/base.py:26-31).
It also satisfies the duck-type ``hnsw_bind._loadPQ`` checks (bindings/hnsw_bindings.cpp:851-928:
``encode``, ``get_codebook``, ``get_subspace_splitting``).
"""
from typing import Optional

import numpy as np
import torch

from ... import ops
from ..._capi import LAYOUT_BMK, LAYOUT_TILED, LUT_IPDIST, LUT_L2
from ...enums import Metric
from .base import BaseCodec


def _is_np(x) -> bool:
    return not isinstance(x, torch.Tensor)


class PQCodec(BaseCodec):
    """Product Quantization codec (see module docstring).

    :param dim: dimensionality of input vectors
    :param n_subvectors: number of sub-spaces ``M``
    :param n_clusters: codewords per sub-space ``Ks`` (<=256 -> uint8 codes, <=65536 -> uint16)
    :param metric: ``Metric.EUCLIDEAN`` / ``INNER_PRODUCT`` / ``COSINE``
    :param n_init: number of k-means restarts in :meth:`fit`; the best run per sub-space is kept
    """

    SEED_ROWS_PER_CENTRE = 256  # k-means++ seeding looks at no more than this many rows per centre (see _kmeanspp_centres)

    def __init__(
        self,
        dim: int,
        n_subvectors: int = 8,
        n_clusters: int = 256,
        metric: Metric = Metric.EUCLIDEAN,
        n_init: int = 4,
    ):
        super(PQCodec, self).__init__(require_train=True)
        self.dim = dim
        self.n_subvectors = n_subvectors
        self.n_clusters = n_clusters

        assert dim % n_subvectors == 0, 'input dimension must be dividable by number of sub-space'  # pq.py:51-53
        self.d_subvector = dim // n_subvectors

        self.code_dtype = (
            np.uint8 if n_clusters <= 2 ** 8 else (np.uint16 if n_clusters <= 2 ** 16 else np.uint32)
        )  # pq.py:56-60
        self.metric = metric
        self.normalize_input = self.metric == Metric.COSINE  # pq.py:67-69

        self._codebooks = np.zeros((self.n_subvectors, self.n_clusters, self.d_subvector), dtype=np.float32)
        self.kmeans = []  # attribute kept for API compatibility (the reference stores sklearn objects here)
        self.n_init = n_init
        self.seed: Optional[int] = None  # set for reproducible training (the reference is unseeded)
        # With a seed the initial centres repeat, the Lloyd steps' float atomics do not: codebooks of two runs agree to the last
        # few bits only.  ``deterministic = True`` accumulates the cluster sums in a fixed order instead (``_assign_accumulate_det``):
        # bit-identical codebooks run after run -- what lets bench.py's result digests be compared ACROSS runs (N = 1 against N = 8)
        self.deterministic: bool = False

        self._cb_dev: dict = {}  # device copies of the codebooks, one per HIP device that asked (a multi-GPU index shares ONE codec)
        self._pf = None  # streaming k-means state of partial_fit: (centers, sums, counts) device tensors

    # ------------------------------------------------------------------ pickling / device cache
    def __getstate__(self):
        st = self.__dict__.copy()
        st['_cb_dev'] = {}
        if self._pf is not None:
            st['_pf'] = tuple(t.cpu().numpy() for t in self._pf)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        if not isinstance(self.__dict__.get('_cb_dev'), dict):  # (files written before the per-device cache)
            self._cb_dev = {}
        if self._pf is not None and isinstance(self._pf[0], np.ndarray):
            self._pf_np = self._pf
            self._pf = None

    def __hash__(self):  # pq.py:77-87
        return hash((self.__class__.__name__, self.dim, self.n_subvectors, self.n_clusters, self.metric, self.code_dtype))

    @property
    def codebooks(self) -> np.ndarray:  # pq.py:226-228
        return self._codebooks

    @property
    def codebooks_dev(self) -> torch.Tensor:
        """f32 [M, Ks, dsub] on the current HIP device (cached)."""
        dev = ops.device()
        cb = self._cb_dev.get(dev)
        if cb is None:
            cb = self._cb_dev[dev] = ops.to_dev(np.ascontiguousarray(self._codebooks, dtype=np.float32))
        return cb

    def _set_codebooks(self, cb_dev: torch.Tensor):
        cb_dev = cb_dev.contiguous()
        self._cb_dev = {cb_dev.device: cb_dev}
        self._codebooks = cb_dev.cpu().numpy()

    def set_codebooks(self, codebooks) -> 'PQCodec':
        """Install externally trained codebooks [M, Ks, dsub] (numpy or torch) and mark trained."""
        cb = codebooks if isinstance(codebooks, np.ndarray) else codebooks.detach().cpu().numpy()
        assert cb.shape == (self.n_subvectors, self.n_clusters, self.d_subvector)
        self._codebooks = np.ascontiguousarray(cb, dtype=np.float32)
        self._cb_dev = {}
        self._is_trained = True
        return self

    # ------------------------------------------------------------------ training
    def _prep_train(self, x) -> torch.Tensor:
        x = ops.to_dev(x, torch.float32)
        if self.normalize_input:
            x = ops.l2_normalize(x)  # pq.py:100-101 / 125-126
        return x

    def _random_centres(self, x: torch.Tensor, gen: torch.Generator) -> torch.Tensor:
        N = x.shape[0]
        M, Ks, ds = self.n_subvectors, self.n_clusters, self.d_subvector
        if N < Ks:  # sklearn: ValueError (pq.py:106-110 -> KMeans.fit)
            raise ValueError(f'n_samples={N} should be >= n_clusters={Ks}.')
        cb = torch.empty((M, Ks, ds), dtype=torch.float32, device=x.device)
        for m in range(M):
            idx = torch.randperm(N, generator=gen, device=x.device)[:Ks]
            cb[m] = x[idx, m * ds:(m + 1) * ds]
        return cb

    def _kmeanspp_centres(self, x: torch.Tensor, gen: torch.Generator) -> torch.Tensor:
        """k-means++ seeding (sklearn's default ``init`` behind pq.py:106-110), all M sub-spaces at once: every next
        centre is drawn with probability proportional to the squared distance to the nearest centre chosen so far,
        the best of ``2 + log(Ks)`` draws per step (sklearn's greedy variant).  Batched tensor ops on the device, no
        host synchronisation inside the Ks steps.  Seeding runs on at most ``SEED_ROWS_PER_CENTRE * Ks`` rows drawn at
        random (65 536 at Ks = 256): every step materialises [M, trials, rows] temporaries -- on the whole of a large training
        set that was gigabytes per step, 255 times, and ``torch.multinomial`` refuses more than 2^24 categories; a seeding
        only needs to see every region of the data, Lloyd's iterations then run on ALL rows."""
        M, Ks, ds = self.n_subvectors, self.n_clusters, self.d_subvector
        if x.shape[0] < Ks:  # sklearn: ValueError (pq.py:106-110 -> KMeans.fit)
            raise ValueError(f'n_samples={x.shape[0]} should be >= n_clusters={Ks}.')
        cap = self.SEED_ROWS_PER_CENTRE * Ks
        if x.shape[0] > cap:
            x = x[torch.randperm(x.shape[0], generator=gen, device=x.device)[:cap]]
        N = x.shape[0]
        dev = x.device
        xs = x.reshape(N, M, ds).permute(1, 0, 2).contiguous()  # [M, N, ds]
        x2 = (xs * xs).sum(2)  # [M, N]
        cb = torch.empty((M, Ks, ds), dtype=torch.float32, device=dev)
        ar = torch.arange(M, device=dev)
        first = torch.randint(0, N, (M,), generator=gen, device=dev)
        cb[:, 0] = xs[ar, first]
        d2 = ((xs - cb[:, :1]) ** 2).sum(2)  # [M, N]
        trials = 2 + int(np.log(Ks))
        for c in range(1, Ks):
            total = d2.sum(1, keepdim=True)
            # (fewer distinct rows than centres: distances all zero -> uniform draws)
            prob = torch.where(total > 0, d2 / total.clamp(min=1e-30), torch.full_like(d2, 1.0 / N))
            cand = torch.multinomial(prob, trials, replacement=True, generator=gen)  # [M, trials]
            xc = torch.gather(xs, 1, cand[:, :, None].expand(M, trials, ds))  # [M, trials, ds]
            dist = (x2[:, None, :] + (xc * xc).sum(2)[:, :, None] - 2.0 * torch.bmm(xc, xs.transpose(1, 2))).clamp_(min=0.0)
            dc = torch.minimum(d2[:, None, :], dist)  # [M, trials, N]
            best = torch.argmin(dc.sum(2), dim=1)  # [M]
            cb[:, c] = xc[ar, best]
            d2 = dc[ar, best]
        return cb

    def _assign_accumulate_det(self, x: torch.Tensor, cb: torch.Tensor, sums: torch.Tensor, counts: torch.Tensor, inertia: torch.Tensor):
        """What ``annlite_kmeans_assign_accumulate`` leaves in (sums, counts, inertia), in a FIXED summation order: the
        assignment is the encode kernel's (first minimum), the rows of a cluster are laid out side by side in assignment order
        (a stable sort: unique scatter targets) and summed along that axis in float64 by ``torch.sum`` -- no atomics anywhere."""
        M, Ks, ds = self.n_subvectors, self.n_clusters, self.d_subvector
        N = x.shape[0]
        codes = ops.pq_encode(x, cb).to(torch.int64)  # [N, M]
        ar = torch.arange(N, device=x.device)
        for m in range(M):
            idx = codes[:, m]
            order = torch.argsort(idx, stable=True)
            cnt = torch.bincount(idx, minlength=Ks)
            start = torch.cumsum(cnt, 0) - cnt  # (integers: exact whatever the order)
            sidx = idx[order]
            pos = ar - start[sidx]
            xm = x[:, m * ds:(m + 1) * ds][order].to(torch.float64)
            mat = torch.zeros((Ks, int(cnt.max().item()) if N else 1, ds), dtype=torch.float64, device=x.device)
            mat[sidx, pos] = xm
            s64 = mat.sum(1)
            sums[m] = s64.to(torch.float32)
            counts[m] = cnt.to(torch.int32)
            diff = xm - cb[m].to(torch.float64)[sidx]
            inertia[m] = (diff * diff).sum(1).sum()

    def _init_centres(self, x: torch.Tensor, gen: torch.Generator) -> torch.Tensor:
        return self._random_centres(x, gen) if getattr(self, 'init', 'k-means++') == 'random' else self._kmeanspp_centres(x, gen)

    def fit(self, x, iter: int = 100):
        """Train one k-means per sub-space (pq.py:89-115: sklearn ``KMeans(n_clusters, max_iter=iter,
        n_init)``).  Lloyd iterations run on the GPU: assignment + accumulation is ONE kernel over all
        sub-spaces (``annlite_kmeans_assign_accumulate``), the update another.  k-means++ seeding (sklearn's default;
        ``self.init = 'random'`` for random rows), ``n_init`` restarts,
        best inertia kept per sub-space, early stop on centre shift <= 1e-4 * mean variance (sklearn's
        ``tol``).  Like the reference the result is not bit-reproducible unless ``self.seed`` is set."""
        if _is_np(x):
            assert x.dtype == np.float32  # pq.py:97-98
        else:
            assert x.dtype == torch.float32
        assert x.ndim == 2
        x = self._prep_train(x)
        N, D = x.shape
        assert D == self.dim
        M, Ks, ds = self.n_subvectors, self.n_clusters, self.d_subvector
        dev = x.device
        gen = torch.Generator(device=dev)
        if self.seed is not None:
            gen.manual_seed(int(self.seed))
        else:
            gen.seed()
        var = x.var(dim=0, unbiased=False).reshape(M, ds).mean(dim=1)  # per sub-space mean variance
        tol = 1e-4 * var
        best_cb = torch.zeros((M, Ks, ds), dtype=torch.float32, device=dev)
        best_inertia = torch.full((M,), float('inf'), dtype=torch.float64, device=dev)
        sums = torch.empty((M, Ks, ds), dtype=torch.float32, device=dev)
        counts = torch.empty((M, Ks), dtype=torch.int32, device=dev)
        inertia = torch.empty((M,), dtype=torch.float64, device=dev)
        for _ in range(max(1, int(self.n_init))):
            cb = self._init_centres(x, gen)
            accumulate = self._assign_accumulate_det if getattr(self, 'deterministic', False) else ops.kmeans_assign_accumulate
            for it in range(max(1, int(iter))):
                sums.zero_(); counts.zero_(); inertia.zero_()
                accumulate(x, cb, sums, counts, inertia)
                old = cb.clone()
                ops.kmeans_update(sums, counts, cb)
                empty = counts == 0
                if bool(empty.any()):
                    # re-seed empty clusters with random training rows (sklearn relocates them too)
                    for m in torch.nonzero(empty.any(dim=1)).flatten().tolist():
                        ks = torch.nonzero(empty[m]).flatten()
                        idx = torch.randint(0, N, (ks.numel(),), generator=gen, device=dev)
                        cb[m, ks] = x[idx, m * ds:(m + 1) * ds]
                shift = ((cb - old) ** 2).sum(dim=(1, 2))
                if bool((shift <= tol).all()):
                    break
            sums.zero_(); counts.zero_(); inertia.zero_()
            accumulate(x, cb, sums, counts, inertia)
            better = inertia < best_inertia
            best_cb[better] = cb[better]
            best_inertia = torch.where(better, inertia, best_inertia)
        self._set_codebooks(best_cb)
        self.inertia_ = best_inertia.cpu().numpy()
        self._is_trained = True

    def partial_fit(self, x):
        """Streaming update (pq.py:117-142: sklearn ``MiniBatchKMeans.partial_fit`` per sub-space).
        Same update rule in aggregate: every centre is the running mean of all points ever assigned
        to it (per-centre learning rate 1/count); the first batch seeds the centres."""
        assert x.ndim == 2
        x = self._prep_train(x)
        M, Ks, ds = self.n_subvectors, self.n_clusters, self.d_subvector
        dev = x.device
        if self._pf is None and getattr(self, '_pf_np', None) is not None:
            self._pf = tuple(ops.to_dev(a) for a in self._pf_np)
            self._pf_np = None
        if self._pf is None:
            gen = torch.Generator(device=dev)
            if self.seed is not None:
                gen.manual_seed(int(self.seed))
            else:
                gen.seed()
            centres = self._init_centres(x, gen)
            sums = torch.zeros((M, Ks, ds), dtype=torch.float32, device=dev)
            counts = torch.zeros((M, Ks), dtype=torch.int32, device=dev)
            self._pf = (centres, sums, counts)
        centres, sums, counts = self._pf
        ops.kmeans_assign_accumulate(x, centres, sums, counts, None)
        ops.kmeans_update(sums, counts, centres)

    def build_codebook(self):
        """pq.py:144-156 -- publish the streaming centres as the codebooks."""
        if self._pf is None and getattr(self, '_pf_np', None) is not None:
            self._pf = tuple(ops.to_dev(a) for a in self._pf_np)
            self._pf_np = None
        assert self._pf is not None, 'call partial_fit first'
        self._set_codebooks(self._pf[0].clone())
        self._is_trained = True

    # ------------------------------------------------------------------ encode / decode
    def encode(self, x):
        """pq.py:158-177 -- codes[n, m] = argmin_k |x[n, m-th slice] - C[m, k]|^2, first minimum wins.
        Does NOT normalise (callers do, hnsw/index.py:28-29).  Returns [N, M] of ``code_dtype``."""
        if _is_np(x):
            assert x.dtype == np.float32  # pq.py:164
        else:
            assert x.dtype == torch.float32
        assert x.ndim == 2
        N, D = x.shape
        assert D == self.d_subvector * self.n_subvectors, 'input dimension must be Ds * M'
        codes = ops.pq_encode(ops.to_dev(x), self.codebooks_dev)
        if _is_np(x):
            return ops.codes_to_numpy(codes).astype(self.code_dtype, copy=False)
        return codes

    def decode(self, codes):
        """pq.py:179-198 -- reconstruct by gathering codewords; [N, M] -> f32 [N, D]."""
        assert codes.ndim == 2
        N, M = codes.shape
        assert M == self.n_subvectors
        if _is_np(codes):
            assert codes.dtype == self.code_dtype  # pq.py:190
        out = ops.pq_decode(ops.to_dev(codes), self.codebooks_dev)
        return out.cpu().numpy() if _is_np(codes) else out

    # ------------------------------------------------------------------ distance tables
    def precompute_adc(self, query) -> 'DistanceTable':
        """pq.py:200-224 -- single-query table; ALWAYS squared L2 and never normalised, whatever the
        metric (reference quirk)."""
        if _is_np(query):
            assert query.dtype == np.float32
        assert query.ndim == 1, 'input must be a single vector'
        q = ops.to_dev(query, torch.float32)
        dtable = ops.lut_build(q[None, :], self.codebooks_dev, LUT_L2, LAYOUT_BMK)[0]
        return DistanceTable(dtable.cpu().numpy() if _is_np(query) else dtable)

    def get_codebook(self) -> np.ndarray:  # pq.py:231-237
        return np.ascontiguousarray(self.codebooks, dtype='float32')

    def get_subspace_splitting(self):  # pq.py:239-244
        return (self.n_subvectors, self.n_clusters, self.d_subvector)

    def scan_inputs(self, x: torch.Tensor):
        """(LUT kind, queries as the table build sees them) for ``get_dist_mat`` on device input: the metric
        dispatch of pq.py:309-322 without building the tables (``ops.pq_search_topk`` builds them itself)."""
        if self.normalize_input:
            x = ops.l2_normalize(x)  # pq.py:309-310 (yes: again, even if the caller normalised)
        if self.metric == Metric.EUCLIDEAN:
            kind = LUT_L2
        elif self.metric in (Metric.INNER_PRODUCT, Metric.COSINE):
            kind = LUT_IPDIST  # float32(1/n_clusters) - <q_sub, codeword>, pq.py:316-322
        else:
            raise ValueError(f'Unable support metrics {self.metric}')
        return kind, x

    def _dist_mat_dev(self, x: torch.Tensor, layout: int, qi: int = 4) -> torch.Tensor:
        kind, x = self.scan_inputs(x)
        return ops.lut_build(x, self.codebooks_dev, kind, layout, qi)

    def get_dist_mat(self, x):
        """pq.py:293-325 -- batched tables for the codec's metric: [B, M, Ks] C-contiguous float32."""
        if _is_np(x):
            assert x.dtype == np.float32  # pq.py:303
        else:
            assert x.dtype == torch.float32
        assert x.ndim == 2
        N, D = x.shape
        assert D == self.d_subvector * self.n_subvectors, 'input dimension must be Ds * M'
        if _is_np(x) and self.normalize_input:
            # host buffers: the normalisation of pq.py:309-310 in the reference's own numpy arithmetic (bit-equal tables)
            from ...math import l2_normalize_host

            kind, _ = self.scan_inputs(torch.empty((0, D), dtype=torch.float32, device=ops.device()))
            out = ops.lut_build(ops.to_dev(l2_normalize_host(x)), self.codebooks_dev, kind, LAYOUT_BMK)
        else:
            out = self._dist_mat_dev(ops.to_dev(x), LAYOUT_BMK)
        return np.ascontiguousarray(out.cpu().numpy(), dtype='float32') if _is_np(x) else out

    def get_dist_mat_tiled(self, x_dev: torch.Tensor, qi: int) -> torch.Tensor:
        """Same tables, written directly in the scan kernel's LDS-friendly layout (device only)."""
        assert x_dev.dtype == torch.float32 and x_dev.ndim == 2
        assert x_dev.shape[1] == self.d_subvector * self.n_subvectors, 'input dimension must be Ds * M'
        return self._dist_mat_dev(x_dev, LAYOUT_TILED, qi)


class DistanceTable(object):
    """pq.py:330-368 -- one query's table ``dtable`` [M, Ks] and the flat ADC scan over codes."""

    def __init__(self, dtable):
        assert dtable.ndim == 2
        self.dtable = dtable

    def adist(self, codes):
        """pq.py:350-368 -> pq_bind.dist_pqcodes_to_codebooks: d[n] = sum_m dtable[m, codes[n, m]]
        (float32 array; the reference returns the same numbers as a python list)."""
        assert codes.ndim == 2
        out = ops.adc_dist(ops.to_dev(self.dtable, torch.float32), ops.to_dev(codes))
        return out.cpu().numpy() if _is_np(codes) else out
