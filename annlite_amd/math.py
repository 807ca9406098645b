"""Device-side mirror of ``annlite/math.py`` (reference lines cited per function).

Inputs may be numpy arrays or torch tensors; numpy in -> numpy out (so the reference's own call
sites / tests read the same), torch in -> torch out (stays in HBM).  The arithmetic runs on the GPU, except the row
normalisation of host buffers (see ``l2_normalize_host``).
"""
from typing import Tuple

import numpy as np
import torch

from . import ops


def _wrap(x):
    is_np = isinstance(x, np.ndarray)
    return ops.to_dev(x), is_np


def l2_normalize_host(x: np.ndarray, eps: float = np.finfo(np.float32).eps) -> np.ndarray:
    """annlite/math.py:6-18 on HOST buffers, in the reference's own numpy arithmetic (einsum row sums, in-place sqrt,
    divide): what a numpy caller's vectors go through before they are uploaded, so that cosine tables and neighbour
    ids equal the reference's bit for bit -- the order in which einsum adds the squares belongs to the numpy build,
    the device kernel (one wave per row) adds in another and differs in the last ulp."""
    norms = np.einsum('ij,ij->i', x, x)
    np.sqrt(norms, norms)
    constant_mask = norms < 10 * eps
    norms[constant_mask] = 1.0
    return x / norms[:, np.newaxis]


def l2_normalize(x, eps: float = np.finfo(np.float32).eps):
    """annlite/math.py:6-18 -- rows with norm < 10*eps are left unscaled.  numpy in: the reference's arithmetic on the
    host (``l2_normalize_host``, bit-equal); torch in: the device kernel (fp32, the row sum in a different order than
    numpy's einsum: ~1 ulp)."""
    if isinstance(x, np.ndarray):
        assert x.ndim == 2
        return l2_normalize_host(x, eps)
    t, _ = _wrap(x)
    assert t.ndim == 2
    return ops.l2_normalize(t.float())


def top_k(values, k: int, descending: bool = False) -> Tuple:
    """annlite/math.py:94-120 -- k smallest per row, ascending; ties broken by index ascending
    (the reference leaves tie order to numpy's introselect).  k <= 64 runs on the wave-list kernel;
    larger k falls to a full device sort (torch.sort is stable => same tie rule)."""
    t, is_np = _wrap(values)
    assert t.ndim == 2
    if descending:
        t = -t
    B, N = t.shape
    kk = min(k, N)
    if 1 <= kk <= 64:
        d, i = ops.topk_rows(t.float().contiguous(), kk)
    else:
        d, i = torch.sort(t, dim=1, stable=True)
        d, i = d[:, :kk], i[:, :kk]
    if descending:
        d = -d
    if is_np:
        return d.cpu().numpy(), i.cpu().numpy()
    return d, i


def sqeuclidean(x_mat, y_mat):
    """annlite/math.py:41-51 (|y|^2 + |x|^2 - 2 x.y) -- brute-force ground truth for recall."""
    x, is_np = _wrap(x_mat)
    y, _ = _wrap(y_mat)
    d = (y * y).sum(1)[None, :] + (x * x).sum(1)[:, None] - 2.0 * (x @ y.T)
    return d.cpu().numpy() if is_np else d


def euclidean(x_mat, y_mat):
    """annlite/math.py:54-61"""
    d = sqeuclidean(x_mat, y_mat)
    return np.sqrt(d) if isinstance(d, np.ndarray) else torch.sqrt(d)


def cosine(x_mat, y_mat, eps: float = np.finfo(np.float32).eps):
    """annlite/math.py:21-38"""
    x, is_np = _wrap(x_mat)
    y, _ = _wrap(y_mat)
    d = 1 - torch.clip((x @ y.T + eps) / (torch.outer(torch.linalg.norm(x, dim=1), torch.linalg.norm(y, dim=1)) + eps), -1, 1)
    return d.cpu().numpy() if is_np else d


def cdist(x_mat, y_mat, metric: str):
    """annlite/math.py:77-91"""
    return {'cosine': cosine, 'sqeuclidean': sqeuclidean, 'euclidean': euclidean}[metric](x_mat, y_mat)


def pdist(x_mat, metric: str):
    """annlite/math.py:64-74: all pairwise distances of one set"""
    return cdist(x_mat, x_mat, metric)
