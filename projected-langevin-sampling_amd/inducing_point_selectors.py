"""Inducing-point selection (drop-in for src/inducing_point_selectors/{base,conditional_variance,random}.py;
SURVEY.md 8f row N3).  The greedy conditional-variance rule runs as libplship kernels: every iteration's pivot stays on
the device, and no N x N Gram matrix is formed (the reference builds one only to read its diagonal)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib as L
from .kernel import PLSKernel, _dev, as_base_kernel


class InducingPointSelector(ABC):
    @abstractmethod
    def compute_induce_data(self, x: torch.Tensor, m: int, kernel, **params) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError

    def __call__(self, x: torch.Tensor, m: int, kernel, **params) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.compute_induce_data(x=x, m=m, kernel=kernel, **params)


class RandomInducingPointSelector(InducingPointSelector):
    """inducing_point_selectors/random.py:9-18."""

    def compute_induce_data(self, x: torch.Tensor, m: int, kernel=None, **params) -> Tuple[torch.Tensor, torch.Tensor]:
        indices = torch.randperm(x.shape[0])[:m]
        return x[indices, ...], indices


class ConditionalVarianceInducingPointSelector(InducingPointSelector):
    """Greedy MAP for a DPP == partial pivoted Cholesky of k(X, X) (conditional_variance.py:11-120)."""

    def __init__(self, threshold: Optional[float] = 0.0):
        self.threshold = threshold

    def compute_induce_data(self, x: torch.Tensor, m: int, kernel, jitter: float = 1e-12) -> Tuple[torch.Tensor, torch.Tensor]:
        assert m > 1, "Must have at least 2 inducing points"  # conditional_variance.py:57
        if isinstance(kernel, PLSKernel):
            # the reference would select on r = k(., S) k(S, .)^T / |S| (an N x N Gram of the PLS kernel); its experiments
            # always pass the base ScaleKernel(RBFKernel) (experiments/uci/regression/main.py:203), and so must callers here
            raise TypeError("ConditionalVarianceInducingPointSelector selects on a base kernel (ARDKernel / LinearKernel / a "
                            "gpytorch ScaleKernel(RBFKernel)); pass pls_kernel.base_kernel, not the PLSKernel")
        base = as_base_kernel(kernel)  # ARDKernel / LinearKernel as is; ScaleKernel(RBFKernel): lengthscale AND outputscale
        n = x.shape[0]
        perm = np.random.permutation(n)  # permute entries so tie-breaking is random (:58-61)
        xp = x[torch.as_tensor(perm)] if isinstance(x, torch.Tensor) else torch.as_tensor(x)[perm]
        xd = _dev(xp if xp.dim() == 2 else xp[:, None])
        d = xd.shape[1]
        ls = base._lengthscale_dev(d)
        lib = L.load()
        ws_bytes = lib.pls_select_inducing_workspace_bytes(n, m)
        ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.float64, device=xd.device)
        idx = torch.full((m,), n, dtype=torch.int64, device=xd.device)  # the reference's out-of-range sentinel (:63)
        count = torch.zeros(1, dtype=torch.int64, device=xd.device)
        L.check(
            lib.pls_select_inducing_conditional_variance(
                base.kind, xd.data_ptr(), n, d, L.ptr(ls), float(base.outputscale), m, float(jitter),
                float(self.threshold if self.threshold is not None else -1.0), idx.data_ptr(), count.data_ptr(), ws.data_ptr(),
                ws_bytes, L.stream_ptr(),
            ),
            "pls_select_inducing_conditional_variance",
        )
        k = int(count.item())
        if k < m:
            print("ConditionalVariance: Terminating selection of inducing points early.")  # :110-112
        idx_host = idx[:k].cpu().numpy()
        return xp[torch.as_tensor(idx_host)], torch.from_numpy(perm[idx_host])
