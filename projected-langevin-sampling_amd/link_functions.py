"""Link functions (drop-in for src/projected_langevin_sampling/link_functions.py:6-80).
transform() runs libplship's element-wise kernel on device tensors."""
from __future__ import annotations

from abc import ABC, abstractmethod

import torch

from . import _lib as L


class PLSLinkFunction(ABC):
    """Transforms prediction samples to the output space (link_functions.py:6-27)."""

    #: libplship link id; None for user-defined links (those run as ordinary torch code)
    kind: int | None = None
    jitter: float = 1e-10

    @abstractmethod
    def transform(self, y: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def __call__(self, *args, **kwargs):
        return self.transform(*args, **kwargs)

    def _native_transform(self, y: torch.Tensor, col_offset: torch.Tensor | None = None) -> torch.Tensor:
        """link(y + col_offset[None, :]) in one libplship kernel."""
        y = L.require_gpu_tensor(y, "y", promote=True)
        y2 = y.reshape(1, -1) if y.dim() != 2 else y
        y2 = y2 if y2.stride(-1) == 1 else y2.contiguous()
        out = torch.empty(y2.shape, dtype=torch.float64, device=y.device)
        L.check(
            L.load().pls_link_transform(
                self.kind, float(self.jitter), y2.data_ptr(), L.ld(y2), y2.shape[0], y2.shape[1],
                None if col_offset is None else L.require_gpu_tensor(col_offset, "col_offset").contiguous().data_ptr(),
                out.data_ptr(), L.ld(out), L.stream_ptr(),
            ),
            "pls_link_transform",
        )
        return out.reshape(y.shape)


class IdentityLinkFunction(PLSLinkFunction):
    """link_functions.py:48-55."""

    kind = L.LINK_IDENTITY

    def transform(self, y: torch.Tensor) -> torch.Tensor:
        return y  # link_functions.py:54-55 returns its argument


class SquareLinkFunction(PLSLinkFunction):
    """link_functions.py:73-80."""

    kind = L.LINK_SQUARE

    def transform(self, y: torch.Tensor) -> torch.Tensor:
        return self._native_transform(y)


class SigmoidLinkFunction(PLSLinkFunction):
    """link_functions.py:58-70 (clip to [jitter, 1 - jitter])."""

    kind = L.LINK_SIGMOID

    def __init__(self, jitter: float = 1e-10):
        self.jitter = jitter

    def transform(self, y: torch.Tensor) -> torch.Tensor:
        return self._native_transform(y)


class ProbitLinkFunction(PLSLinkFunction):
    """link_functions.py:30-45.  sqrt(2) is exact fp64 here; the reference evaluates it in torch's default
    dtype (identical under the float64 default its experiments set)."""

    kind = L.LINK_PROBIT

    def __init__(self, jitter: float = 1e-10):
        self.jitter = jitter

    def transform(self, y: torch.Tensor) -> torch.Tensor:
        return self._native_transform(y)
