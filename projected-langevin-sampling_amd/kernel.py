"""Base kernels k and the PLS kernel r (drop-in for src/projected_langevin_sampling/kernel.py:5-79).

The reference takes a gpytorch kernel object; gpytorch is a host-side hyper-parameter container there.
Here kernels are plain parameter holders whose Gram matrices are built by libplship on the MI355X.
A gpytorch ScaleKernel(RBFKernel) instance is accepted wherever a base kernel is expected: its
lengthscale / outputscale are read once (``as_base_kernel``)."""
from __future__ import annotations

import torch

from . import _lib as L


def _dev(x: torch.Tensor) -> torch.Tensor:
    """float64 contiguous copy on the current GPU (inputs may arrive as CPU / float32 tensors, as in the reference)."""
    if not torch.cuda.is_available():
        raise L.PlsHipError("no MI355X visible: the projected-Langevin hot path has no CPU fallback")
    return x.detach().to(device="cuda", dtype=torch.float64).contiguous()


class BaseKernel:
    """k(x1, x2) -> dense (n1, n2) float64 device tensor."""

    kind: int
    outputscale: float = 1.0

    def _lengthscale_dev(self, d: int) -> torch.Tensor | None:
        return None

    def __call__(self, x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
        a, b = _dev(x1), _dev(x2)
        if a.dim() == 1:
            a = a[:, None]
        if b.dim() == 1:
            b = b[:, None]
        assert a.shape[1] == b.shape[1], "x1 and x2 must share the input dimension"
        n1, d = a.shape
        n2 = b.shape[0]
        out = torch.empty((n1, n2), dtype=torch.float64, device=a.device)
        ls = self._lengthscale_dev(d)
        L.check(
            L.load().pls_kernel_gram(
                self.kind, a.data_ptr(), n1, b.data_ptr(), n2, d, L.ptr(ls), float(self.outputscale), out.data_ptr(),
                max(n2, 1), L.stream_ptr(),
            ),
            "pls_kernel_gram",
        )
        return out

    forward = __call__


class ARDKernel(BaseKernel):
    """ScaleKernel(RBFKernel(ard_num_dims=D)): k(a,b) = outputscale * exp(-0.5 sum_d ((a_d-b_d)/lengthscale_d)^2)
    (constructed in the reference at experiments/uci/regression/main.py:171-173, README.md:144-146)."""

    kind = L.KERNEL_RBF_ARD

    def __init__(self, lengthscale, outputscale: float = 1.0):
        self.lengthscale = torch.as_tensor(lengthscale, dtype=torch.float64).reshape(-1).cpu()
        self.outputscale = float(outputscale)
        self._ls_dev: dict[int, torch.Tensor] = {}

    def _lengthscale_dev(self, d: int) -> torch.Tensor:
        if d not in self._ls_dev:
            ls = self.lengthscale
            if ls.numel() == 1:
                ls = ls.expand(d)
            assert ls.numel() == d, f"lengthscale has {ls.numel()} entries, data has {d} dims"
            self._ls_dev[d] = _dev(ls)
        return self._ls_dev[d]


class LinearKernel(BaseKernel):
    """k(x1, x2) = x1 x2^T: the reference's test double (mockers/kernel.py:8-23)."""

    kind = L.KERNEL_LINEAR


def as_base_kernel(kernel) -> BaseKernel:
    """Accept our kernels, or read the hyper-parameters of a gpytorch ScaleKernel(RBFKernel)."""
    if isinstance(kernel, BaseKernel):
        return kernel
    inner = getattr(kernel, "base_kernel", None)
    if inner is not None and hasattr(inner, "lengthscale") and hasattr(kernel, "outputscale"):
        return ARDKernel(
            lengthscale=torch.as_tensor(inner.lengthscale).detach().reshape(-1),
            outputscale=float(torch.as_tensor(kernel.outputscale).detach()),
        )
    raise TypeError(f"unsupported base kernel {type(kernel).__name__}: use ARDKernel / LinearKernel")


class PLSKernel:
    """r(x1, x2) = (1/n_S) k(x1, S) k(x2, S)^T over the unique approximation samples S (kernel.py:31-76)."""

    def __init__(self, base_kernel, approximation_samples: torch.Tensor, **kwargs):
        self.base_kernel = as_base_kernel(base_kernel)
        self.approximation_samples = approximation_samples

    def forward(
        self,
        x1: torch.Tensor,
        x2: torch.Tensor,
        additional_approximation_samples: torch.Tensor | None = None,
        last_dim_is_batch: bool = False,
        diag: bool = False,
        **params,
    ) -> torch.Tensor:
        samples = [self.approximation_samples.detach().cpu().to(torch.float64)]
        if additional_approximation_samples is not None:
            samples.append(additional_approximation_samples.detach().cpu().to(torch.float64))
        s = torch.cat([t if t.dim() == 2 else t[:, None] for t in samples], dim=0).unique(dim=0)  # kernel.py:43-45
        n_s = s.shape[0]
        g1 = self.base_kernel(s, x1)  # k(S, x1)  (n_S, n1): k-major operand
        g2 = self.base_kernel(s, x2)  # k(S, x2)  (n_S, n2)
        n1, n2 = g1.shape[1], g2.shape[1]
        res = torch.empty((n1, n2), dtype=torch.float64, device=g1.device)
        L.check(
            L.load().pls_gemm_tn(
                g1.data_ptr(), L.ld(g1), g2.data_ptr(), L.ld(g2), res.data_ptr(), max(n2, 1), n1, n2, n_s, 1.0 / n_s, 0.0,
                L.stream_ptr(),
            ),
            "pls_gemm_tn",
        )
        return res.diag() if diag else res

    __call__ = forward
