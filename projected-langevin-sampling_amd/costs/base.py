"""Cost base class (drop-in for src/projected_langevin_sampling/costs/base.py:8-133).

Native costs carry a libplship descriptor: their values / derivatives run as HIP kernels, and PLS fuses the
derivative into the GEMM epilogue of the Langevin step.  A user-defined subclass that overrides
calculate_cost / calculate_cost_derivative with torch code still works: PLS then composes the un-fused
entry points (pls_onb_forward -> user code -> pls_onb_particle_update)."""
from __future__ import annotations

from abc import ABC, abstractmethod

import torch

from .. import _lib as L
from ..kernel import _dev
from ..link_functions import PLSLinkFunction


class PLSCost(ABC):
    #: libplship cost id; None for user-defined costs
    cost_kind: int | None = None

    def __init__(self, link_function: PLSLinkFunction, observation_noise: float | None = None):
        self.observation_noise = observation_noise
        self.link_function = link_function
        self._y_dev: torch.Tensor | None = None

    # ---- native plumbing ------------------------------------------------------------------------------------
    def _params(self) -> tuple:
        return (0.0, 0.0, 0.0, 0.0)

    def _reference_closed_form_link(self) -> int | None:
        """Link id for which the reference's dispatch uses its closed form (else it falls back to autograd)."""
        return None

    def is_native(self) -> bool:
        """True if both the cost and its link are evaluated by libplship (fused-step eligible)."""
        key = (type(self), id(self.link_function))  # (the answer only depends on the classes involved: cached)
        cached = getattr(self, "_native_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        value = self._is_native_uncached()
        self._native_cache = (key, value)
        return value

    def _is_native_uncached(self) -> bool:
        overridden = any(
            getattr(type(self), name) is not getattr(_native_base_of(type(self)), name)
            for name in ("calculate_cost", "calculate_cost_derivative")
        )
        return self.cost_kind is not None and getattr(self.link_function, "kind", None) is not None and not overridden

    def y_device(self) -> torch.Tensor:
        y = self.y_train
        if self._y_dev is None or self._y_src is not y:
            self._y_dev = _dev(y.reshape(-1))
            self._y_src = y
        return self._y_dev

    def desc(self, force_autograd: bool = False) -> L.CostDesc:
        if not self.is_native():
            raise L.PlsHipError(f"{type(self).__name__} has no native descriptor")
        d = L.CostDesc()
        d.cost = self.cost_kind
        d.link = self.link_function.kind
        d.deriv_mode = L.DERIV_AUTOGRAD if force_autograd else L.DERIV_REFERENCE
        p = self._params()
        for i in range(4):
            d.p[i] = float(p[i])
        d.jitter = float(getattr(self.link_function, "jitter", 1e-10))
        return d

    # ---- reference API --------------------------------------------------------------------------------------
    @abstractmethod
    def predict(self, prediction_samples: torch.Tensor) -> torch.distributions.Distribution:
        raise NotImplementedError()

    def calculate_cost(self, untransformed_train_prediction_samples: torch.Tensor) -> torch.Tensor:
        """(N, J) -> (J,): sum_n cost(y_n, f_nj) (e.g. gaussian.py:63-73)."""
        f = L.require_gpu_tensor(untransformed_train_prediction_samples, "untransformed_train_prediction_samples", promote=True)
        f = f if f.stride(-1) == 1 else f.contiguous()
        n, j = f.shape
        y = self.y_device()
        assert y.shape[0] == n, f"y_train has {y.shape[0]} entries, samples have {n} rows"
        lib = L.load()
        out = torch.empty(j, dtype=torch.float64, device=f.device)
        ws_bytes = lib.pls_cost_value_workspace_bytes(n, j)
        ws = torch.empty(max(ws_bytes // 8, 1), dtype=torch.float64, device=f.device)
        d = self.desc()
        L.check(
            lib.pls_cost_value(d, f.data_ptr(), L.ld(f), y.data_ptr(), n, j, out.data_ptr(), ws.data_ptr(), ws_bytes,
                               L.stream_ptr()),
            "pls_cost_value",
        )
        return out

    def calculate_cost_derivative(
        self, untransformed_train_prediction_samples: torch.Tensor, force_autograd: bool = False
    ) -> torch.Tensor:
        """(N, J) -> (N, J): d cost / d f (closed form or the autograd value, like the reference's dispatch)."""
        f = L.require_gpu_tensor(untransformed_train_prediction_samples, "untransformed_train_prediction_samples", promote=True)
        f = f if f.stride(-1) == 1 else f.contiguous()
        n, j = f.shape
        y = self.y_device()
        assert y.shape[0] == n, f"y_train has {y.shape[0]} entries, samples have {n} rows"
        g = torch.empty((n, j), dtype=torch.float64, device=f.device)
        d = self.desc(force_autograd=force_autograd)
        L.check(
            L.load().pls_cost_derivative(d, f.data_ptr(), L.ld(f), y.data_ptr(), n, j, g.data_ptr(), L.ld(g), L.stream_ptr()),
            "pls_cost_derivative",
        )
        return g

    def sample_observation_noise(self, number_of_particles: int, seed: int | None = None, j_offset: int = 0,
                                 normal_stream: str | None = None) -> torch.Tensor:
        """costs/base.py:86-115: one N(0, observation_noise^2) draw per particle (observation_noise is a STD here, SURVEY
        H6).  The normals come from samplers.DEFAULT_NORMAL_STREAM ("auto": the reference's stream unless the run is
        J-sharded): "device" = libplship's generator keyed by ``seed`` (or
        one draw from torch's global generator) and the GLOBAL particle index ``j_offset`` + column, so a J-sharded
        prediction gives every particle its own draw whatever the GPU count; "reference" = torch.normal on the host
        generator, the reference's stream."""
        if self.observation_noise is None:
            return torch.zeros(number_of_particles, dtype=torch.float64, device="cuda")
        from .. import samplers

        stream = samplers.resolve_normal_stream(normal_stream, j_offset)
        if stream == "reference":
            generator = torch.Generator().manual_seed(seed) if seed is not None else None
            noise = torch.normal(
                mean=0.0, std=self.observation_noise, size=(number_of_particles,), generator=generator
            ).flatten()
            return _dev(noise)
        z = samplers.standard_normals(1, (number_of_particles,), seed=seed, normal_stream="device", j_offset=j_offset)
        return (z.reshape(-1) * float(self.observation_noise)).contiguous()

    def predict_samples(
        self, untransformed_samples: torch.Tensor, observation_noise: torch.Tensor | None = None, j_offset: int = 0
    ) -> torch.Tensor:
        """costs/base.py:117-133.  ``j_offset`` (extension): global index of the first particle column (J-sharded runs)."""
        if observation_noise is None:
            observation_noise = self.sample_observation_noise(number_of_particles=untransformed_samples.shape[1],
                                                              j_offset=j_offset)
        if getattr(self.link_function, "kind", None) is not None:  # one kernel: link(f + eps_j)
            return self.link_function._native_transform(untransformed_samples, col_offset=_dev(observation_noise))
        return self.link_function(untransformed_samples + observation_noise[None, :])


def _native_base_of(cls) -> type:
    """The library class in cls's MRO that defines the native behaviour (cls itself for library classes)."""
    for c in cls.__mro__:
        if c.__module__.startswith(__name__.rsplit(".", 1)[0]):
            return c
    return PLSCost
