from .base import NoiseSpec, PLSBasis
from .inducing_point import InducingPointBasis
from .orthonormal import OrthonormalBasis

__all__ = ["PLSBasis", "NoiseSpec", "InducingPointBasis", "OrthonormalBasis"]
