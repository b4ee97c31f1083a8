from .base import BlockSpec, NoiseSpec, PLSBasis
from .inducing_point import InducingPointBasis
from .orthonormal import OrthonormalBasis

__all__ = ["PLSBasis", "NoiseSpec", "BlockSpec", "InducingPointBasis", "OrthonormalBasis"]
