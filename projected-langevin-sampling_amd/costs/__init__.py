from .base import PLSCost
from .bernoulli import BernoulliCost
from .gaussian import GaussianCost
from .multimodal import MultiModalCost
from .poisson import PoissonCost
from .student_t import StudentTCost

__all__ = ["PLSCost", "BernoulliCost", "GaussianCost", "PoissonCost", "StudentTCost", "MultiModalCost"]
