"""MI355X-native projected Langevin sampling: the per-step particle update of
jswu18/projected-langevin-sampling behind the reference's PLS / basis / cost / link-function API.

All numerics run in libplship.so (hand-written HIP for gfx950, see csrc/); this package is the thin host
side: it owns device memory through torch tensors and calls the C ABI (include/plship.h) with raw pointers."""
from . import _lib
from .kernel import ARDKernel, LinearKernel, PLSKernel
from .projected_langevin_sampling import PLS
from .trainers import EarlyStopper, train_pls, train_pls_captured

__all__ = ["PLS", "PLSKernel", "ARDKernel", "LinearKernel", "EarlyStopper", "train_pls", "train_pls_captured", "_lib"]
