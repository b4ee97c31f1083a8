"""set_seed (drop-in for src/utils.py:8-22): seeds every generator the reference's loops draw from."""
import os
import random

import numpy as np
import torch


def set_seed(seed: int = 42) -> None:
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)  # the per-step Philox key is drawn from this generator (basis/base.py: _draw_noise_spec)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
