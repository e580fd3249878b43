"""Seeding contract of the reference (utils.py:6-14)."""
import os
import random

import numpy as np
import torch


def configure_seed(seed):
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.backends.cudnn.deterministic = True     # MIOpen is never used by this engine; kept for parity
        torch.backends.cudnn.benchmark = False
    # the HIP noise streams behind the non-injected randn_like draws restart with the seed, like torch's generator
    from . import ops
    ops.reset_noise()
