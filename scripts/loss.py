"""Drop-in for the reference's scripts/loss.py: the reference's own train.py and
getEmbeddingExample.py do `sys.path.append('./scripts/')` then `from loss import *`.
No __all__ on purpose: like the reference module, a star-import also passes on torch/nn/F."""
import os as _os
import sys as _sys

import numpy as np  # noqa: F401
import torch  # noqa: F401
from torch import nn  # noqa: F401
from torch.nn import functional as F  # noqa: F401

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from doubleattentionspeakerverification_amd.loss import *  # noqa: E402,F401,F403
