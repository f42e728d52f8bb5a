"""Drop-in for the reference's scripts/featureExtractor.py (`from featureExtractor import *` in
getEmbeddingExample.py:5): same function names, computed with torch on the device."""
import os as _os
import sys as _sys

import numpy as np  # noqa: F401

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from doubleattentionspeakerverification_amd.featureExtractor import *  # noqa: E402,F401,F403
