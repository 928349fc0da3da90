"""Build-owned minimal stand-in for the `gym` package (gym is not installed and
there is no network).  It exists ONLY so that `tests/golden/make_golden.py` can
import the upstream reference in the build container and record golden vectors.
It is not part of the product, is never imported by `mate_amd`, and does not
travel any reference code: only the small slice of the classic gym<=0.21 API the
reference touches is provided here (Env, Wrapper, spaces, seeding, registry).
"""
import numpy as _np

if not hasattr(_np, 'bool8'):  # numpy>=2 dropped the alias the reference uses
    _np.bool8 = _np.bool_

from gym import error, logger, spaces, utils  # noqa: E402
from gym.core import ActionWrapper, Env, ObservationWrapper, RewardWrapper, Wrapper  # noqa: E402
from gym.registry import make, register, registry  # noqa: E402

__version__ = '0.21.0'
