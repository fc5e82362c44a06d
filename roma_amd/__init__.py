"""roma_amd - MI355X-native RoMa dense matching (`RegressionMatcher.match()` hot path).

Public surface mirrors the reference package (`romatch/__init__.py:2`): model factories and the
local-correlation operator.  Everything numerical runs in libroma_hip.so (hand-written HIP, gfx950).
"""
from .matcher import RegressionMatcher, roma_indoor, roma_model, roma_outdoor  # noqa: F401
from .local_correlation import local_corr, local_correlation  # noqa: F401
from .kde import kde  # noqa: F401
from .sampling import multinomial  # noqa: F401
from .tiny import TinyRoMa, tiny_roma_v1_outdoor  # noqa: F401

__all__ = ["RegressionMatcher", "roma_model", "roma_outdoor", "roma_indoor", "local_corr", "local_correlation", "kde",
           "multinomial", "TinyRoMa", "tiny_roma_v1_outdoor"]
