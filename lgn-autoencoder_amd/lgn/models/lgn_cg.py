"""Module path of the reference's LGNCG (lgn/models/lgn_cg.py:8-180); the parameter container is lgn.nn.LGNCG and the
forward loop is lgn/models/common.py:run_levels (one native level call + one native CGMLP call per level)."""
from ..nn import LGNCG

__all__ = ["LGNCG"]
