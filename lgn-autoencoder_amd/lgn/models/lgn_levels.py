"""Module path of the reference's LGNNodeLevel / CGMLP (lgn/models/lgn_levels.py:9-241); parameter containers in lgn.nn,
arithmetic in csrc/level_*.hip and csrc/mlp_mfma*.hip."""
from ..nn import CGMLP, LGNNodeLevel

__all__ = ["LGNNodeLevel", "CGMLP"]
