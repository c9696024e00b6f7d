"""lgn.models -- the module API the reference exposes (lgn/models/__init__.py:1-5)."""
from .lgn_levels import LGNNodeLevel, CGMLP
from .lgn_cg import LGNCG
from .lgn_encoder import LGNEncoder
from .lgn_decoder import LGNDecoder
from .utils import adapt_var_list

__all__ = ["LGNEncoder", "LGNDecoder", "LGNCG", "LGNNodeLevel", "CGMLP", "adapt_var_list"]
