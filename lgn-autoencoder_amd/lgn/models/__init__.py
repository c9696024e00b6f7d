"""lgn.models -- the module API the reference exposes (lgn/models/__init__.py:1-5)."""
from .encoder import LGNEncoder
from .decoder import LGNDecoder

__all__ = ["LGNEncoder", "LGNDecoder"]
