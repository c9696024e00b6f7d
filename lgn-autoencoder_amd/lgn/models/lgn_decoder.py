"""Module path the reference's callers import (utils/train.py:5: ``from lgn.models.lgn_decoder import LGNDecoder``).
The class itself lives in lgn/models/decoder.py."""
from .decoder import LGNDecoder

__all__ = ["LGNDecoder"]
