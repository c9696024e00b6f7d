"""Module path the reference's callers import (utils/train.py:6: ``from lgn.models.lgn_encoder import LGNEncoder``).
The class itself lives in lgn/models/encoder.py; the pooling helpers the reference keeps in this module
(lgn/models/lgn_encoder.py:419-583) are the ones of lgn/ops.py."""
from ..ops import aggregate_latent as aggregate          # lgn_encoder.py:419-496 (dict-of-tensors form)
from ..ops import pool_max as get_max_features           # lgn_encoder.py:561-583
from ..ops import pool_min as get_min_features           # lgn_encoder.py:540-558
from ..ops import _msq as get_msq                        # lgn_encoder.py:499-505
from .encoder import LGNEncoder

__all__ = ["LGNEncoder", "aggregate", "get_min_features", "get_max_features", "get_msq"]
