"""lgn/models/utils.py:4-42 of the reference."""
from .common import adapt_var_list

__all__ = ["adapt_var_list"]
