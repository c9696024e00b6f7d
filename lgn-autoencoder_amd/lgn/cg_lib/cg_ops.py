"""Import path of the reference (lgn/cg_lib/cg_ops.py): the product lives in lgn/cg_lib/product.py."""
from .product import CGProduct, cg_product, cg_product_tau  # noqa: F401
