"""
lgn -- MI355X-native drop-in for the message-passing hot path of zichunhao/lgn-autoencoder.

Put this directory's parent (``lgn-autoencoder_amd/``) on ``sys.path`` *before* the reference
checkout and the reference's ``main.py`` / ``test.py`` pick up these ``lgn.models.LGNEncoder`` /
``LGNDecoder`` unchanged.  All numerics run in hand-written HIP kernels
(``csrc/`` -> ``lgn/_lib/liblgn_amd.so``); there is no CPU or PyTorch fallback.
"""
__all__ = ["models", "g_lib", "cg_lib", "nn"]
