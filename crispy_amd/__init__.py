"""crispy_amd -- MI355X (gfx950) implementation of crispy's audio compute hot path.

The product is `libcrispy_hip.so` (C ABI: include/crispy_hip.h); this package is the thin
Python host side that mirrors the reference's adapter interfaces on top of it."""
from .rnn_weights import BLOB_BYTES, synthetic_weights, load_rnnoise_nu_text  # noqa: F401

__all__ = ["BLOB_BYTES", "synthetic_weights", "load_rnnoise_nu_text"]
__version__ = "0.1.0"
