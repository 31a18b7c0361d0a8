"""vers_amd -- MI355X (gfx950) IVFFlat hot path behind vers's Index API.

The product is libvers_hip.so (C ABI in include/vers_hip.h, sources in vers_amd/csrc);
this package is the thin Python host side used by tests and bench.py.
"""
