"""exaconstit_amd — MI355X-native hot path of LLNL/ExaConstit behind a C ABI (include/exaconstit_hip.h).

The product is the HIP library `libexaconstit_hip.so` (built from exaconstit_amd/csrc) plus the C++ host driver that
mirrors the reference's operator/solver classes.  This Python package is only a ctypes binding used by the tests,
bench.py and __graft_entry__.py; it never falls back to a CPU path: importing `exaconstit_amd.lib` without the built
library raises.
"""
from . import lib  # noqa: F401
