"""kiwi_amd -- MI355X-native forward-modelling + misfit engine for Kiwi's inner inversion loop.

The product is the C-ABI library ``kiwi_amd/libkiwi_hip.so`` (include/kiwi_hip.h, sources in
kiwi_amd/csrc); this package is the Python host side of it (the reference's Python layer,
python/tunguska/seismosizer.py, drives the Fortran engine the same way).
"""
from .lib import build, load, KiwiHipError  # noqa: F401
from .engine import Engine  # noqa: F401
from . import gridsearch, lm  # noqa: F401
