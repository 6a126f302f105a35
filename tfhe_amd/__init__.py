"""tfhe_amd -- MI355X-native gate bootstrapping (KowerKoint/TFHE's HomNAND::nand path).

This package is a thin ctypes binding of libtfhe_amd.so (C ABI in include/tfhe_amd.h),
used by the tests and the benchmark.  There is no CPU fallback: importing works without
the library, but every operation raises if it is missing.
"""
from .binding import (Context, MultiContext, Params, TfheAmdError, cmux, external_product, lib_path, library,  # noqa: F401
                      keygen, load_error, params_default, seed_words, version)
