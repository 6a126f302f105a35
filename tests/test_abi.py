"""The C-ABI library loads and exports every symbol include/tfhe_amd.h declares; argument
checking that needs no GPU."""
import ctypes as C
import pathlib
import re

import numpy as np
import pytest

import tfhe_amd

HEADER = pathlib.Path(__file__).resolve().parent.parent / "include" / "tfhe_amd.h"


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(tfhe_amd_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ["tfhe_amd_ctx_create", "tfhe_amd_blind_rotate_batch", "tfhe_amd_external_product_batch",
                 "tfhe_amd_cmux_batch", "tfhe_amd_ctx_destroy", "tfhe_amd_nand_batch",
                 "tfhe_amd_nand_batch_device", "tfhe_amd_bootstrap_batch", "tfhe_amd_bootstrap_batch_device",
                 "tfhe_amd_gate_bootstrapping_batch_device", "tfhe_amd_identity_key_switch_batch_device"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = tfhe_amd.library()  # raises if the .so is missing: there is no fallback
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_params_and_sizes():
    p = tfhe_amd.params_default()
    assert (p.n, p.N, p.K, p.bg_bit, p.L, p.T, p.base_bit) == (636, 512, 2, 8, 2, 5, 2)
    lib = tfhe_amd.library()
    assert lib.tfhe_amd_bk_words(C.byref(p)) == 636 * 6 * 3 * 512      # 23,445,504 bytes
    assert lib.tfhe_amd_ks_words(C.byref(p)) == 1024 * 5 * 4 * 637     # 52,183,040 bytes
    assert lib.tfhe_amd_params_supported(C.byref(p)) == 0
    bad = tfhe_amd.Params(636, 500, 2, 8, 2, 5, 2)  # ring degree not a power of two
    assert lib.tfhe_amd_params_supported(C.byref(bad)) == 1
    bad = tfhe_amd.Params(636, 512, 2, 8, 4, 5, 2)  # L * bg_bit > 31
    assert lib.tfhe_amd_params_supported(C.byref(bad)) == 1
    assert tfhe_amd.version().count(".") == 2


def test_every_parameter_set_the_reference_accepts_is_supported():
    """The reference's constraints are static_asserts and shift widths: a power-of-two ring
    (src/trgsw.hpp:142-143), BG_BIT in {8, 16, 32} with L * BG_BIT <= 31 (src/types.hpp:11-19,
    src/trgsw.hpp:28-29), T * BASE_BIT <= 31 (src/tlwe_key_switching.hpp:52-53).  Every such set has
    kernels; everything else is an argument error."""
    lib = tfhe_amd.library()
    for N in (2, 16, 512, 1024, 2048, 4096, 8192, 1 << 16):
        for K in (1, 2, 3, 5):
            for bg, L in ((8, 1), (8, 2), (8, 3), (16, 1)):
                for T, bb in ((5, 2), (8, 2), (10, 3), (7, 4), (6, 5), (3, 7), (3, 10), (1, 16), (1, 31), (31, 1)):
                    p = tfhe_amd.Params(7, N, K, bg, L, T, bb)
                    assert lib.tfhe_amd_params_supported(C.byref(p)) == 0, (N, K, bg, L, T, bb)
    for bad in ((7, 512, 1, 8, 2, 8, 4), (7, 512, 1, 16, 2, 5, 2), (7, 512, 1, 32, 1, 5, 2), (7, 768, 1, 8, 2, 5, 2),
                (0, 512, 1, 8, 2, 5, 2), (7, 512, 0, 8, 2, 5, 2), (7, 512, 1, 8, 2, 0, 2), (7, 512, 1, 8, 2, 5, 0)):
        assert lib.tfhe_amd_params_supported(C.byref(tfhe_amd.Params(*bad))) == 1, bad


def test_argument_errors_without_gpu():
    lib = tfhe_amd.library()
    p = tfhe_amd.params_default()
    h = C.c_void_p()
    assert lib.tfhe_amd_ctx_create(C.byref(p), None, None, 0, 0, C.byref(h)) == 1
    assert b"null" in lib.tfhe_amd_last_error()
    assert lib.tfhe_amd_nand_batch(None, None, None, None, 4) == 1
    with pytest.raises(ValueError):
        tfhe_amd.Context(p, np.zeros(3, np.uint32), np.zeros(3, np.uint32))


def test_header_is_plain_c_and_links(tmp_path):
    """include/tfhe_amd.h compiles as strict C99 (no C++ or torch type crosses the boundary) and a C
    program that takes the address of every declared entry point links against the library."""
    import subprocess

    root = HEADER.parent.parent
    src = tmp_path / "abi.c"
    refs = "\n".join(f"    keep[{i}] = (void (*)(void))&{name};" for i, name in enumerate(declared_functions()))
    src.write_text('#include "tfhe_amd.h"\n#include <stdio.h>\nint main(void) {\n'
                   f"    void (*keep[{len(declared_functions())}])(void);\n{refs}\n"
                   "    tfhe_amd_params p;\n    tfhe_amd_params_default(&p);\n"
                   '    printf("%d %s %d\\n", p.N, tfhe_amd_version(), keep[0] != 0);\n    return 0;\n}\n')
    exe = tmp_path / "abi"
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", str(root / "include"), str(src),
                        "-L", str(root / "tfhe_amd"), "-ltfhe_amd", f"-Wl,-rpath,{root / 'tfhe_amd'}", "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.split()[0] == "512", out.stdout + out.stderr
