"""The verification mode's double -> torus conversion on the device against the reference's formula
(reference src/types.hpp:28-33) on 1.9 million values, among them the neighbourhoods where its shortcut
for values outside (-1/2, 0) could differ (tests/hip/convert_check.hip, built by `make`)."""
import pathlib
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_conversion_matches_the_reference_formula():
    exe = ROOT / "tests" / "hip" / "convert_check"
    if not exe.exists():
        subprocess.run(["make", "-C", str(ROOT), "tests/hip/convert_check"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "convert_check ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
