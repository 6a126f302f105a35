#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- regenerate tests/golden/*.npz from the real reference.

Runs only in the development container (needs /root/reference): builds
oracle/_ref/ref_golden (the reference's own classes, compiled from its sources in
place), runs it, and packs the .npy arrays it dumps into one compressed .npz per
vector family.  The fixtures are data only: inputs and the reference's outputs.
Keys inside them come from the reference's unseeded std::random_device
(reference src/random.hpp:10), so every regeneration yields different, equally valid
vectors.
"""
import pathlib
import subprocess
import sys
import tempfile

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
GOLDEN = HERE.parent / "tests" / "golden"


def main() -> int:
    """make_golden.py            regenerate every family
    make_golden.py kswide     only the wide-base key-switch family (tests/golden/kswide.npz)"""
    only = sys.argv[1:2]
    subprocess.run(["make", "-C", str(HERE), "ref"], check=True)
    GOLDEN.mkdir(parents=True, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([str(HERE / "_ref" / "ref_golden"), tmp] + only, check=True)
        groups = {}
        for f in sorted(pathlib.Path(tmp).glob("*.npy")):
            family, _, field = f.stem.partition("_")
            groups.setdefault(family, {})[field] = np.load(f)
        for family, arrays in groups.items():
            out = GOLDEN / f"{family}.npz"
            np.savez_compressed(out, **arrays)
            print(f"{out.name}: {sorted(arrays)} {out.stat().st_size} bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
