#!/usr/bin/env python3
"""
Build container only (needs /root/reference): regenerate every fixture with make_golden.py into a scratch
directory and compare it, array for array (NaN equal to NaN), with the committed file.

    python3 tests/golden/verify_golden.py
"""
import glob
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def same(x, y):
    if x.shape != y.shape or x.dtype != y.dtype:
        return False
    if x.dtype == object:
        return all(same(np.asarray(p), np.asarray(q)) for p, q in zip(x.ravel(), y.ravel()))
    if x.dtype.kind in "fc":
        return bool(np.array_equal(x, y, equal_nan=True))
    return bool(np.array_equal(x, y))


def main():
    with tempfile.TemporaryDirectory(prefix="golden_") as tmp:
        env = dict(os.environ, PROSSTT_GOLDEN_OUT=tmp, PYTHONDONTWRITEBYTECODE="1")
        subprocess.check_call([sys.executable, os.path.join(HERE, "make_golden.py")], env=env, stdout=subprocess.DEVNULL)
        fresh = sorted(glob.glob(os.path.join(tmp, "*.npz")))
        committed = sorted(glob.glob(os.path.join(HERE, "*.npz")))
        assert [os.path.basename(f) for f in fresh] == [os.path.basename(f) for f in committed], "file sets differ"
        arrays = bad = 0
        for f, g in zip(fresh, committed):
            a, b = np.load(f, allow_pickle=True), np.load(g, allow_pickle=True)
            assert sorted(a.files) == sorted(b.files), os.path.basename(f)
            for k in a.files:
                arrays += 1
                if not same(a[k], b[k]):
                    bad += 1
                    print("DIFFERS: %s[%s]" % (os.path.basename(f), k))
        print("%d files, %d arrays, %d differ from the committed fixtures" % (len(fresh), arrays, bad))
        return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
