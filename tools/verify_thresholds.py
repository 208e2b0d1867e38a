#!/usr/bin/env python3
"""tools/verify_thresholds.py -- EXHAUSTIVE check (build container only) that the 255-entry f32
threshold table reproduces the reference's encode for EVERY float32 in [0, 1]:
encode must be non-decreasing over the ordered floats and step exactly at the thresholds.
Runs the reference's own functions (animals/animal_utils.py via the placeholder-cv2 import).
~1.07e9 values; takes a few minutes single-threaded."""
import importlib
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.modules["cv2"] = types.ModuleType("cv2")
pkg = types.ModuleType("animals")
pkg.__path__ = ["/root/reference/animals"]
sys.modules["animals"] = pkg
au = importlib.import_module("animals.animal_utils")
thr = np.load(os.path.join(ROOT, "tests", "golden", "srgb_tables.npz"))["enc_thr_f32"]


def enc(x):  # animals/dog.py:54-57
    s = np.clip(au.linear_to_srgb(np.clip(x, 0.0, 1.0)), 0.0, 1.0)
    return (s * 255.0 + 0.5).astype(np.uint8)


one = int(np.array([1.0], np.float32).view(np.uint32)[0])
CH = 1 << 24
prev = 0
steps = []
for start in range(0, one + 1, CH):
    bits = np.arange(start, min(start + CH, one + 1), dtype=np.uint32)
    e = enc(bits.view(np.float32)).astype(np.int16)
    d = np.diff(np.concatenate([[prev], e]))
    assert d.min() >= 0, f"encode not monotone near bits {start}"
    idx = np.nonzero(d)[0]
    for i in idx:
        assert d[i] == 1, "encode skips a code"
        steps.append(int(bits[i]))
    prev = int(e[-1])
    if (start // CH) % 8 == 0:
        print(f"  {start/one*100:5.1f}%  codes so far {prev}", flush=True)
steps = np.array(steps, np.uint32).view(np.float32)
assert steps.size == 255 and np.array_equal(steps, thr), "threshold table mismatch"
print("OK: encode is monotone over all", one + 1, "float32 in [0,1]; its 255 steps are exactly the table")
