"""How fast one CU takes in a freshly written weight image (test hook sg_test_fetch_probe): every wave of a block issues
27 loads of 1 KiB, as k_disc_chain4's waves do for their slices of W1 | W2 | W2^T | W1^T.
Run on the GPU box:  python tools/fetch_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simgan_amd import _lib  # noqa: E402

ctx = _lib.Context.default()
fn = _lib.load_test().sg_test_fetch_probe
NL = 27
print("blocks waves mode                       bytes/block   first pass: cycles (B/clk)    re-read from L2: cycles (B/clk)")
for mode, name in ((0, "16-B loads, shared image"), (8, "same, nontemporal writer"), (1, "16-B loads, own copy/block"),
                   (2, "4-B loads, shared image"), (4, "16-B loads, 32 of 64 lanes")):
    for waves in (7, 4, 1):
        for nb in (1, 8, 96, 256):
            buf = (C.c_longlong * (3 * nb))()
            _lib.check_test(fn(ctx.h, nb, waves, mode, buf))
            t = np.array(buf, dtype=np.int64).reshape(nb, 3)
            nbytes = waves * NL * 1024 // (2 if mode == 4 else 1)
            f, w = np.median(t[:, 0]), np.median(t[:, 1])
            print(f"{nb:6d} {waves:5d} {name:28s} {nbytes:9d}   {f:9.0f} ({nbytes / f:5.1f})   max {t[:, 0].max():7d}      {w:9.0f} ({nbytes / w:5.1f})"
                  f"   first load back after {np.median(t[:, 2]):6.0f}")
