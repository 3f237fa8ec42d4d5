"""GPU diagnostic: operand layout of v_mfma_f32_4x4x1 with CBSZ=2 and the thin-row engine against numpy."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from simgan_amd import _lib  # noqa: E402

lib = _lib.load_test()
ctx = _lib.Context.default()
rng = np.random.default_rng(0)
a = rng.standard_normal(64).astype(np.float32)
b = rng.standard_normal(64).astype(np.float32)
for abid in range(4):
    d = np.zeros((64, 4), np.float32)
    _lib.check_test(lib.sg_test_mfma_probe(ctx.h, abid, _lib.fptr(a), _lib.fptr(b), _lib.fptr(d)))
    # hypothesis: lane l = 16 s + 4 c + j, register r:  d = a[16 s + 4 abid + r] * b[l]
    exp = np.zeros((64, 4), np.float32)
    for l in range(64):
        s = l >> 4
        for r in range(4):
            exp[l, r] = a[16 * s + 4 * abid + r] * b[l]
    print("abid", abid, "max err vs hypothesis", np.abs(d - exp).max())
    if np.abs(d - exp).max() > 1e-6:
        print(d[:8]); print(exp[:8])
for (M, N, K) in [(4, 16, 16), (4, 32, 32), (8, 112, 112), (4, 112, 96), (8, 96, 112), (4, 128, 112)]:
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    Cc = np.zeros((M, N), np.float32)
    _lib.check_test(lib.sg_test_gemm(ctx.h, 3, M, N, K, _lib.fptr(A), _lib.fptr(B), _lib.fptr(Cc)))
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    print((M, N, K), "thin gemm max err", np.abs(Cc - ref).max())
