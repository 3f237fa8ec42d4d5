"""Micro-benchmark of the LDS/MFMA tile engine: cycles per workgroup-level layer GEMM.
Run on the GPU box:  python tools/gemm_bench.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simgan_amd import _lib  # noqa: E402

ctx = _lib.Context.default()
fn = _lib.load_test().sg_test_gemm_bench
ITERS = 200
print("mode MT   K  Np thr epi | cycles/iter | MFMA-bound(cyc) | eff")
for mode, MT, K, Np in [(0, 1, 112, 112), (0, 1, 96, 112), (0, 2, 112, 112), (0, 4, 64, 64), (1, 1, 112, 112),
                        (1, 1, 112, 96), (1, 2, 112, 112), (2, 1, 112, 112), (2, 1, 112, 96), (2, 2, 112, 112)]:
    for thr in (256, 512):
        for epi in ((0, 1) if mode == 0 else (0,)):
            cyc = C.c_longlong(0)
            _lib.check_test(fn(ctx.h, mode, MT, K, Np, thr, ITERS, epi, C.byref(cyc)))
            per = cyc.value / ITERS
            if mode == 2:
                mfma = (K // 16) * (Np // 16) * (16 * MT // 4)   # tiles x MFMAs (K index = rows R=16*MT)
            else:
                mfma = MT * (Np // 16) * (K // 4)
            bound = mfma * 32 / 4
            print(f"{mode:4d} {MT:2d} {K:4d} {Np:3d} {thr:4d} {epi:3d} | {per:11.0f} | {bound:15.0f} | {bound / per:4.2f}")
