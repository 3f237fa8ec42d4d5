"""Per-phase shader-clock breakdown of k_disc_grad (test hook sg_test_disc_phase_times).
Run on the GPU box:  python tools/phase_times.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simgan_amd as sg  # noqa: E402
from simgan_amd import _lib  # noqa: E402
from bench import WORKLOADS, build_problem  # noqa: E402

w = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "northstar"]
pol, disc, agent, ro, loader, expert, learner = build_problem(sg, w, 0)
lib = _lib.load()
_lib.check(lib.sg_rollout_fill_synthetic(ro.h, pol.h, 1234, 0.01))
fn = lib.sg_test_disc_phase_times
fn.restype = C.c_int
fn.argtypes = [_lib.H, C.c_int, C.POINTER(C.c_longlong), C.c_int]
disc.update_gail_dyn(loader, ro)  # warm
_lib.check(fn(disc.h, 1, None, 0))
disc.update_gail_dyn(loader, ro)
nb = 2 * ((w["B"] + 15) // 16)
buf = (C.c_longlong * (32 * nb))()
_lib.check(fn(disc.h, 1, buf, nb))
t = np.array(buf, dtype=np.int64).reshape(nb, 32)
for name, b in (("BCE block 0", 0), ("mix block", nb // 2)):
    row = t[b]
    idx = [i for i in range(32) if row[i] != 0]
    print(name, "total cycles", row[idx[-1]] - row[idx[0]])
    for i0, i1 in zip(idx[:-1], idx[1:]):
        print(f"   phase {i0:2d}->{i1:2d}: {row[i1] - row[i0]:8d} cycles")
_lib.check(fn(disc.h, 0, None, 0))
