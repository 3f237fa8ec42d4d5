"""Per-phase shader-clock breakdown of k_disc_chain4 / k_disc_chain (test hook sg_test_disc_phase_times).
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
fn = _lib.load_test().sg_test_disc_phase_times
disc.update_gail_dyn(loader, ro)  # warm
_lib.check_test(fn(disc.h, 1, None, 0))
disc.update_gail_dyn(loader, ro)
thin = os.environ.get("SG_DISC_CHAIN", "thin") != "wide"
nb = (12 if thin else 2) * ((w["B"] + 15) // 16)
buf = (C.c_longlong * (32 * nb))()
_lib.check_test(fn(disc.h, 1, buf, nb))
t = np.array(buf, dtype=np.int64).reshape(nb, 32)
for name, b in ((("mix block 0", 0), ("mix block 9", 9), ("BCE block", nb // 3 + 5)) if thin else (("BCE block 0", 0), ("mix block", nb // 2))):
    row = t[b]
    idx = sorted([i for i in range(32) if row[i] != 0 and i not in (28, 29)], key=lambda i: row[i])
    print(name, "total cycles", row[idx[-1]] - row[idx[0]])
    for i0, i1 in zip(idx[:-1], idx[1:]):
        print(f"   phase {i0:2d}->{i1:2d}: {row[i1] - row[i0]:8d} cycles")
# wall-clock (100 MHz) timeline of the LAST step of the epoch: chain blocks, then k_disc_wgrad blocks by role
full = (C.c_longlong * (32 * 512))()
_lib.check_test(fn(disc.h, 1, full, 512))
f = np.array(full, dtype=np.int64).reshape(512, 32)
cs, ce = f[:nb, 28], f[:nb, 29]
t0 = cs.min()
print(f"chain: blocks start {10 * (cs.min() - t0)}..{10 * (cs.max() - t0)} ns, end {10 * (ce.min() - t0)}..{10 * (ce.max() - t0)} ns")
wg = f[256:].reshape(-1)[: 4 * 400].reshape(400, 4)
G = (w["B"] + 15) // 16
kf, kh = (w["F"] + 15) // 16, (w["Hd"] + 15) // 16
nt, nv = kh * kh + kh * kf, (3 * 16 * kh + 4 + 63) // 64
st = wg[:, 0]
live = st[st > 0]
print(f"wgrad: all blocks start {10 * (live.min() - t0)}..{10 * (live.max() - t0)} ns after the chain's first block")
tl = wg[wg[:, 1] > 0]     # tile blocks are the ones that stamp their load / reduce / store points
for nm, k in (("operands loaded + MFMA", 1), ("LDS reduce barrier", 2), ("Adam + stores issued", 3)):
    dt = 10 * (tl[:, k] - tl[:, 0])
    print(f"wgrad tiles: {nm:24s} at +{dt.min()}..{dt.max()} ns (median {int(np.median(dt))})")
sp = wg[(wg[:, 1] == 0) & (wg[:, 3] > 0)]
if len(sp):
    print(f"wgrad Adam-scalar lane (double pow): done at +{10 * int((sp[:, 3] - sp[:, 0]).max())} ns")
_lib.check_test(fn(disc.h, 0, None, 0))
