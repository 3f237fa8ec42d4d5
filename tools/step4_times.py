"""Wall-clock timeline of k_disc_step4 (one launch per discriminator step): chain blocks, hand-off, weight-gradient blocks.
Needs a library built with -DSG_STEP4_STAMPS=1 (make -C simgan_amd/csrc CXXFLAGS+=...); run on the GPU box."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["SG_DISC_GRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simgan_amd as sg  # noqa: E402
from simgan_amd import _lib  # noqa: E402
from bench import WORKLOADS, build_problem  # noqa: E402

w = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "northstar"]
pol, disc, agent, ro, loader, expert, learner = build_problem(sg, w, 0)
lib = _lib.load()
_lib.check(lib.sg_rollout_fill_synthetic(ro.h, pol.h, 1234, 0.01))
fn = _lib.load_test().sg_test_disc_step4_times
disc.update_gail_dyn(loader, ro)  # warm
_lib.check_test(fn(disc.h, 1, None, 0))
disc.update_gail_dyn(loader, ro)
G = (w["B"] + 15) // 16
kf, kh = (w["F"] + 15) // 16, (w["Hd"] + 15) // 16
nc, wb = 12 * G, (12 * G + 7) & ~7
n = wb + 8 * (kf + kh) + 2 * G
S = 16   # stamp slots per workgroup (SG_STEP4_STAMP_SLOTS)
buf = (C.c_longlong * (S * 512))()
_lib.check_test(fn(disc.h, 1, buf, 512))
raw = np.array(buf, dtype=np.int64).reshape(512, S)[:n]
t = raw * 10  # ns
c = t[:nc]
t0 = c[:, 0].min()


def rng(x):
    x = x[x > 0]
    return f"{x.min() - t0}..{x.max() - t0} (median {int(np.median(x)) - t0})" if len(x) else "-"


mix, bce = c[:4 * G], c[4 * G:]
print(f"mixup chain blocks: start {rng(mix[:, 0])}, body done {rng(mix[:, 1])}; flag form: stores drained {rng(mix[:, 2])}, flag stored {rng(mix[:, 3])}")
print(f"BCE chain blocks:   start {rng(bce[:, 0])}, body done {rng(bce[:, 1])}; flag form: stores drained {rng(bce[:, 2])}, flag stored {rng(bce[:, 3])}")
ntv = 8 * (kf + kh)
xcd = np.arange(ntv) & 7
tt = t[wb + np.nonzero(xcd < kh)[0]]
vt = t[wb + np.nonzero(xcd >= kh)[0]]
vt = vt[vt[:, 0] > 0]
gt = t[wb + ntv:]
gt = gt[gt[:, 0] > 0]
if len(vt):
    print(f"vector blocks ({len(vt)}): start {rng(vt[:, 0])}, partials in {rng(vt[:, 1])}, reduce barrier {rng(vt[:, 2])}, stores acknowledged {rng(vt[:, 3])}")
if len(gt):
    print(f"row-copy blocks ({len(gt)}): start {rng(gt[:, 0])}, stores acknowledged {rng(gt[:, 1])}")
rounds = raw[wb + np.nonzero(xcd < kh)[0]][:, 7]
print(f"tile blocks ({len(tt)}), wave 0: ready {rng(tt[:, 0])}")
print(f"  BCE half:   requested {rng(tt[:, 1])}, first answer {rng(tt[:, 2])}, contracted {rng(tt[:, 3])}, re-request rounds {np.bincount(rounds & 0xffff)}")
print(f"  mixup half: requested {rng(tt[:, 4])}, first answer {rng(tt[:, 5])}, contracted {rng(tt[:, 6])}, re-request rounds {np.bincount(rounds >> 16)}")
print(f"  past the reduce barrier {rng(tt[:, 8])}, Adam done and stores acknowledged {rng(tt[:, 9])}")
print(f"last chain body done -> median tile wave 0 has its operands contracted: {int(np.median(tt[:, 6])) - c[:, 1].max()} ns; -> last tile block's stores acknowledged: {tt[:, 9].max() - c[:, 1].max()} ns")
_lib.check_test(fn(disc.h, 0, None, 0))
