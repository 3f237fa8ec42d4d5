"""Per-phase shader-clock breakdown of k_ppo_fwd (test hook sg_test_ppo_phase_times).
Run on the GPU box:  python tools/ppo_phase_times.py [workload]"""
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
_lib.check(lib.sg_rollout_compute_returns_policy(ro.h, pol.h, 1, 0.99, 0.95, 1))
fn = _lib.load_test().sg_test_ppo_phase_times
agent.update(ro)
_lib.check_test(fn(agent.h, 1, None, 0))
agent.update(ro)
nb = 1024
buf = (C.c_longlong * (16 * nb))()
_lib.check_test(fn(agent.h, 1, buf, nb))
t = np.array(buf, dtype=np.int64).reshape(nb, 16)
names = ["rows->LDS (fused bwd: until the weight commit)", "commit+sync", "L1", "L2", "head"]
t1 = t[512:]
t = t[:256]
t = t[t[:, 8] > 0]          # row groups that exist at this launch geometry (16- or 32-row groups)
if t[:, 5].max() == 0:      # fused backward: stamps 0..4 sit inside its forward recomputation, measured from stamp 8
    t[:, 5] = t[:, 4]
    t[:, 0:5] = np.concatenate([t[:, 8:9], t[:, 0:4]], axis=1)
d = np.diff(t[:, :6], axis=1)
print("k_ppo_fwd phases (or the fused backward's head), median cycles over", nb, "row groups:")
for i, n in enumerate(names):
    print(f"   {n:12s} {int(np.median(d[:, i])):8d}   (min {int(d[:, i].min())}, max {int(d[:, i].max())})")
print("   total       ", int(np.median(t[:, 5] - t[:, 0])))
names = ["stage+rows", "loss", "head TN", "head NN", "W2 TN", "W2 NN", "W1 TN"]
d = np.diff(t[:, 8:16], axis=1)
print("k_ppo_bwd phases (trunk 0), median cycles over", nb, "row groups:")
for i, n in enumerate(names):
    print(f"   {n:12s} {int(np.median(d[:, i])):8d}   (min {int(d[:, i].min())}, max {int(d[:, i].max())})")
print("   total       ", int(np.median(t[:, 15] - t[:, 8])))
_lib.check_test(fn(agent.h, 0, None, 0))

st, en = t[:, 6], t[:, 7]     # wall clock, 100 MHz, comparable across CUs
ok = st > 0
print(f"k_ppo_bwd trunk-0 blocks with stamps: {int(ok.sum())}; first block start -> last block start {10 * int(st[ok].max() - st[ok].min())} ns, "
      f"first start -> last end {10 * int(en[ok].max() - st[ok].min())} ns, median block {10 * int(np.median(en[ok] - st[ok]))} ns")
st1, en1 = t1[:, 6], t1[:, 7]
ok1 = st1 > 0
if ok1.any():
    print(f"k_ppo_bwd trunk-1 blocks: {int(ok1.sum())}; start {10 * int(st1[ok1].min() - st[ok].min())}..{10 * int(st1[ok1].max() - st[ok].min())} ns "
          f"after the first trunk-0 block, last end {10 * int(en1[ok1].max() - st[ok].min())} ns, median block {10 * int(np.median(en1[ok1] - st1[ok1]))} ns")
