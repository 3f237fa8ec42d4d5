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
n = wb + 8 * (kf + kh)
buf = (C.c_longlong * (8 * 512))()
_lib.check_test(fn(disc.h, 1, buf, 512))
t = np.array(buf, dtype=np.int64).reshape(512, 8)[:n, :4] * 10  # ns
c = t[:nc]
t0 = c[:, 0].min()
mix, bce = c[:4 * G], c[4 * G:]
for nm, x in (("mixup chain blocks", mix), ("BCE chain blocks", bce)):
    print(f"{nm}: start {x[:, 0].min() - t0}..{x[:, 0].max() - t0} ns, body done {x[:, 1].min() - t0}..{x[:, 1].max() - t0}, "
          f"stores drained {x[:, 2].min() - t0}..{x[:, 2].max() - t0}, flag stored {x[:, 3].min() - t0}..{x[:, 3].max() - t0}")
wt = t[wb:]
wt = wt[wt[:, 0] > 0]
print(f"tile blocks ({len(wt)}): ready to wait {wt[:, 0].min() - t0}..{wt[:, 0].max() - t0} ns, flags seen {wt[:, 1].min() - t0}..{wt[:, 1].max() - t0} "
      f"(median {int(np.median(wt[:, 1])) - t0}), operands in + MFMA {wt[:, 2].min() - t0}..{wt[:, 2].max() - t0} (median {int(np.median(wt[:, 2])) - t0}), "
      f"LDS reduce barrier {wt[:, 3].min() - t0}..{wt[:, 3].max() - t0}")
print(f"last flag stored -> median tile block saw the flags: {int(np.median(wt[:, 1])) - c[:, 3].max()} ns; -> operands in {int(np.median(wt[:, 2])) - c[:, 3].max()} ns")
_lib.check_test(fn(disc.h, 0, None, 0))
