"""Device-clock split of the ASYNCHRONOUS device-resident update bench.py times: one HIP event (sg_ctx_mark) between the phases of
GailDynLearner._update_resident -- each discriminator epoch (row copy + n_d graph-replayed steps), the fused relabel, returns
(+ get_value), the PPO update, after_update + publish -- read after several back-to-back updates (nothing waits inside).
Run on the GPU box:  python tools/update_marks.py [workload] [updates]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simgan_amd as sg  # noqa: E402
from simgan_amd import _lib  # noqa: E402
from bench import GAMMA, LAM, WORKLOADS, build_problem  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "northstar"
n_up = int(sys.argv[2]) if len(sys.argv) > 2 else 6
w = WORKLOADS[name]
pol, disc, agent, ro, loader, expert, learner = build_problem(sg, w, 0)
ctx, lib = ro.ctx, ro.lib
_lib.check(lib.sg_rollout_fill_synthetic(ro.h, pol.h, 1234, 0.01))
for _ in range(3):
    learner.update()
ctx.synchronize()
ctx.marks_reset()
labels, marks = [], [ctx.mark()]


def mark(label):
    labels.append(label)
    marks.append(ctx.mark())


for u in range(n_up):
    if disc is not None:
        for e in range(learner.gail_epoch):
            disc.update_gail_dyn(learner.loader, ro, fetch_losses=False)
            mark(f"D epoch {e}")
        disc.relabel_rewards_auto(ro, GAMMA, 500.0, False)
        mark("relabel")
    _lib.check(lib.sg_rollout_compute_returns_policy(ro.h, pol.h, 1, GAMMA, LAM, 1))
    mark("returns")
    agent.update(ro, fetch_losses=False)
    mark("ppo")
    ro.after_update()
    learner._ring.publish(disc, agent, ("value_loss",))
    mark("after_update + publish")
ctx.synchronize()
ms = [ctx.mark_elapsed(marks[i], marks[i + 1]) for i in range(len(labels))]
per = len(labels) // n_up
tot = {}
for i, (lab, v) in enumerate(zip(labels, ms)):
    if i >= per:                       # skip the first update (its marks follow an idle stream)
        tot.setdefault(lab, []).append(v)
print(f"{name}: device-clock milliseconds per phase, mean over {n_up - 1} back-to-back updates (min .. max)")
s = 0.0
for lab, v in tot.items():
    print(f"  {lab:24s} {np.mean(v):8.3f}   ({min(v):.3f} .. {max(v):.3f})")
    s += float(np.mean(v))
print(f"  {'sum':24s} {s:8.3f}")
n_d = min(w["Ne"] // w["B"], w["T"] * w["N"] // w["B"]) if w["E_d"] else 0
if n_d:
    print(f"  -> {1e3 * np.mean(tot['D epoch 1']) / n_d:.3f} us per discriminator step incl. the epoch's row copy ({n_d} steps per epoch)")
print(f"  -> {1e3 * np.mean(tot['ppo']) / (w['E_p'] * w['M']):.3f} us per PPO step incl. the epoch gathers ({w['E_p'] * w['M']} steps)")
