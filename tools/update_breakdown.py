"""Wall-clock breakdown of one learner update (every call below is synchronous: it returns losses).
Run on the GPU box:  python tools/update_breakdown.py [workload]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simgan_amd as sg  # noqa: E402
from simgan_amd import _lib  # noqa: E402
from bench import WORKLOADS, build_problem  # noqa: E402

w = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "northstar"]
pol, disc, agent, ro, loader, expert, learner = build_problem(sg, w, 0)
lib = _lib.load()
_lib.check(lib.sg_rollout_fill_synthetic(ro.h, pol.h, 1234, 0.01))
learner.update()
for it in range(3):
    t = [time.perf_counter()]
    for _ in range(learner.gail_epoch):
        disc.update_gail_dyn(loader, ro)
        t.append(time.perf_counter())
    dones = C.c_double(0)
    _lib.check(lib.sg_rollout_count_dones(ro.h, C.byref(dones)))
    t.append(time.perf_counter())
    disc.relabel_rewards(ro, 0.99, 0.1, learner.ret_rms)
    t.append(time.perf_counter())
    _lib.check(lib.sg_rollout_compute_returns_policy(ro.h, pol.h, 1, 0.99, 0.95, 1))
    t.append(time.perf_counter())
    agent.update(ro)
    t.append(time.perf_counter())
    ro.after_update()
    t.append(time.perf_counter())
    d = [1e3 * (b - a) for a, b in zip(t[:-1], t[1:])]
    n = learner.gail_epoch
    print(f"iter {it}: D epochs {[round(x, 2) for x in d[:n]]} ms | count_dones {d[n]:.2f} | relabel {d[n+1]:.2f} | "
          f"returns {d[n+2]:.2f} | ppo {d[n+3]:.2f} | after_update {d[n+4]:.2f} | total {sum(d):.2f} ms")
